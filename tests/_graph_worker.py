"""Worker of tests/test_graph_step_gpu.py (one process per case: a capture that goes wrong inside the HIP runtime must
not take the test session with it): eager steps against the same number of steps replayed from a HIP graph."""

import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DEV = "cuda:0"


def _build(sharded: bool, rccl: bool = False, optimizer: str = "adagrad"):
    import keras_rs_amd.layers as kl
    from keras_rs_amd.layers import base

    B, D, hots, vocabs = 33024, 32, [3, 1, 7, 2], [5000, 300, 20000, 1000]   # (above autograd.WGRAD_SIDE_MIN_ROWS)
    # "adam": the bias correction changes with every update; "sched": a learning-rate schedule called with the update count
    # (jax/config_conversion.py:136-176) -- both are host-computed per step and reach the kernels through device memory
    opt = {"adagrad": lambda: kl.Adagrad(learning_rate=0.05, initial_accumulator_value=0.1),
           "adam": lambda: kl.Adam(learning_rate=0.01),
           "sched": lambda: kl.Adagrad(learning_rate=lambda step: 0.05 / (1.0 + step), initial_accumulator_value=0.1)}[optimizer]()
    feats = {}
    for t in range(4):
        tc = kl.TableConfig(name=f"t{t}", vocabulary_size=vocabs[t], embedding_dim=D,
                            initializer=base.RandomUniform(-0.05, 0.05, seed=7 + t), optimizer=opt, combiner="sum",
                            placement="sparsecore")
        feats[f"f{t}"] = kl.FeatureConfig(f"f{t}", tc, (B, hots[t]), (B, D))
    if sharded:
        from keras_rs_amd.sharded import ShardedDistributedEmbedding

        emb = ShardedDistributedEmbedding(feats, dtype="bfloat16", slab_lead_cols=D, exchange="static")
        emb._collectives_at_world1 = rccl       # the all-to-alls through the one-rank RCCL communicator
    else:
        emb = kl.DistributedEmbedding(feats, dtype="bfloat16", slab_lead_cols=D)
    dot = kl.DotInteraction(dtype="bfloat16")
    cross = torch.nn.ModuleList(kl.FeatureCross(projection_dim=64, kernel_initializer=base.GlorotUniform(seed=3 + i),
                                                dtype="mixed_bfloat16") for i in range(2))
    g = torch.Generator(device=DEV).manual_seed(5)
    ids = {f"f{t}": torch.randint(0, vocabs[t], (B, hots[t]), device=DEV, generator=g, dtype=torch.int32) for t in range(4)}
    dense = (torch.rand(B, D, device=DEV, generator=g) * 0.9).to(torch.bfloat16)
    pre = emb.preprocess(ids)
    g_xl = torch.full((B, 5 * D), 1.0 / B, dtype=torch.bfloat16, device=DEV)
    g_in = torch.full((B, 10), 0.1 / B, dtype=torch.bfloat16, device=DEV)
    box = [None]
    red = [None]

    def step():
        out = emb(pre)
        fs = [dense] + [out[k] for k in out]
        inter = dot(fs)
        x0 = kl.concat_features(fs)
        xl = x0
        for layer in cross:
            xl = layer(x0, xl)
        torch.autograd.backward([xl, inter], [g_xl, g_in])
        if box[0] is None:
            from keras_rs_amd.optim import Adagrad

            params = [p for layer in cross for p in layer.parameters()]
            box[0] = Adagrad(params, lr=0.01, initial_accumulator_value=0.1, prepare_casts=True)
            if rccl:
                # the dense weights' data-parallel all-reduce as well (hooks from the next backward on)
                from keras_rs_amd.dp import GradAllReduce

                red[0] = GradAllReduce(params, run_at_world1=True)
                for p in params:
                    red[0].launch(p)
        if red[0] is not None:
            red[0].wait()
        box[0].step()
        box[0].zero_grad(set_to_none=True)

    def state():
        torch.cuda.synchronize()
        s = {f"table.{k}": v.clone() for k, v in emb.get_embedding_tables().items()}
        s.update({f"cross.{k}": v.detach().clone() for k, v in cross.state_dict().items()})
        s.update({f"emb.{k}": v.detach().clone() for k, v in emb.state_dict().items() if isinstance(v, torch.Tensor)})
        s.update({f"opt.{i}": st["sum"].clone() for i, st in enumerate(box[0].state.values())})
        return s

    return emb, step, state


def main(sharded: bool, rccl: bool = False, optimizer: str = "adagrad", bare: bool = False, fwd_only: bool = False):
    from keras_rs_amd.graphs import GraphedStep

    if rccl:
        import faulthandler
        import socket

        import torch.distributed as dist

        faulthandler.dump_traceback_later(150, exit=True)    # a capture that hangs must not take the session with it
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(port))
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    build = lambda: _build(sharded, rccl, optimizer)   # noqa: E731
    if fwd_only:
        # a forward pass captured WITHOUT its backward, gradients enabled: the layer forks its plan stream in forward and only
        # the backward joins it -- GraphedStep joins it before the capture ends
        import keras_rs_amd.layers as kl
        from keras_rs_amd.layers import base
        from keras_rs_amd.sharded import ShardedDistributedEmbedding

        B, D = 4096, 32
        tc = kl.TableConfig(name="t", vocabulary_size=5000, embedding_dim=D, initializer=base.RandomUniform(-0.05, 0.05, seed=7),
                            optimizer=kl.Adagrad(0.05, 0.1), combiner="sum", placement="sparsecore")
        emb = ShardedDistributedEmbedding({"f": kl.FeatureConfig("f", tc, (B, 3), (B, D))}, dtype="bfloat16", exchange="static")
        ids = {"f": torch.randint(0, 5000, (B, 3), device=DEV, dtype=torch.int32)}
        pre = emb.preprocess(ids)
        outs = []

        def fwd():
            assert torch.is_grad_enabled()
            out = emb(pre)["f"]
            outs[:] = [out.detach()]

        fwd()
        eager = outs[0].clone()
        graphed = GraphedStep(fwd, warmup=1)
        assert emb.plans_ahead >= 2 and len(graphed.joins) == 1
        graphed()
        torch.cuda.synchronize()
        assert torch.equal(outs[0], eager)
        print("GRAPH_FWD_ONLY_OK", flush=True)
        os._exit(0)
    if bare:
        # a capture that is NOT GraphedStep's has nobody to refresh the constants before a replay: refused, not replayed stale
        # (a process of its own: the refused capture is this process's only one)
        from keras_rs_amd import _lib as L

        _, step_c, _ = build()
        step_c()
        step_c()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                step_c()
        except Exception as e:   # noqa: BLE001 -- the capture's own teardown may wrap the layer's error
            chain, seen = [], e
            while seen is not None:
                chain.append(seen)
                seen = seen.__cause__ or seen.__context__
            assert any(isinstance(x, L.KrsError) and "GraphedStep" in str(x) for x in chain) or "GraphedStep" in str(e), repr(chain)
        else:
            raise AssertionError("a bare capture of a step with step-dependent optimizer constants must raise")
        print("GRAPH_REFUSED", optimizer, flush=True)
        os._exit(0)
    emb_a, step_a, state_a = build()
    for _ in range(5):
        step_a()
    ref = state_a()
    steps_of = lambda e: [g.step for g in (e._sgroups if sharded else e._groups["sparsecore"])]   # noqa: E731
    assert steps_of(emb_a) == [5]

    emb, step_b, state_b = build()
    graphed = GraphedStep(step_b, warmup=2)      # two eager steps, then the capture (which runs nothing)
    for _ in range(3):
        graphed()
    emb.check_ids(wait=True)                     # the flag word of the replays: no id was out of range
    if sharded:
        assert emb.poll_exchange_stats() is False and emb.overflow_steps == 0
        assert emb.last_exchange["need"][0] <= emb.last_exchange["capacity"][0]
    got = state_b()
    assert steps_of(emb) == [5], steps_of(emb)     # two eager updates + three replays: the count (checkpointed `iterations`) advanced
    assert ref.keys() == got.keys() and len(ref) >= 10
    for k in ref:
        assert torch.equal(ref[k], got[k]), k
    # ... and the steps did move the tables (the comparison is not between two untouched states)
    emb0, step0, state0 = build()
    step0()
    first = state0()
    assert not torch.equal(first["table.t0"], ref["table.t0"])
    print("GRAPH_OK", ("sharded_rccl" if rccl else "sharded") if sharded else "single", optimizer, flush=True)
    if rccl:
        # (destroy_process_group() waits forever while graphs that hold RCCL kernels are alive in this process, ROCm 7.2:
        #  leave without the teardown)
        os._exit(0)


if __name__ == "__main__":
    main(sys.argv[1].startswith("sharded"), sys.argv[1] == "sharded_rccl", sys.argv[2] if len(sys.argv) > 2 else "adagrad",
         len(sys.argv) > 3 and sys.argv[3] == "bare", len(sys.argv) > 3 and sys.argv[3] == "fwdonly")
