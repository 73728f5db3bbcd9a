"""Worker of tests/test_sharded_gpu.py: world_size-2 run of the sharded embedding with the REAL HIP
kernels, both ranks on the one GPU of the test box, collectives over gloo (staged through the host).
Checks outputs and updated tables against the single-GPU DistributedEmbedding fed the same data."""

import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def criteo_shaped(rank, world, kl, Sharded):
    """The 26 Criteo-1TB vocabularies (examples/ml_perf/configs/v6e_8.py:15-172) scaled down by 2000, the ml_perf
    bag lengths, embedding_threshold scaled alike: tables under it are REPLICATED (main.py:135-141), the rest
    MOD-sharded.  Checks: (1) outputs equal the single-GPU layer's, (2) the lookups every rank receives are within
    5 % of each other -- with the 3-, 4-, 10-row tables sharded they would not be --, (3) grad_average scales the
    table update by 1 / world, (4) replicated tables hand back dense gradients."""
    vocabs = [40000000, 39060, 17295, 7424, 20265, 3, 7122, 1543, 63, 40000000, 3067956, 405282, 10, 2209, 11938,
              155, 4, 976, 14, 40000000, 40000000, 40000000, 590152, 12973, 108, 36]
    hots = [3, 2, 1, 2, 6, 1, 1, 1, 1, 7, 3, 8, 1, 6, 9, 5, 1, 1, 1, 12, 100, 27, 10, 3, 1, 1]
    V = [max(3, v // 2000) for v in vocabs]
    threshold, D, B, lr = 11, 16, 64, 0.05

    def configs():
        tcs = [kl.TableConfig(f"cat_{t}", V[t], D, optimizer=kl.SGD(lr), combiner="sum", placement="sparsecore")
               for t in range(26)]
        return {f"cat_{t:02d}_id": kl.FeatureConfig(f"cat_{t}", tcs[t], (B, hots[t]), (B, D)) for t in range(26)}

    rng = np.random.default_rng(11)
    full = {f"cat_{t}": rng.uniform(-1, 1, (V[t], D)).astype(np.float32) for t in range(26)}
    ids = [{f"cat_{t:02d}_id": np.random.default_rng(1000 + 50 * r + t).integers(0, V[t], (B, hots[t])).astype(np.int32)
            for t in range(26)} for r in range(world)]
    g = [{k: np.random.default_rng(5000 + 50 * r + i).uniform(-1, 1, (B, D)).astype(np.float32)
          for i, k in enumerate(ids[r])} for r in range(world)]
    small = [f"cat_{t}" for t in range(26) if V[t] < threshold]
    assert len(small) >= 8
    results = {}
    for avg in (False, True):
        layer = Sharded(configs(), replicate_below=threshold, grad_average=avg)
        layer.build(None)
        layer.set_embedding_tables(full)
        out = layer(ids[rank])
        sum((o * torch.from_numpy(g[rank][k]).cuda()).sum() for k, o in out.items()).backward()
        torch.cuda.synchronize()
        recv = torch.tensor([float(sum(layer.last_exchange["recv_lookups"]))])
        all_recv = [torch.zeros(1) for _ in range(world)]
        dist.all_gather(all_recv, recv)
        loads = np.array([float(x) for x in all_recv])
        assert loads.max() / loads.mean() < 1.05 and loads.min() / loads.mean() > 0.95, loads
        rep = layer._replicated
        rep_grads = {tc.name: rep._table_params[id(tc)].grad.clone() for grp in rep._groups["default_device"]
                     for tc in grp.table_configs}
        assert sorted(rep_grads) == sorted(small)
        results[avg] = (out, {k: v.cpu().numpy() for k, v in layer.get_embedding_tables().items()}, rep_grads)

    ref = kl.DistributedEmbedding({k: kl.FeatureConfig(k, fc.table, (world * B, fc.input_shape[1]), (world * B, D))
                                   for k, fc in configs().items()})
    ref.build(None)
    ref.set_embedding_tables(full)
    cat = lambda xs, k: np.concatenate([x[k] for x in xs], 0)  # noqa: E731
    rout = ref({k: cat(ids, k) for k in ids[0]})
    sum((o * torch.from_numpy(cat(g, k)).cuda()).sum() for k, o in rout.items()).backward()
    ref_tables = {k: v.cpu().numpy() for k, v in ref.get_embedding_tables().items()}
    out, tables, rep_grads = results[False]
    for k in out:
        np.testing.assert_allclose(out[k].detach().cpu().numpy(), rout[k].detach().cpu().numpy()[rank * B:(rank + 1) * B],
                                   rtol=1e-5, atol=1e-5)
    for t in range(26):
        name = f"cat_{t}"
        if name in small:
            # replicated: untouched by the fused update; this rank's dense gradient = scatter-add of its own batch
            np.testing.assert_array_equal(tables[name], full[name])
            dense = np.zeros((V[t], D), np.float64)
            np.add.at(dense, ids[rank][f"cat_{t:02d}_id"].reshape(-1),
                      np.repeat(g[rank][f"cat_{t:02d}_id"], hots[t], axis=0))
            np.testing.assert_allclose(rep_grads[name].cpu().numpy(), dense, rtol=1e-5, atol=1e-5)
        else:
            np.testing.assert_allclose(tables[name], ref_tables[name], rtol=2e-5, atol=2e-6)
            # grad_average: the same update scaled by 1 / world (SGD is linear in the gradient)
            np.testing.assert_allclose(results[True][1][name] - full[name], (ref_tables[name] - full[name]) / world,
                                       rtol=1e-4, atol=2e-6)


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    import keras_rs_amd.layers as kl
    from keras_rs_amd.sharded import ShardedDistributedEmbedding

    kind = sys.argv[1]
    opt = {"sgd": kl.SGD(0.1), "adagrad": kl.Adagrad(0.1, 0.1), "adam": kl.Adam(0.1)}[kind]
    V, D, B = [501, 37, 1000], 32, 96
    hots = [1, 7, 3, 12]
    tix = [0, 1, 2, 0]
    combs = ["sum", "mean", "sqrtn"]

    def configs():
        tcs = [kl.TableConfig(f"t{i}", V[i], D, optimizer=opt, combiner=combs[i], placement="sparsecore") for i in range(3)]
        return {f"f{i}": kl.FeatureConfig(f"f{i}", tcs[tix[i]], (B, hots[i]), (B, D)) for i in range(4)}

    rng = np.random.default_rng(3)
    full = {f"t{i}": rng.uniform(-1, 1, (V[i], D)).astype(np.float32) for i in range(3)}
    # every rank's batch, known to all (the reference layer below consumes the concatenation)
    all_ids = [{f"f{i}": np.random.default_rng(50 + r * 10 + i).integers(0, V[tix[i]], (B, hots[i])).astype(np.int32)
                for i in range(4)} for r in range(world)]
    all_w = [{f"f{i}": np.random.default_rng(90 + r * 10 + i).uniform(0.1, 1, (B, hots[i])).astype(np.float32)
              for i in range(4)} for r in range(world)]
    all_g = [{f"f{i}": np.random.default_rng(130 + r * 10 + i).uniform(0, 1, (B, D)).astype(np.float32)
              for i in range(4)} for r in range(world)]

    exchange = sys.argv[2] if len(sys.argv) > 2 else "exact"
    prefetch = exchange == "static_prefetch"          # round 4: id exchange on the layer's stream ahead of the call,
    if prefetch:                                      # segment gradients gathered out of the slab gradient (lead = D)
        exchange = "static"
    layer = ShardedDistributedEmbedding(configs(), slab_lead_cols=D if prefetch else 8, exchange=exchange)
    layer.build(None)
    layer.set_embedding_tables(full)
    if prefetch:
        pre = layer.preprocess(all_ids[rank], all_w[rank])
        torch.cuda.synchronize()
        layer.prefetch(pre)
        # (work on the main stream while the exchange stream runs ahead)
        busy = torch.randn(2048, 2048, device="cuda") @ torch.randn(2048, 2048, device="cuda")
        out = layer(pre)
        assert layer.prefetch_hits == 1 and busy is not None
        head = torch.zeros((B, D), device="cuda", requires_grad=True)
        catd = kl.concat_features([head] + [out[f"f{i}"] for i in range(4)])
        gm = torch.cat([torch.zeros(B, D)] + [torch.from_numpy(all_g[rank][f"f{i}"]) for i in range(4)], dim=1).cuda()
        (catd * gm).sum().backward()
        assert layer.slab_grad_gathers == 1
    else:
        out = layer(all_ids[rank], all_w[rank])
        sum((o * torch.from_numpy(all_g[rank][k]).cuda()).sum() for k, o in out.items()).backward()
    torch.cuda.synchronize()
    got_tables = {k: v.cpu().numpy() for k, v in layer.get_embedding_tables().items()}

    # single-GPU layer on the concatenated batch
    ref = kl.DistributedEmbedding({k: kl.FeatureConfig(k, fc.table, (world * B, fc.input_shape[1]), (world * B, D))
                                   for k, fc in configs().items()})
    ref.build(None)
    ref.set_embedding_tables(full)
    cat = lambda xs, k: np.concatenate([x[k] for x in xs], 0)  # noqa: E731
    rout = ref({k: cat(all_ids, k) for k in out}, {k: cat(all_w, k) for k in out})
    sum((o * torch.from_numpy(cat(all_g, k)).cuda()).sum() for k, o in rout.items()).backward()
    for k in out:
        np.testing.assert_allclose(out[k].detach().cpu().numpy(), rout[k].detach().cpu().numpy()[rank * B:(rank + 1) * B],
                                   rtol=1e-5, atol=1e-5)
    for k, v in ref.get_embedding_tables().items():
        np.testing.assert_allclose(got_tables[k], v.cpu().numpy(), rtol=2e-5, atol=2e-6)
    if exchange == "static":
        assert layer.last_exchange["mode"] == "static" and layer.overflow_steps == 0
    if kind == "adagrad" and exchange == "exact":
        criteo_shaped(rank, world, kl, ShardedDistributedEmbedding)
    dist.barrier()
    if rank == 0:
        print("SHARDED_HIP_OK", kind)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
