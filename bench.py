#!/usr/bin/env python
"""bench.py -- the hot path's headline benchmark (see BASELINE.json / SURVEY.md section 8d).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one synthetic batch, through the drop-in layers
(keras_rs_amd.layers, HIP kernels behind the C ABI):

    embedding gather+pool of 26 tables (K1) -> DotInteraction over the 27 feature vectors (K4)
    -> 3 x FeatureCross(d = 3456, projection 512) (K3) -> loss
    -> backward of all of it, including the index-scatter embedding gradient with the fused
       Adagrad row update (K2) and an Adagrad step on the FeatureCross weights.

Workload C3 (DLRM-small): 26 tables x 1,000,000 rows x 128, batch 65,536 global, bf16 tables
and activations with fp32 accumulation, fp32 master weights for the dense kernels.  SURVEY.md
section 8d defines C3 with two bag-length lists; both are measured in one run: the primary one
(`value`, `ms_per_step`, `roofline`) uses the ml_perf DLRM lengths (sum L = 214, i.e. a real
gather+POOL; examples/ml_perf/configs/v6e_8.py), the L = 1 variant is reported under `also`
(`--hotness 1` swaps them).
With N > 1 the tables are MOD row-sharded over the ranks (C4), B_local = 65,536 / N.

The K timed steps run with Python's cyclic garbage collector switched off (reference counting still frees every
tensor): a full collection inside the region starves the GPU for a whole step on a fresh box.

The LAST stdout line (rank 0) is ONE JSON object below 4 KB (`compact_line`: the driver keeps an ~8 KB tail of stdout + stderr)
with `value` = whole-job embedding lookups/s, `ms_per_step` = the DCN fwd+bwd step time, a `roofline` object for K1 measured
live with HIP events on the launch stream (priced at its in-step duration, the isolated launch beside it), `roofline_dominant`
(the kernel family that costs the step the most time) and a `cpu_baseline` object (the reference's op composition on this
box's host cores on a bounded sample of the same workload).  Everything else the run measured -- `roofline_step` per kernel
family, the L = 1 and C2 legs, per-step times, the sharded run's `phases` / `graph_leg` / `parity` -- goes to the side file the
line names under `detail` (`--detail`, default ./bench_detail.json).
"""

from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
T_START = time.perf_counter()

ML_PERF_HOTS = [3, 2, 1, 2, 6, 1, 1, 1, 1, 7, 3, 8, 1, 6, 9, 5, 1, 1, 1, 12, 100, 27, 10, 3, 1, 1]
# Criteo-1TB vocabulary sizes of the 26 categorical features (data of examples/ml_perf/configs/v6e_8.py:15-172)
CRITEO_VOCABS = [40000000, 39060, 17295, 7424, 20265, 3, 7122, 1543, 63, 40000000, 3067956, 405282, 10, 2209, 11938,
                 155, 4, 976, 14, 40000000, 40000000, 40000000, 590152, 12973, 108, 36]
HBM_PEAK = 8.0e12  # MI355X_MICROARCH.md: 8 TB/s spec
MFMA_BF16_PEAK = 2.5e15  # MI355X_MICROARCH.md: dense bf16 MFMA peak (no sparsity)
MFMA_F32_PEAK = 157.3e12  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 (fp32 in / fp32 acc) = the fp32 vector rate


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--tables", type=int, default=26)
    ap.add_argument("--vocab", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--batch", type=int, default=65536, help="global batch")
    ap.add_argument("--projection", type=int, default=512)
    ap.add_argument("--cross-layers", type=int, default=3)
    ap.add_argument("--hotness", choices=["mlperf", "1"], default="mlperf",
                    help="primary bag lengths: the ml_perf list (sum L = 214, default) or L = 1; the other one is reported under `also`")
    ap.add_argument("--criteo-vocab", type=int, default=0, metavar="CAP",
                    help="per-table vocabularies = min(Criteo-1TB size, CAP) instead of --vocab everywhere: CAP 1000000 "
                         "is SURVEY.md's C3', CAP 40000000 is C5 (204 M rows: 52 GB of bf16 tables + 105 GB of fp32 "
                         "Adagrad accumulators on one GPU); needs --tables 26; no cpu_baseline leg")
    ap.add_argument("--id-skew", type=float, default=0.0, metavar="E",
                    help="power-law ids: id = perm(floor(V * u^E)) with a fixed affine permutation of the rows "
                         "(E = 4: 1 %% of the rows draw 32 %% of the lookups); 0 = uniform ids")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-sharded", action="store_true",
                    help="run the MOD-sharded embedding path even at N = 1 (dry run of the multi-GPU code)")
    ap.add_argument("--cpu-sample-batch", type=int, default=2048)
    ap.add_argument("--rowwise-adagrad", action="store_true",
                    help="table optimizer = layers.RowwiseAdagrad (opt-in variant with one accumulator per row; the "
                         "default line uses the reference's exact Adagrad)")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                    help="torch.distributed backend for --gpus > 1 (nccl = RCCL; gloo stages through the host and lets "
                         "several ranks share one GPU: a functional rig, not a measurement)")
    ap.add_argument("--exchange", choices=["static", "exact"], default="static",
                    help="sharded runs: 'static' = fixed-capacity blocks, no host wait in the step (the reference's "
                         "max_ids_per_partition contract); 'exact' = data-dependent sizes through one host wait")
    ap.add_argument("--probe-steps", type=int, default=3,
                    help="extra steps after each timed region with event spans around the C-ABI calls (K2 apply, GEMMs): "
                         "the `roofline_step` entries; 0 = off")
    ap.add_argument("--sustained-steps", type=int, default=500,
                    help="back-to-back steps run BEHIND the K timed steps (HIP events at every step boundary): the "
                         "power-capped steady state, reported under `sustained`; 0 = off")
    ap.add_argument("--no-c2", action="store_true",
                    help="skip the `also_c2` leg (BASELINE.json configs[1]: 8 tables x 100k x 64 fp32, 3 full-rank FeatureCross, "
                         "batch 8192, fp32)")
    ap.add_argument("--c2-leg", action="store_true",
                    help="(internal) run ONLY the C2 leg and print its object: the default run starts this in a process of its "
                         "own, so that the leg's graph capture cannot take the headline line with it")
    ap.add_argument("--replicate-below", type=int, default=0, metavar="ROWS",
                    help="sharded runs: tables with fewer rows are REPLICATED on every rank instead of sharded (dense gradients, "
                         "joined to the dense all-reduce) -- what the reference's model does below its embedding_threshold "
                         "(examples/ml_perf/main.py:135-141); the Criteo vocabularies hold tables of 3 ... 155 rows that would "
                         "send all their lookups to a few owners")
    ap.add_argument("--virtual-world", type=int, default=0, metavar="N",
                    help="with --force-sharded on ONE GPU: run rank 0's step of an N-way job with the shapes of N ranks (1/N shard, "
                         "batch / N samples, ids routed to N owners, N blocks per exchange, N partials per bag) and device copies "
                         "for the links -- the per-rank KERNEL time of an N-way job (ShardedDistributedEmbedding(virtual_world=N)); "
                         "`value` is then a projection that leaves the link time out, and the line says so")
    ap.add_argument("--no-parity", action="store_true",
                    help="sharded runs: skip the self-check of one step against an unsharded recompute of a slice (`parity`)")
    ap.add_argument("--no-graph-leg", action="store_true",
                    help="sharded runs over RCCL: do not try the graph-replayed step behind the eager timed region")
    ap.add_argument("--graph", action="store_true",
                    help="one rank only: capture the step in a HIP graph (keras_rs_amd.graphs.GraphedStep) and time its "
                         "replays -- for the host-bound per-rank step of a strongly-scaled job (--force-sharded --batch 8192)")
    ap.add_argument("--capacity-settle", type=int, default=0, metavar="STEPS",
                    help="sharded runs, static exchange: shrink the block capacities to the running statistics once STEPS steps "
                         "in a row fit (the layer's default is 16; 0 = keep the first sizing for the whole run)")
    ap.add_argument("--prefetch", action="store_true",
                    help="sharded runs: the id side of the next step's lookup (route -> id all-to-all -> unpack) runs ahead "
                         "on the layer's exchange stream (ShardedDistributedEmbedding.prefetch)")
    ap.add_argument("--no-prefetch", action="store_true", help="(kept for old command lines: prefetch is opt-in)")
    ap.add_argument("--rccl-self", action="store_true",
                    help="with --force-sharded at N = 1: route the layer's collectives through a ONE-rank RCCL communicator "
                         "instead of device copies (proves the RCCL call path on a one-GPU box)")
    ap.add_argument("--full-model", action="store_true",
                    help="also time one training step of the whole DLRM-DCN-v2 model (examples/dlrm_dcn_v2.py: bottom "
                         "MLP, embeddings, 3 cross layers, top MLP, BCE), reported under `full_model`; never `value`")
    ap.add_argument("--host-inputs", type=int, default=0, metavar="WORKERS",
                    help="also time the step with ids that start in HOST memory, fed through "
                         "keras_rs_amd.data.ThreadedDataLoader with this many loader threads (PCIe-inclusive rate, "
                         "reported under `host_inputs`; never `value`)")
    ap.add_argument("--detail", default="bench_detail.json", metavar="PATH",
                    help="side file with everything the run measured (roofline_step, also, also_c2, phases, graph_leg, per-step "
                         "times); the last stdout line is a summary below %d bytes that names it" % 4096)
    ap.add_argument("--print-detail", action="store_true",
                    help="also print the side file's content as an earlier stdout line (KRS_BENCH_DETAIL ...)")
    ap.add_argument("--cpu-pools", default="16", metavar="LIST",
                    help="intra-op pool sizes of the torch-CPU leg of cpu_baseline, comma separated, each capped at the host's "
                         "thread count (BASELINE.md section 4 records the 256- and 64-thread pools as 130x / 2.4x SLOWER than 16)")
    a = ap.parse_args()
    if a.criteo_vocab:
        if a.tables != len(CRITEO_VOCABS):
            ap.error("--criteo-vocab needs --tables 26")
        a.vocabs = [min(v, a.criteo_vocab) for v in CRITEO_VOCABS]
    else:
        a.vocabs = [a.vocab] * a.tables
    return a


def self_launch(n):
    """`python bench.py --gpus N` with N > 1 and no launcher around it (RANK unset): start the N ranks ourselves,
    one process per GPU, as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N` on 127.0.0.1 with a
    free port (the reference's entry point enumerates its devices itself too: examples/ml_perf/main.py:117-119).
    Returns the launcher's exit code; rank 0's JSON line goes to this process's stdout unchanged."""
    import subprocess

    port = _free_port()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this driver (RCCL peer buffers)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(n, 1))))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def dist_setup(n, backend="nccl", single_rank_group=False):
    """(rank, world, local device, backend actually used)."""
    import torch.distributed as dist

    if n <= 1:
        if single_rank_group:
            # --force-sharded dry run: a ONE-rank RCCL communicator, so that the layer's collectives
            # (all_to_all_single with split sizes, all_reduce) really go through RCCL on this GPU
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
            torch.cuda.set_device(0)
            dist.init_process_group(backend, rank=0, world_size=1,
                                    **({"device_id": torch.device("cuda", 0)} if backend == "nccl" else {}))
            return 0, 1, 0, backend
        return 0, 1, 0, None
    rank = int(os.environ["RANK"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    world = int(os.environ["WORLD_SIZE"])
    n_dev = torch.cuda.device_count()
    if backend == "nccl" and world > n_dev:
        # RCCL refuses two ranks on one GPU: fall back to the functional rig (ranks share the GPUs, collectives
        # over gloo staged through the host) and say so in the line -- not a measurement of the links
        backend = "gloo"
    if backend == "gloo":
        local = local % n_dev
        torch.cuda.set_device(local)
        dist.init_process_group("gloo")
    else:
        torch.cuda.set_device(local)     # bind LOCAL_RANK's device before any allocation
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    return rank, world, local, backend


def _free_port():
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


class Model(torch.nn.Module):
    """Embedding -> [DotInteraction] -> DCN cross stack, from the drop-in layers."""

    def __init__(self, a, hots, world, rank):
        super().__init__()
        import keras_rs_amd.layers as kl
        from keras_rs_amd.layers import base

        self.a, self.hots = a, hots
        self.concat = kl.concat_features
        opt = kl.Adagrad(learning_rate=0.0034, initial_accumulator_value=0.1)  # configs/v6e_8.py lr
        if a.rowwise_adagrad:   # opt-in variant, NOT the reference's optimizer (one accumulator per row)
            opt = kl.RowwiseAdagrad(learning_rate=0.0034, initial_accumulator_value=0.1)
        feats = {}
        for t in range(a.tables):
            # the 40 M-row tables of C5 are drawn on the device, they never exist in host memory
            init = base.RandomUniform(-0.05, 0.05, seed=1337 + t, device_rng=bool(a.criteo_vocab))
            tc = kl.TableConfig(name=f"cat_{t}", vocabulary_size=a.vocabs[t], embedding_dim=a.dim,
                                initializer=init, optimizer=opt,
                                combiner="sum", placement="sparsecore")
            feats[f"cat_{t:02d}_id"] = kl.FeatureConfig(f"cat_{t}", tc, (a.batch // world, hots[t]),
                                                       (a.batch // world, a.dim))
        fp32 = bool(getattr(a, "fp32", False))       # the C2 leg: fp32 tables, activations and weights
        self.lead = 0 if getattr(a, "no_dense", False) else a.dim
        emb_policy, cross_policy = ("float32", "float32") if fp32 else ("bfloat16", "mixed_bfloat16")
        if world > 1 or a.force_sharded:
            from keras_rs_amd.sharded import ShardedDistributedEmbedding

            # (capacity_settle_steps: the blocks of the static exchange shrink to the settled statistics after that many
            #  fitting steps -- a resize, i.e. fresh buffers, in the middle of a 20-step timed region; off unless asked for)
            self.embedding = ShardedDistributedEmbedding(feats, dtype=emb_policy, slab_lead_cols=self.lead,
                                                         exchange=a.exchange, capacity_settle_steps=a.capacity_settle,
                                                         replicate_below=getattr(a, "replicate_below", 0),
                                                         virtual_world=getattr(a, "virtual_world", 0))
            self.embedding._collectives_at_world1 = bool(a.rccl_self)
        else:
            # the dense feature's 128 columns are reserved in front of the 26 embeddings: the lookups land
            # directly in the [B, 3456] interaction input (concat-free layout)
            self.embedding = kl.DistributedEmbedding(feats, dtype=emb_policy, name="embedding_layer",
                                                     slab_lead_cols=self.lead)
        self.dot = None if getattr(a, "no_dot", False) else kl.DotInteraction(dtype=emb_policy)
        self.cross = torch.nn.ModuleList(
            kl.FeatureCross(projection_dim=a.projection, kernel_initializer=base.GlorotUniform(seed=1337 + i),
                            dtype=cross_policy) for i in range(a.cross_layers))

    def forward(self, dense_out, pre):
        emb = self.embedding(pre)
        feats = ([dense_out] if self.lead else []) + [emb[k] for k in emb]
        inter = self.dot(feats) if self.dot is not None else None     # [B, 351]
        x0 = self.concat(feats)                                   # [B, 3456]  (model.py:204-207)
        xl = x0
        for layer in self.cross:                                  # DCNBlock.call, model.py:332-336
            xl = layer(x0, xl)
        return xl, inter


def make_inputs(a, hots, b_local, rank, dev):
    g = torch.Generator(device=dev).manual_seed(1338 + rank)
    if a.id_skew > 0:
        ids = {}
        for t, v in enumerate(a.vocabs):
            u = torch.rand(b_local, hots[t], device=dev, generator=g, dtype=torch.float64)
            r = (u.pow(a.id_skew) * v).long().clamp_(max=v - 1)
            mult = next(m for m in (7368787, 7368791, 7368793, 7368799, 7368803) if math.gcd(m, v) == 1)
            ids[f"cat_{t:02d}_id"] = ((r * mult + 12345) % v).to(torch.int32)
    else:
        ids = {f"cat_{t:02d}_id": torch.randint(0, a.vocabs[t], (b_local, hots[t]), device=dev, generator=g,
                                                dtype=torch.int32) for t in range(a.tables)}
    dense = (torch.rand(b_local, a.dim, device=dev, generator=g) * 0.9).to(
        torch.float32 if getattr(a, "fp32", False) else torch.bfloat16)
    return ids, dense


_BLOCK = {}


def gpu_block(dev, ms):
    """Keeps the GPU busy for about `ms` milliseconds (copies of a 1 GiB buffer, ~0.4 ms each)."""
    if "buf" not in _BLOCK:
        _BLOCK["buf"] = torch.empty(2, 1 << 29, dtype=torch.uint8, device=dev)
    b = _BLOCK["buf"]
    for _ in range(max(1, int(ms / 0.2))):
        b[1].copy_(b[0])


def k1_bytes(nnz, bags, dim, es):
    # nnz*(D*s_t + i) + bags*(D*s_o + 4)
    return nnz * (dim * es + 4) + bags * (dim * es + 4)


def _median_time(fn, warm, reps, budget_s=6.0):
    """Median / minimum / count of up to `reps` timed runs after `warm` warm-ups; stops early (after at least one
    timed run) once `budget_s` seconds have been spent, so that a slow host cannot stretch the bench run."""
    t_start = time.perf_counter()
    for _ in range(warm):
        fn()
        if time.perf_counter() - t_start > budget_s / 2:
            break
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
        if time.perf_counter() - t_start > budget_s:
            break
    return float(np.median(ts)), float(np.min(ts)), len(ts)


def cpu_baseline(a, hots):
    """The hot path on the host cores, on a bounded sample of the same workload (batch `cpu_sample_batch` of the
    same 26 tables x 1M rows x 128 bf16, same bag lengths, same cross stack), two implementations
    (BASELINE.md section 3):
      * `port`: the C oracle (oracle/krs_oracle.c, OpenMP over all host threads) -- gather+pool, then one
        FeatureCross layer forward + backward (6 GEMMs + the elementwise pass) x cross_layers;
      * `torch_cpu`: how Keras-on-CPU composes the reference -- embedding_bag (ops.take + sum) with sparse
        gradients, matmul + bias + elementwise, autograd backward -- fp32, at all threads and at one thread.
    3 warm-ups + 10 timed runs, medians (the one-thread leg: a quarter of the sample, 1 + 3 runs).  Neither leg
    updates the tables (the reference's dense [V, D] gradient + dense optimizer pass over all 26 M rows per step
    would come on top)."""
    from oracle import krs_oracle as ko

    b = a.cpu_sample_batch
    vocab = a.vocab
    rng = np.random.default_rng(1337)
    one = ko.f32_to_bf16_bits(rng.uniform(-0.05, 0.05, (vocab, a.dim)).astype(np.float32))
    tables = [one] + [one.copy() for _ in range(a.tables - 1)]        # distinct memory, same statistics
    ids_list = [rng.integers(0, vocab, b * h).astype(np.int32) for h in hots]
    ids = np.concatenate(ids_list)
    tabs = ko.make_tables(tables)
    feats = ko.make_features(list(range(a.tables)), ["sum"] * a.tables, [t * a.dim for t in range(a.tables)],
                             hots=hots, batch=b)
    out = np.zeros((b, a.tables * a.dim), np.uint16)
    t_emb, _, n_emb = _median_time(lambda: ko.embed_bag_fwd_raw(tabs, ko.BF16, feats, ids, None, None, b, a.dim, out), 3, 10, 4.0)
    d, p = (a.tables + 1) * a.dim, a.projection
    bits = lambda lo, hi, shape: ko.f32_to_bf16_bits(rng.uniform(lo, hi, shape).astype(np.float32))  # noqa: E731
    bc = min(b, 512)     # rows of the cross-layer leg (the oracle's GEMM is a plain triple loop): scaled to b below
    x0, x, g = bits(-1, 1, (bc, d)), bits(-1, 1, (bc, d)), bits(-1, 1, (bc, d))
    u_, v_ = bits(-0.03, 0.03, (d, p)), bits(-0.03, 0.03, (p, d))

    def cross_layer():
        h, _ = ko.gemm(x, u_, bc, p, d)
        y, uo = ko.gemm(h, v_, bc, d, p, x0=x0, x=x, want_u=True)
        dz, dx0, _, _ = ko.cross_epilogue_bwd(g, uo, x0, x)
        ko.gemm(h, dz, p, d, bc, a_is_km=True, out_dtype=ko.F32)
        dh, _ = ko.gemm(dz, v_, bc, p, d, b_is_nk=True)
        ko.gemm(x, dh, d, p, bc, a_is_km=True, out_dtype=ko.F32)
        ko.gemm(dh, u_, bc, d, p, b_is_nk=True, r=g)

    t_layer, _, n_layer = _median_time(cross_layer, 1, 3, 8.0)
    t_layer *= b / bc
    t_port = t_emb + t_layer * a.cross_layers
    lookups = b * sum(hots)

    # ---- torch on the CPU: the composition Keras would run (fp32) ----
    tt = [torch.from_numpy(ko.bf16_bits_to_f32(tables[0]))] + [None] * (a.tables - 1)
    for t in range(1, a.tables):
        tt[t] = tt[0].clone()
    tt = [t.requires_grad_() for t in tt]
    del tables, tabs

    def torch_leg(threads, bb, warm, reps):
        torch.set_num_threads(threads)
        tid = [torch.from_numpy(ids_list[t][: bb * hots[t]].astype(np.int64)) for t in range(a.tables)]
        offs = [torch.arange(0, bb * hots[t], hots[t]) for t in range(a.tables)]
        dense = torch.rand(bb, a.dim)
        U = [torch.randn(d, p).mul_(0.03).requires_grad_() for _ in range(a.cross_layers)]
        V = [torch.randn(p, d).mul_(0.03).requires_grad_() for _ in range(a.cross_layers)]
        bias = [torch.zeros(d, requires_grad=True) for _ in range(a.cross_layers)]
        gy = torch.rand(bb, d)
        emb_t = [0.0]

        def step():
            t0 = time.perf_counter()
            embs = [torch.nn.functional.embedding_bag(tid[t], tt[t], offs[t], mode="sum", sparse=True)
                    for t in range(a.tables)]
            emb_t[0] = time.perf_counter() - t0
            x0_ = torch.cat([dense] + embs, dim=1)
            xl = x0_
            for i in range(a.cross_layers):
                xl = x0_ * ((xl @ U[i]) @ V[i] + bias[i]) + xl
            xl.backward(gy)
            for w in tt + U + V + bias:
                w.grad = None

        embs = []

        def step_and_note():
            step()
            embs.append(emb_t[0])

        med, _, n = _median_time(step_and_note, warm, reps, 8.0)
        return {"threads": threads, "batch": bb, "ms_per_step": med * 1e3, "value": bb * sum(hots) / med,
                "unit": "lookups/s", "embed_fwd_lookups_per_s": bb * sum(hots) / float(np.median(embs[-n:])),
                "timed_runs": n}

    n_thr = os.cpu_count() or 1
    # "all threads": torch's intra-op pool at every hardware thread is far from its best on a many-core host (a
    # 256-thread box measured 75x SLOWER per lookup than one thread: synchronisation, not work), so the pool size is
    # swept and the fastest is the one reported as the all-threads leg; every size tried is listed
    pools = sorted({max(1, min(n_thr, int(v))) for v in str(getattr(a, "cpu_pools", "16")).split(",") if v.strip()}, reverse=True)
    tried = [torch_leg(t, b, 2, 10) for t in pools]
    torch_all = max(tried, key=lambda r: r["value"])
    torch_one = torch_leg(1, max(b // 4, 64), 1, 3)
    torch.set_num_threads(n_thr)
    port = {
        "value": lookups / t_port, "unit": "lookups/s", "threads": n_thr,
        "embed_fwd_lookups_per_s": lookups / t_emb,
        "ms_per_step_on_sample": t_port * 1e3,
        "sample": f"oracle/krs_oracle.c (the parity checker: plain loops, OpenMP on {n_thr} threads, not tuned) -- "
                  f"gather+pool (3 warm-ups + {n_emb} runs, median) + one cross layer forward+backward "
                  f"on {bc} rows scaled to the sample (1 + {n_layer} runs, median) x {a.cross_layers}; no table update",
    }
    # `value` = the BEST host composition of the path (the number a reader compares with): the reference's op
    # composition on torch's CPU kernels at the best intra-op pool size; the C oracle is reported beside it
    return {
        "value": torch_all["value"], "unit": "lookups/s", "cores": torch_all["threads"], "kind": "port",
        "implementation": "the reference's composition (embedding_bag with sparse gradients -> matmul + bias + "
                          "elementwise cross stack -> autograd backward) on torch's CPU kernels, fp32, no table update",
        "ms_per_step_on_sample": torch_all["ms_per_step"],
        "embed_fwd_lookups_per_s": torch_all["embed_fwd_lookups_per_s"],
        "sample": f"batch {b} of the C3 workload ({a.tables} tables x {vocab} rows x {a.dim}, sum L = "
                  f"{sum(hots)}, {a.cross_layers} x FeatureCross(d={d}, p={p})), 2 warm-ups + {torch_all['timed_runs']} "
                  f"timed steps (median), intra-op pool sizes tried: {[r['threads'] for r in tried]} of {n_thr} hardware threads",
        "host_threads": n_thr,
        "torch_cpu": {"all_threads": torch_all, "one_thread": torch_one, "pool_sizes_tried": tried},
        "oracle_port": port,
    }


def measure(model, a, hots, world, rank, dev, b_local, steps, warmup, opt_box, loader=None, probe_steps=0,
            sustained_steps=0):
    """Times `steps` steps of the hot path for one bag-length list.  Returns a dict: `elapsed` (wall seconds of the
    timed region, max over ranks), `k1_s` (one K1 launch, events), `step_ms` (per-step GPU time of the timed steps,
    from events recorded at the step boundaries) and, with `probe_steps`, `probe` (in-step kernel spans of that many
    EXTRA steps run after the timed region; keras_rs_amd/probe.py).
    With `loader`, every step takes its preprocessed ids from it (host-resident inputs)."""
    from keras_rs_amd import probe as krs_probe

    ids, dense = make_inputs(a, hots, b_local, rank, dev)
    pre = model.embedding.preprocess(ids)
    sharded_run = world > 1 or a.force_sharded
    adt = torch.float32 if getattr(a, "fp32", False) else torch.bfloat16     # activation / table dtype of the leg
    n_slots = a.tables + (0 if getattr(a, "no_dense", False) else 1)
    scale = 1.0 / (b_local * n_slots * a.dim)
    k1_ev = []
    g_xl = torch.full((b_local, n_slots * a.dim), scale, dtype=adt, device=dev)
    n_inter = n_slots * (n_slots - 1) // 2
    g_inter = None if getattr(a, "no_dot", False) else torch.full((b_local, n_inter), 0.1 * scale, dtype=adt, device=dev)
    dp = world > 1 or (a.force_sharded and a.rccl_self)   # dense gradients go through the all-reduce

    # sharded + static exchange: the id side of the NEXT step's lookup (route -> id all-to-all -> unpack) runs on the
    # layer's exchange stream under this step's backward pass (ShardedDistributedEmbedding.prefetch)
    # (opt-in, --prefetch.  On ONE GPU the per-rank step is bound by the host's enqueue rate and prefetch only adds stream
    #  bookkeeping: through the one-rank RCCL communicator it measured 3.00 -> 2.81 ms in one call and 2.72 -> 2.98 in
    #  another (profiles/r4e_* / r4z_sharded_b8192_rccl_one_rank*.json: inside the run-to-run spread of a host-bound step).
    #  Between real ranks it takes the id all-to-all's link time off the critical path, but RCCL between ranks has never
    #  run in this environment: the first hardware run of `--gpus N` stays on the plain, fully tested order of collectives)
    prefetch = (sharded_run and loader is None and not getattr(a, "graph", False) and a.prefetch
                and getattr(model.embedding, "exchange", None) == "static")

    def step():
        xl, inter = model(dense, pre if loader is None else next(loader))
        if prefetch:
            model.embedding.prefetch(pre)
        # loss = scale * sum(xl) + 0.1 * scale * sum(inter), taken through its (constant) output
        # gradients: the reduction to a scalar is not part of the hot path
        if inter is None:
            torch.autograd.backward([xl], [g_xl])
        else:
            torch.autograd.backward([xl, inter], [g_xl, g_inter])
        if opt_box[0] is None:  # the first step has built the cross layers
            params = [p for layer in model.cross for p in layer.parameters()]
            from keras_rs_amd.optim import Adagrad   # torch.optim.Adagrad's arithmetic, one launch (krs_dense_adagrad)

            opt_box[0] = Adagrad(params, lr=0.0034, initial_accumulator_value=0.1, prepare_casts=True)
            if dp:
                # dense weights are data-parallel: from the next backward on, each gradient's all-reduce starts
                # the moment autograd has produced it and overlaps the rest of the backward pass
                from keras_rs_amd.dp import GradAllReduce

                # (same communicator as the embedding's all-to-alls: RCCL runs them in issue order, no second
                # communicator whose kernels could interleave differently on different ranks)
                opt_box.append(GradAllReduce(params, run_at_world1=True))
                for p in params:
                    opt_box[1].launch(p)
        opt = opt_box[0]
        if dp:
            with krs_probe.span("allreduce_wait"):
                opt_box[1].wait()
        opt.step()
        opt.zero_grad(set_to_none=True)

    for _ in range(max(warmup - 1, 0)):     # (the last warm-up step runs behind the collection below)
        step()
    eager_step = step
    if getattr(a, "graph", False) and not getattr(a, "_graph_used", False):
        a._graph_used = True      # (the primary leg only: see keras_rs_amd/graphs.py on a second capture in one process)
        if loader is not None:
            raise SystemExit("--graph: device-resident inputs only")
        # (collectives inside the capture -- N > 1 or --rccl-self: RCCL's kernels and torch's stream hand-offs are
        #  captured with the step; every rank captures and replays the same graph)
        from keras_rs_amd.graphs import GraphedStep

        step = GraphedStep(eager_step, warmup=2)
    sharded = world > 1 or a.force_sharded
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    # The timed region runs without the cyclic garbage collector: a full collection of a process that has imported
    # torch walks ~10^6 objects (tens of milliseconds) and, when it lands inside the K steps, starves the GPU for a
    # whole step -- the first run of this file on a fresh box did that once per run (one 55 ms step among 10.4 ms ones).
    # Reference counting frees the step's tensors as before; the collector is switched back on behind the region.
    import gc

    gc.collect()
    gc.disable()
    if warmup > 0:
        # the LAST of the W warm-up steps: a full collection walks ~10^6 objects and leaves the interpreter's caches cold;
        # timed right behind it, the first step enqueued slowly enough to starve the device (11.7 ms against 10.1)
        step()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks[0].record()
    dbg = [] if os.environ.get("KRS_BENCH_DEBUG") else None
    for i in range(steps):
        if dbg is not None:
            th = time.perf_counter()
        step()
        marks[i + 1].record()     # step boundaries on the launch stream: per-step GPU time, no host wait
        if dbg is not None:
            ms = torch.cuda.memory_stats(dev)
            dbg.append((round((time.perf_counter() - th) * 1e3, 2), ms.get("reserved_bytes.all.current", 0) >> 20,
                        ms.get("num_device_alloc", 0), ms.get("num_device_free", 0), ms.get("num_alloc_retries", 0)))
    enqueue_s = time.perf_counter() - t0     # the host has ENQUEUED every step; the device may still be running
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if dbg is not None and rank == 0:
        print("KRS_BENCH_DEBUG per step (host ms, reserved MiB, device allocs, frees, retries):", dbg, file=sys.stderr)
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(steps)]
    sustained = None
    if sustained_steps > 0:
        # `sustained`: the same step, back to back for seconds instead of K = 20 steps -- the chip clocks to its power
        # budget (MI355X_MICROARCH.md, DVFS; the ring GEMMs run 12 % faster on all-zero operands, profiles/
        # r4_gemm_layout_waves_clock_probe.txt), and 0.2 s of timed region says nothing about that steady state.  HIP events at
        # every step boundary, no host wait inside; collector off as in the timed region.
        sm = [torch.cuda.Event(enable_timing=True) for _ in range(sustained_steps + 1)]
        gc.collect()
        gc.disable()
        step()
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        ts = time.perf_counter()
        sm[0].record()
        for i in range(sustained_steps):
            step()
            sm[i + 1].record()
        torch.cuda.synchronize()
        wall = time.perf_counter() - ts
        gc.enable()
        each = [sm[i].elapsed_time(sm[i + 1]) for i in range(sustained_steps)]
        q = max(1, min(50, sustained_steps // 4))
        sustained = {"steps": sustained_steps, "ms_per_step": wall / sustained_steps * 1e3, "median_ms": float(np.median(each)),
                     "first_%d_median_ms" % q: float(np.median(each[:q])), "last_%d_median_ms" % q: float(np.median(each[-q:])),
                     "max_ms": float(np.max(each)), "seconds": wall,
                     "vs_timed_region": (wall / sustained_steps) / (elapsed / steps),
                     "source": "wall clock / steps for ms_per_step, HIP events at the step boundaries for the medians; run "
                               "behind the K timed steps, before the K1 probe and the CPU baseline"}
    # K1 launch duration, measured live with events on the launch stream, BEHIND the timed steps (one krs_embed_bag_fwd
    # launch per event pair).  It used to sit between the warm-up and the timed steps: its blocker copies left the first
    # timed steps 0.3-1.8 ms slow (11.99, 10.46, then 10.1-10.2 ms in `step_stats.each_ms`), i.e. the warm-up was undone.
    k1_s = None
    if sharded:
        # sharded run: the embedding call contains the all-to-alls, so K1 is timed on its own in the form
        # the owner side runs it (row gather of this rank's share of the lookups from its shard)
        emb = model.embedding
        nloc = b_local * sum(hots)
        rows = torch.randint(0, emb.shard.shape[0], (nloc,), device=dev, dtype=torch.int32)
        call = lambda: emb.kernels.gather_rows(emb.shard.data, rows)   # noqa: E731
    else:
        # One K1 launch per event pair, through the thin op wrapper (krs_embed_bag_fwd on the layer's own
        # tables / descriptors, same slab shape as the layer call).  A blocker kernel is queued first, so
        # the event pairs and launches are all enqueued while the GPU is still busy: the interval between
        # two events is then the kernel alone, not the host's launch overhead (which a profiler inflates).
        group = model.embedding._groups["sparsecore"][0]
        fi = pre["preprocessed_inputs_per_placement"]["sparsecore"]["inputs"]["group0"]
        n = len(group.bags.features)
        lead = model.embedding.slab_lead_cols
        slab = torch.empty((b_local, lead + n * a.dim), dtype=adt, device=dev)
        call = lambda: group.bags.forward(fi["ids"], b_local, hots=fi["hots"], offsets=fi["offsets"],   # noqa: E731
                                          out=slab[:, lead:])
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    for _ in range(4):
        gpu_block(dev, 3.0)  # launches are enqueued behind a busy GPU: events bracket the kernel alone
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            call()
            e1.record()
            k1_ev.append((e0, e1))
        torch.cuda.synchronize()
    k1_s = float(np.median([e0.elapsed_time(e1) for e0, e1 in k1_ev])) * 1e-3
    res = {"elapsed": elapsed, "k1_s": k1_s, "enqueue_s": enqueue_s, "step_ms": step_ms, "sustained": sustained}
    if probe_steps > 0:
        from keras_rs_amd import autograd as krs_autograd
        from keras_rs_amd import probe

        # (the probe steps run the weight-gradient GEMMs on the main stream: beside the elementwise passes, as in the
        #  timed steps, their event spans would measure the overlap, not the kernels)
        side_was, krs_autograd.WGRAD_SIDE_STREAM = krs_autograd.WGRAD_SIDE_STREAM, False
        step = eager_step     # (spans are host-side event pairs: eager steps)
        step()
        probe.start()
        for _ in range(probe_steps):
            step()
        res["probe"] = probe.stop()
        res["probe_steps"] = probe_steps
        krs_autograd.WGRAD_SIDE_STREAM = side_was
        if not sharded:
            # unique touched rows (K2's algorithmic bytes need them): measurement bookkeeping, outside every timed region
            uniq = 0
            for t, v in enumerate(ids.values()):      # a touched-row mask per table (no sort: keeps foreign kernels out of the traces)
                mask = torch.zeros(a.vocabs[t], dtype=torch.bool, device=dev)
                mask[v.reshape(-1).long()] = True
                uniq += int(mask.sum())
            res["unique_rows"] = uniq
    if probe_steps > 0 and sharded:
        res["phases"] = exchange_phases(res["probe"], probe_steps, world)
    flush = getattr(model.embedding, "flush_exchange_stats", None)
    res["overflow_steps"] = int(flush()) if flush is not None else 0
    ex = getattr(model.embedding, "last_exchange", None)
    if ex:
        res["exchange"] = dict(ex)
    return res


def measure_c2(a, dev):
    """BASELINE.json configs[1] / SURVEY.md section 8d C2 on the same GPU, timed behind the headline legs: 8 tables
    [100000, 64] fp32, ids [8192] per table (L = 1, uniform), d = 8 * 64 = 512, three FULL-RANK FeatureCross layers
    (glorot kernels, zero bias), fp32 everywhere (tables, activations, weights: `v_mfma_f32_32x32x2_f32`, exact fmaf chain),
    fused Adagrad on the tables, Adagrad on the kernels.  Same step function, same probes; its GEMMs are priced against
    the fp32 MFMA peak (157.3 TF/s)."""
    import copy

    c = copy.copy(a)
    c.tables, c.vocab, c.dim, c.batch, c.projection, c.cross_layers = 8, 100_000, 64, 8192, None, 3
    c.vocabs, c.criteo_vocab, c.id_skew, c.rowwise_adagrad, c.force_sharded, c.graph = [100_000] * 8, 0, 0.0, False, False, False
    c.fp32, c.no_dot, c.no_dense = True, True, True
    hots = [1] * c.tables
    model = Model(c, hots, 1, 0)
    model.embedding.build(None)
    steps, warm = 100, 10
    opt_box = [None]
    r = measure(model, c, hots, 1, 0, dev, c.batch, steps, warm, opt_box, probe_steps=3, sustained_steps=0)
    ms = r["elapsed"] / steps * 1e3
    out = {"workload": "C2 (BASELINE.json configs[1]): 8 tables x 100000 rows x 64 fp32, batch 8192, hotness L = 1, uniform ids, "
                       "3 x full-rank FeatureCross(d=512) fp32, fused Adagrad on tables; no DotInteraction (the config has none)",
           "dtype": "f32", "value": c.batch * sum(hots) / (r["elapsed"] / steps), "unit": "lookups/s", "ms_per_step": ms,
           "steps": steps, "warmup": warm, "step_stats": step_stats(r["step_ms"]),
           "host_enqueue_ms_per_step": r["enqueue_s"] / steps * 1e3,
           "note": "a step of ~40 launches on 4 MB of activations: bound by launch boundaries and the host's enqueue rate, not "
                   "by a roofline -- the entries below say how far each kernel is from its own"}
    if r["k1_s"]:
        out["roofline"] = k1_roofline(c, hots, c.batch, r["k1_s"], "embed_gather_hot1 (K1 one-hot form, fp32 rows of 256 bytes)")
    rs = roofline_step(c, hots, c.batch, r)
    if rs:
        out["roofline_step"] = rs
    if getattr(a, "c2_leg", False):
        # (own process, see main(): the first and only capture of this process) the same step replayed from a HIP graph --
        # one hipGraphLaunch per step instead of ~40 Python / ctypes launches
        c.graph, c._graph_used = True, False
        rg = measure(model, c, hots, 1, 0, dev, c.batch, steps, warm, opt_box, probe_steps=0)
        g_ms = rg["elapsed"] / steps * 1e3
        out["graph_replay"] = {"ms_per_step": g_ms, "value": c.batch * sum(hots) / (rg["elapsed"] / steps),
                               "step_stats": step_stats(rg["step_ms"]), "host_enqueue_ms_per_step": rg["enqueue_s"] / steps * 1e3,
                               "launch": "every timed step is one replay of a HIP graph captured from the eager step"}
        if g_ms < ms:
            out["eager"] = {"ms_per_step": ms, "value": out["value"]}
            out.update(ms_per_step=g_ms, value=out["graph_replay"]["value"], launch=out["graph_replay"]["launch"],
                       step_stats=out["graph_replay"]["step_stats"],
                       host_enqueue_ms_per_step=out["graph_replay"]["host_enqueue_ms_per_step"])
            out["eager"].update(host_enqueue_ms_per_step=r["enqueue_s"] / steps * 1e3, step_stats=step_stats(r["step_ms"]))
    del model
    torch.cuda.empty_cache()
    return out


def _bf16_ulps(got, ref):
    """|got - ref| in units of the bf16 ulp of the larger magnitude (fp32 tensors holding bf16-representable values)."""
    big = torch.maximum(got.abs(), ref.abs())
    ulp = torch.ldexp(torch.ones_like(big), torch.frexp(big).exponent - 8)      # 2^(floor(log2 x) - 7)
    d = (got - ref).abs() / ulp
    return torch.where(big > 0, d, torch.zeros_like(d))


def sharded_parity(model, a, hots, world, rank, dev, b_local, ids, pre, backend):
    """Self-check of the sharded step, HIP path only (no oracle here): ONE forward + ONE fused table update of the layer
    the timed steps use, checked on a slice -- the first 64 samples of rank 0's batch -- against an UNSHARDED recompute
    with plain torch arithmetic on rows read back from their owners:
      forward   rows of the slice's lookups are fetched from the shards that own them (MOD layout: row r of a table lives
                on rank r % N at stacked local row off_t + r // N; jax/embedding_utils.py:187-197), pooled per owner in
                ascending position in fp32, rounded to the partial dtype, summed in owner order, rounded once -- the
                arithmetic of K1 + krs_shard_combine -- and compared bit for bit with the layer's output (`fwd_max_ulp`);
                the plain single-pass fp32 pooling of the unsharded layer is reported beside it;
      update    the step's output gradient is a constant vector per feature, so the gradient of row r is
                (global count of r over ALL ranks' batches) x that vector, exactly; the rows' weights and Adagrad
                accumulators are read before and after and compared with acc += g^2; w -= lr g / sqrt(acc)
                (jax/test_utils.py:474-497) to one bf16 ulp.
    Collectives used by the check itself: broadcast and all-reduce (sum) of a few hundred KB.  Returns the `parity`
    object of the JSON line; `ok` false marks the line invalid."""
    import torch.distributed as dist

    emb = model.embedding
    if len(emb._sgroups) != 1 or emb._sgroups[0].fused.kind != "adagrad":
        return {"checked": False, "ok": True, "reason": "the self-check covers one sharded Adagrad group (the bench's model)"}
    # (replicated tables -- --replicate-below -- are looked up locally by the unsharded layer: not part of this check)
    g, n, d = emb._sgroups[0], emb.world, emb._sgroups[0].dim
    hots = [hots[emb._paths.index(p)] for p in g.paths]      # bag lengths of the SHARDED features, in the group's order
    multi = world > 1
    staged = multi and backend == "gloo"

    def allreduce(t):
        if not multi:
            return t
        if staged:
            h = t.cpu()
            dist.all_reduce(h)
            return h.to(t.device)
        dist.all_reduce(t)
        return t

    s_n = min(64, b_local)
    flat = torch.cat([ids[p][:s_n].reshape(-1).long() for p in g.paths])            # rank 0's slice, feature-major
    if multi:
        h = flat.cpu() if staged else flat
        dist.broadcast(h, src=0)
        flat = h.to(dev)
    f_of = torch.cat([torch.full((s_n * hots[f],), f, dtype=torch.long, device=dev) for f in range(len(g.paths))])
    t_of = torch.tensor(g.table_of_feature, device=dev)[f_of]
    off = torch.tensor(g.row_off[:-1], device=dev, dtype=torch.long)[t_of]
    own = (flat % n) == emb.rank
    local = torch.where(own, off + flat // n, torch.zeros_like(flat))
    shard, slot = emb.shard.data, emb._slot(g)

    def read_rows():
        w = allreduce(shard[local].float() * own[:, None])
        acc = allreduce(slot[local] * own[:, None])
        return w, acc

    w0, acc0 = read_rows()
    if os.environ.get("KRS_BENCH_PARITY_SABOTAGE") and rank == 0:      # (tests: the check must notice a row that is not
        shard[int(local[own][0])] += 1.0                               #  what its owner was asked for)
    lr = float(g.fused.lr_at(g.step))
    out = emb(pre)
    got = torch.stack([out[p][:s_n].detach().float() for p in g.paths], 1)          # [S, F, D]
    as_dt = lambda v: {"float32": torch.float32, "bfloat16": torch.bfloat16}[v] if isinstance(v, str) else v   # noqa: E731
    pdt, cdt = as_dt(emb._partial_dtype or emb.compute_dtype), as_dt(emb.compute_dtype)
    exp, single, pmax = torch.empty_like(got), torch.empty_like(got), torch.zeros_like(got)
    pos = 0
    for f in range(len(g.paths)):
        hot = hots[f]
        rows = w0[pos:pos + s_n * hot].view(s_n, hot, d)
        owners = (flat[pos:pos + s_n * hot] % n).view(s_n, hot)
        total, one = torch.zeros(s_n, d, device=dev), torch.zeros(s_n, d, device=dev)
        for o in range(n):
            part = torch.zeros(s_n, d, device=dev)
            for pos_l in range(hot):
                part = torch.where((owners[:, pos_l] == o)[:, None], part + rows[:, pos_l], part)
            total = total + part.to(pdt).float()
            pmax[:, f] = torch.maximum(pmax[:, f], part.abs())
        for pos_l in range(hot):
            one = one + rows[:, pos_l]
        exp[:, f], single[:, f] = total.to(cdt).float(), one.to(cdt).float()
        pos += s_n * hot
    fwd_ulp = _bf16_ulps(got, exp)
    # against the UNSHARDED single-pass pooling the sharded sum carries one rounding per partial: the distance is counted in
    # ulps of the bag's LARGEST partial (a sum of partials that cancel is small against what was rounded)
    big = torch.maximum(pmax, single.abs())
    fwd_single = torch.where(big > 0, (got - single).abs() / torch.ldexp(torch.ones_like(big), torch.frexp(big).exponent - 8),
                             torch.zeros_like(big))
    # ---- one fused update with a constant output gradient per feature
    n_f = len(g.paths)
    cvals = ((((torch.arange(n_f, device=dev)[:, None] * 131 + torch.arange(d, device=dev)[None, :] * 17) % 61) - 30).float()
             / 1024.0 + 1.0 / 2048.0).to(cdt)
    loss = sum((out[p] * cvals[f]).sum() for f, p in enumerate(g.paths))
    loss.backward()
    cnt = torch.zeros(flat.numel(), dtype=torch.float64, device=dev)
    grow = torch.zeros(flat.numel(), d, dtype=torch.float64, device=dev)
    for f2, p2 in enumerate(g.paths):                      # every feature that looks the element's table up contributes
        t2 = g.table_of_feature[f2]
        sel = t_of == t2
        if not bool(sel.any()):
            continue
        bc = torch.bincount(ids[p2].reshape(-1).long(), minlength=g.table_configs[t2].vocabulary_size)
        c = torch.zeros(flat.numel(), dtype=torch.float64, device=dev)
        c[sel] = bc[flat[sel]].double()
        c = allreduce(c)
        cnt += c
        grow += c[:, None] * cvals[f2].double()[None, :]
    w1, acc1 = read_rows()
    g32 = grow.float()
    acc_exp = acc0 + g32 * g32
    w_exp = (w0.double() - lr * grow / acc_exp.double().sqrt()).float().to(shard.dtype).float()
    upd_ulp = _bf16_ulps(w1, w_exp)
    acc_rel = ((acc1 - acc_exp).abs() / acc_exp.abs().clamp_min(1e-30)).max()
    moved = bool((w1 != w0).any())
    res = {
        "checked": True, "slice": "first %d samples of rank 0's batch" % s_n, "checked_bags": int(s_n * n_f),
        "checked_rows": int(flat.numel()), "fwd_max_ulp": float(fwd_ulp.max()),
        "fwd_bit_equal_fraction": float((fwd_ulp == 0).float().mean()),
        "fwd_max_err_vs_unsharded_single_pass_in_ulps_of_the_largest_partial": float(fwd_single.max()),
        "update_max_ulp": float(upd_ulp.max()), "update_bit_equal_fraction": float((upd_ulp == 0).float().mean()),
        "accumulator_max_rel_err": float(acc_rel), "max_lookups_of_a_checked_row": int(cnt.max()),
        "rows_moved": moved,
        "max_ulp": float(max(fwd_ulp.max(), upd_ulp.max())),
        "how": "pooled outputs vs per-owner fp32 pooling in ascending position -> partial dtype -> owner-order sum (torch, on "
               "rows read back from their owners); updated rows vs acc += g^2, w -= lr g / sqrt(acc) with g = global lookup "
               "count x the step's constant output gradient; no oracle, no CPU path",
    }
    # (one bf16 ulp on the updated weights: the kernel's lr g / sqrt(acc) in fp32 against this check's float64; the ulp
    #  distance itself is computed in fp32, hence the 1e-3 of slack)
    res["ok"] = bool(res["fwd_max_ulp"] <= 1.001 and res["update_max_ulp"] <= 1.001 and res["accumulator_max_rel_err"] <= 1e-6
                     and moved)
    return res


PHASES = ("route", "counts_host_wait", "a2a_ids", "unpack", "pool", "a2a_partials", "combine", "gather_grads",
          "a2a_grads", "k2", "allreduce_wait")


def exchange_phases(pr, n_steps, world):
    """Per-phase GPU time of the sharded step (event spans on the launch stream inside real steps, keras_rs_amd/probe.py
    spans in sharded.py): this rank's ms per step, then min / max over the ranks; for the all-to-alls the bytes this rank
    sends to OTHER ranks per call and the rate that makes (per rank: xGMI is point-to-point, so with N ranks the bytes
    spread over N - 1 links).  What a first hardware N > 1 run needs to attribute its time (examples/ml_perf/main.py:330-357
    wraps a profiler trace around the run for the same purpose)."""
    mine = {}
    for name in PHASES:
        if name in pr:
            e = pr[name]
            mine[name] = {"ms": e["ms_total"] / n_steps, "calls": e["calls"] // n_steps, "bytes": e["work_total"] / n_steps}
    allr = [mine]
    if world > 1:
        allr = [None] * world
        torch.distributed.all_gather_object(allr, mine)
    out = {}
    for name in PHASES:
        rows = [r[name] for r in allr if r and name in r]
        if not rows:
            continue
        ms = [r["ms"] for r in rows]
        ent = {"ms_per_step": mine.get(name, rows[0])["ms"], "min_ms": min(ms), "max_ms": max(ms), "calls_per_step": rows[0]["calls"]}
        if name.startswith("a2a"):
            b = mine.get(name, rows[0])["bytes"]
            ent["bytes_off_rank_per_step"] = int(b)
            ent["bytes_per_link_per_step"] = int(b / max(world - 1, 1))
            ent["GB_per_s_per_rank"] = b / (max(ms) * 1e-3) / 1e9 if max(ms) > 0 else None
        out[name] = ent
    return out


def step_stats(step_ms):
    """Median / max of the per-step GPU times and how many steps stalled (> 1.5 x median)."""
    med = float(np.median(step_ms))
    return {"median_ms": med, "min_ms": float(np.min(step_ms)), "max_ms": float(np.max(step_ms)),
            "steps_over_1.5x_median": int(sum(t > 1.5 * med for t in step_ms)),
            "each_ms": [round(float(t), 2) for t in step_ms],
            "source": "HIP events at the step boundaries of the timed steps (GPU time; `ms_per_step` is wall clock / steps)"}


def roofline_step(a, hots, b_local, res):
    """In-step roofline entries from the probe spans (events around the C-ABI calls inside real steps, GPU backlogged):
    K2 apply against HBM (SURVEY.md section 8d bytes with U = distinct touched rows of this batch), the FeatureCross
    GEMMs in aggregate against the dense bf16 MFMA peak, K1 and the K2 plan as they run inside the step."""
    pr, n = res.get("probe"), res.get("probe_steps", 0)
    if not pr or not n:
        return None
    out = []
    nnz, bags, d = b_local * sum(hots), b_local * a.tables, a.dim
    if "k2_apply" in pr and "unique_rows" in res:
        u = res["unique_rows"]
        # bags*D*s_g (gradient) + nnz*12 (sorted key + (bag, position)) + U*(2*D*s_t + 2*D*4 Adagrad accumulator)
        slot = 8 if a.rowwise_adagrad else 2 * d * 4
        es = 4 if getattr(a, "fp32", False) else 2
        alg = bags * d * es + nnz * 12 + u * (2 * d * es + slot)
        sec = pr["k2_apply"]["ms_total"] / n * 1e-3
        traffic = None
        if not a.criteo_vocab and a.id_skew == 0 and not a.rowwise_adagrad and a.batch == 65536 and a.vocab == 1_000_000 \
                and a.tables == 26:
            if list(hots) == (ML_PERF_HOTS * 8)[: a.tables]:
                traffic = pmc_traffic_key("bag_apply_fast_kernel<adagrad> multi-hot")
            elif list(hots) == [1] * a.tables:
                traffic = pmc_traffic_key("bag_apply_fast_kernel<adagrad> L=1")
        out.append({"kernel": "bag_apply_fast_kernel<adagrad> (K2 apply, krs_embed_bag_bwd_fused_adagrad)", "bound": "hbm",
                    "achieved": alg / sec / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": alg / sec / HBM_PEAK,
                    "launch_us": sec * 1e6, "algorithmic_bytes": alg, "unique_rows": u, "traffic": traffic,
                    "traffic_source": None if traffic is None else "profiles/k1_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                      "passes of this kernel at this shape, re-read in round 6; Infinity-Cache hits are counted, not excluded)"})
    out += gemm_family_rooflines(a, pr, n, b_local)
    if "gemm_cross_bwd" in pr:
        e = pr["gemm_cross_bwd"]
        calls = e["calls"] // n
        sec = e["ms_total"] / n * 1e-3
        dcols = (a.tables + 1) * a.dim
        # per launch: operands (dh [B, p] + U [d, p]) + R, x0, u, dL/dx0 in and G, dz, dL/dx0 out: seven [B, d] bf16 streams
        alg = calls * (7 * b_local * dcols * 2 + (b_local + dcols) * a.projection * 2)
        traffic = None
        if a.batch == 65536 and a.tables == 26 and a.dim == 128 and a.projection == 512 and b_local == a.batch:
            per_launch = pmc_traffic_key("krs_gemm_cross_bwd (C3 shape)")
            traffic = None if per_launch is None else calls * per_launch
        out.append({"kernel": "krs_gemm_cross_bwd x %d per step (dx = dh U^T + g with the elementwise backward of the layer "
                              "below in its epilogue; NOT in the aggregate above)" % calls,
                    "bound": "hbm", "achieved": alg / sec / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                    "frac": alg / sec / HBM_PEAK, "ms_per_step": sec * 1e3, "algorithmic_bytes": alg,
                    "flops_per_step": e["work_total"] / n, "traffic": traffic,
                    "traffic_source": None if traffic is None else "profiles/k1_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                      "passes of this kernel at this shape under scripts/exp/gemm_bench, x launches per step; read in round 6)"})
    for key, label in (("k1", "K1 inside the step (beside the K2 plan on the side stream)"),
                       ("k2_plan", "K2 plan (radix sort + segment list; side stream)")):
        if key in pr:
            out.append({"kernel": label, "ms_per_step": pr[key]["ms_total"] / n, "calls_per_step": pr[key]["calls"] // n})
    return out


GEMM_ROLES = {   # (layout + epilogue form) -> what the product is in a FeatureCross layer (feature_cross.py:182-194 and its autodiff)
    "nt": "h = x U  and  dh = dz K^T  (same shape)", "nt:cross": "y = x0 * (h K + b) + x  (cross epilogue: x0, x in; u, y out)",
    "nt:res": "dx = dh U^T + g  (residual epilogue; the layers above the bottom one run krs_gemm_cross_bwd instead)",
    "tn": "weight gradients dK = h^T dz / dU = x^T dh  (fp32 output, split-K slabs reduced in a fixed order)",
    "nn": "A B (row-major operands)",
}


def gemm_family_rooflines(a, pr, n, b_local):
    """One roofline entry per krs_gemm product FAMILY (keras_rs_amd/dense_ops.py names its probe spans by operand layout,
    epilogue form, dtype and shape): flops and ALGORITHMIC bytes of the call (operands once, outputs once, the epilogue's
    streams once), `bound` = whichever of flops / MFMA peak and bytes / 8 TB/s is the longer time, `achieved` in that
    bound's unit, `frac` = that time / the measured launch time; `traffic` = HBM bytes per launch from the committed PMC
    passes when this is the C3 shape they were taken on (profiles/k1_pmc.json `gemm families`)."""
    out, agg_fl, agg_ms, agg_calls, agg_peak = [], 0.0, 0.0, 0, MFMA_BF16_PEAK
    for key in sorted(k for k in pr if k.startswith("gemm[")):
        form, dt = key[5:key.index("]")].split(" ")
        m, nn, k = (int(v) for v in key[key.index("]") + 2:].split("x"))
        e = pr[key]
        calls, sec = e["calls"] // n, e["ms_total"] / e["calls"] * 1e-3      # per launch
        es = 2 if dt == "bf16" else 4
        peak = MFMA_BF16_PEAK if dt == "bf16" else MFMA_F32_PEAK
        layout = form.split(":")[0]
        flops = 2.0 * m * nn * k
        byts = (m * k + nn * k) * es + m * nn * (4 if layout == "tn" else es)
        if form.endswith(":cross"):
            byts += 3 * m * nn * es          # x0, x in; u out (y is the product's output, counted above)
        elif form.endswith(":res"):
            byts += m * nn * es              # R in
        t_m, t_h = flops / peak, byts / HBM_PEAK
        bound = "mfma" if t_m >= t_h else "hbm"
        ent = {"kernel": "krs_gemm %s %s %dx%dx%d x %d per step -- %s" % (form, dt, m, nn, k, calls, GEMM_ROLES.get(form, form)),
               "bound": bound, "launch_us": sec * 1e6, "calls_per_step": calls, "flops_per_launch": flops,
               "algorithmic_bytes": byts, "frac": max(t_m, t_h) / sec}
        if bound == "mfma":
            ent.update(achieved=flops / sec / 1e12, peak=peak / 1e12, unit="TFLOP/s")
        else:
            ent.update(achieved=byts / sec / 1e9, peak=HBM_PEAK / 1e9, unit="GB/s")
        ent["hbm_time_us"], ent["mfma_time_us"] = t_h * 1e6, t_m * 1e6
        tr = pmc_traffic_key("gemm families", "%s %s %dx%dx%d" % (form, dt, m, nn, k))
        ent["traffic"] = tr
        if tr is not None:
            ent["traffic_source"] = ("profiles/k1_pmc.json `gemm families` (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                     "kernel at this shape under scripts/exp/gemm_bench, read in round 6: profiles/r6_gemm_hbm_pmc.txt)")
        out.append(ent)
        agg_fl += flops * calls
        agg_ms += e["ms_total"] / n
        agg_calls += calls
        agg_peak = peak
    if out:
        sec = agg_ms * 1e-3
        out.append({"kernel": "krs_gemm x %d per step, all families above in aggregate (against the MFMA peak; the per-family "
                              "entries say which of them are HBM-bound)" % agg_calls, "bound": "mfma",
                    "achieved": agg_fl / sec / 1e12, "peak": agg_peak / 1e12, "unit": "TFLOP/s", "frac": agg_fl / sec / agg_peak,
                    "ms_per_step": agg_ms, "flops_per_step": agg_fl, "traffic": None, "aggregate": True})
    return out


def k1_roofline(a, hots, b_local, k1_s, kernel, gather_form=False, in_step_s=None):
    """`roofline` of K1.  `k1_s` = one launch on an otherwise idle stream (blocker-backed event pairs behind the timed
    steps); `in_step_s` = the same launch INSIDE real steps (probe span: beside the K2 plan's kernels on the side stream, as
    the timed steps run it).  When both exist the in-step duration is the one `achieved` / `frac` are computed from and the
    isolated one is kept beside it as `isolated_us` (round-5 review: the line must price what the step runs)."""
    iso_s = k1_s
    if in_step_s:
        k1_s = in_step_s
    nnz = b_local * sum(hots)
    # gather form (sharded owner side): one output vector per lookup, i.e. `bags` = nnz
    alg = k1_bytes(nnz, nnz if gather_form else b_local * a.tables, a.dim, 4 if getattr(a, "fp32", False) else 2)
    achieved = alg / k1_s
    c3_shape = a.batch == 65536 and a.tables == 26 and a.vocab == 1_000_000 and a.dim == 128 and not getattr(a, "fp32", False)
    traffic = None if gather_form or a.criteo_vocab or a.id_skew > 0 or not c3_shape else pmc_traffic(kernel)
    return {"kernel": kernel, "bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
            "frac": achieved / HBM_PEAK, "traffic": traffic,
            "traffic_source": None if traffic is None else "profiles/k1_pmc.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                              "passes of this kernel at this shape, re-read in round 6 (counters cannot be read inside this "
                              "run)",
            "launch_us": k1_s * 1e6,
            "launch_us_source": ("HIP events around the call inside the probe steps (in-step)" if in_step_s else
                                 "HIP events around one launch behind the timed steps (isolated)"),
            "isolated_us": iso_s * 1e6, "isolated_frac": alg / iso_s / HBM_PEAK,
            "algorithmic_bytes": alg}


def pmc_traffic_key(key, sub=None):
    try:
        with open(os.path.join(ROOT, "profiles", "k1_pmc.json")) as f:
            rec = json.load(f)[key]
        return (rec if sub is None else rec[sub])["hbm_bytes_per_launch"]
    except (OSError, ValueError, KeyError, TypeError):
        return None


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes (profiles/k1_pmc.json,
    FETCH_SIZE doubled per MI355X_MICROARCH.md, + WRITE_SIZE), or None when no counter run is recorded."""
    try:
        with open(os.path.join(ROOT, "profiles", "k1_pmc.json")) as f:
            rec = json.load(f)
        for name, v in rec.items():
            if name in kernel:
                return v["hbm_bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        pass
    return None


LINE_LIMIT = 4096     # bytes of the LAST stdout line (the driver keeps a ~8 KB tail of stdout + stderr: round-5's 21 KB line was cut)


def _sig(v, digits=6):
    """Floats to `digits` significant figures (the line is a summary: the side file keeps every bit)."""
    if isinstance(v, float):
        return float("%.*g" % (digits, v)) if math.isfinite(v) else None
    return v


def _pick(obj, keys):
    return {k: _sig(obj[k]) for k in keys if isinstance(obj, dict) and k in obj}


ROOFLINE_KEYS = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "launch_us", "launch_us_source", "isolated_us",
                 "isolated_frac",
                 "algorithmic_bytes", "ms_per_step", "calls_per_step")


def dominant_roofline(entries, single=False):
    """The `roofline_step` entry that costs the step the most time and has a roofline of its own (a `frac`): per-step time =
    `ms_per_step`, or launch_us x calls_per_step.  `single` leaves the aggregate-over-families entry out: the answer is then
    one kernel (one product family), not a sum."""
    best, best_ms = None, -1.0
    for e in entries or []:
        if "frac" not in e or (single and e.get("aggregate")):
            continue
        ms = e["ms_per_step"] if "ms_per_step" in e else e.get("launch_us", 0.0) * e.get("calls_per_step", 1) * 1e-3
        if ms > best_ms:
            best, best_ms = dict(e, ms_per_step=ms), ms
    return best


def compact_line(full, detail_path):
    """The ONE line the driver parses: the contract's keys, `roofline` (K1, the north-star kernel, at its in-step duration),
    `roofline_dominant` (the kernel family that costs the step the most time, with its own fraction), `cpu_baseline`, the
    sustained / secondary legs as one number each, the N > 1 self-check, and the path of the side file that holds everything
    else (`roofline_step`, `also`, `also_c2`, `phases`, `graph_leg`, per-step times ...).  Always below LINE_LIMIT bytes:
    optional members are dropped, last first, until it fits (tests/test_layers_host.py)."""
    line = _pick(full, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                        "vs_baseline", "dtype", "data"))
    cfg = full.get("config", {})
    line["config"] = {"workload": str(cfg.get("workload", ""))[:330], "global_batch": cfg.get("global_batch"),
                      "parallelism": str(cfg.get("parallelism", ""))[:170]}
    if "roofline" in full:
        line["roofline"] = dict(_pick(full["roofline"], ROOFLINE_KEYS), kernel=str(full["roofline"].get("kernel", ""))[:80])
    cb = full.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = dict(_pick(cb, ("value", "unit", "cores", "kind")), sample=str(cb.get("sample", ""))[:260],
                                    implementation=str(cb.get("implementation", ""))[:150])
    optional = []           # (key, value) in order of importance: dropped from the END when the line is too long
    if full.get("invalid"):
        optional.append(("invalid", str(full["invalid"])[:200]))
    if "overflow_steps" in full:
        optional.append(("overflow_steps", full["overflow_steps"]))
    if isinstance(full.get("parity"), dict):
        optional.append(("parity", _pick(full["parity"], ("checked", "ok", "fwd_max_ulp", "update_max_ulp", "checked_rows"))))
    for key, single in (("roofline_dominant", False), ("roofline_dominant_kernel", True)):
        dom = dominant_roofline(full.get("roofline_step"), single)
        if dom and not (single and dom.get("kernel") == (dominant_roofline(full.get("roofline_step")) or {}).get("kernel")):
            optional.append((key, dict(_pick(dom, ROOFLINE_KEYS), kernel=str(dom.get("kernel", ""))[:80])))
    if isinstance(full.get("sustained"), dict):
        optional.append(("sustained", _pick(full["sustained"], ("steps", "ms_per_step", "median_ms", "vs_timed_region"))))
    if isinstance(full.get("step_stats"), dict):
        optional.append(("step_stats", _pick(full["step_stats"], ("median_ms", "min_ms", "max_ms", "steps_over_1.5x_median"))))
    if "host_enqueue_ms_per_step" in full:
        optional.append(("host_enqueue_ms_per_step", _sig(full["host_enqueue_ms_per_step"])))
    if isinstance(full.get("also"), dict):
        al = _pick(full["also"], ("value", "unit", "ms_per_step"))
        al["workload"] = str(full["also"].get("workload", ""))[:60]
        if isinstance(full["also"].get("roofline"), dict):
            al["roofline"] = _pick(full["also"]["roofline"], ("kernel", "frac", "launch_us", "algorithmic_bytes", "traffic"))
            al["roofline"]["kernel"] = str(al["roofline"].get("kernel", ""))[:60]
        optional.append(("also", al))
    for key, keys in (("graph_leg", ("attempted", "ok", "ms_per_step", "value", "promoted_to_value", "error")),
                      ("eager_leg", ("ms_per_step", "value")), ("full_model", ("ms_per_step", "value")),
                      ("host_inputs", ("ms_per_step", "value", "loader_threads"))):
        if isinstance(full.get(key), dict):
            sub = _pick(full[key], keys)
            if "error" in sub:
                sub["error"] = str(sub["error"])[:160]
            optional.append((key, sub))
    if isinstance(full.get("exchange"), dict):
        optional.append(("exchange", _pick(full["exchange"], ("mode", "bytes_at_capacity", "bytes_at_need", "capacity", "need"))))
    for key in ("ranks", "backend", "virtual_world", "a2a_bytes_per_step", "embed_fwd_lookups_per_s", "launch"):
        if full.get(key) is not None:
            optional.append((key, _sig(full[key]) if not isinstance(full[key], str) else full[key][:120]))
    if isinstance(full.get("also_c2"), dict):
        optional.append(("also_c2", _pick(full["also_c2"], ("value", "unit", "ms_per_step", "dtype", "error"))))
    if isinstance(full.get("phases"), dict):
        optional.append(("phases_ms", {k: _sig(v.get("ms_per_step"), 4) for k, v in full["phases"].items() if isinstance(v, dict)}))
    line["detail"] = detail_path
    for key, val in optional:
        line[key] = val
    while len(json.dumps(line)) >= LINE_LIMIT and optional:
        key, _ = optional.pop()
        line.pop(key, None)
    assert len(json.dumps(line)) < LINE_LIMIT, "the headline members alone exceed the line limit"
    return line


def write_detail(full, path):
    """Everything the run measured, as one JSON file: `path` (default ./bench_detail.json) and, when the run is inside a
    gpurun snapshot, gpurun_out/bench_detail.json as well (the directory that travels back)."""
    paths = [path]
    if os.path.isdir(os.path.join(ROOT, "gpurun_out")) or os.environ.get("GRAFT_REPO_ROOT"):
        paths.append(os.path.join(ROOT, "gpurun_out", "bench_detail.json"))
    written = []
    for p in paths:
        try:
            os.makedirs(os.path.dirname(os.path.abspath(p)), exist_ok=True)
            with open(p, "w") as f:
                json.dump(full, f, indent=1)
            written.append(p)
        except OSError:
            pass
    return written


def main():
    a = parse()
    if a.gpus > 1 and "RANK" not in os.environ:
        sys.exit(self_launch(a.gpus))
    rank, world, local, backend = dist_setup(a.gpus, a.dist_backend, single_rank_group=a.force_sharded and a.rccl_self)
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    if a.virtual_world > 1 and (world > 1 or not a.force_sharded or a.rccl_self):
        raise SystemExit("--virtual-world N goes with --force-sharded on one GPU (no process group)")
    vworld = a.virtual_world if a.virtual_world > 1 else world      # the job size the SHAPES follow
    b_local = a.batch // vworld
    hots_multi = (ML_PERF_HOTS * 8)[: a.tables]
    hots_one = [1] * a.tables
    primary = hots_one if a.hotness == "1" else hots_multi
    secondary = hots_multi if a.hotness == "1" else hots_one

    from keras_rs_amd.build import build

    if rank == 0:
        build()
    if world > 1:
        torch.distributed.barrier()

    # The second stream for the cross layers' weight gradients (keras_rs_amd/autograd.py, opt-in) stays OFF here since the
    # elementwise backward moved into the data-gradient products (krs_gemm_cross_bwd): the pass it used to run beside is
    # gone, dK / dU would run beside the next layer's ring GEMMs, and that measures no gain (A/B in one call,
    # profiles/archive/r4k_wgrad_side_ab.txt: 10.13-10.18 ms off, 10.26 on).  KRS_WGRAD_SIDE=1 switches it on for an A/B.
    from keras_rs_amd import autograd as krs_autograd

    krs_autograd.set_wgrad_side_stream(bool(int(os.environ.get("KRS_WGRAD_SIDE", "0"))))
    if a.c2_leg:
        print(json.dumps(measure_c2(a, dev)))
        return
    model = Model(a, primary, vworld, rank)
    model.embedding.build(None)
    opt_box = [None]
    parity = None
    if a.virtual_world > 1:
        parity = {"checked": False, "ok": True, "reason": "virtual world: the other ranks' shards do not exist in this process"}
    elif (world > 1 or a.force_sharded) and not a.no_parity:
        # one step of the layer the timed steps use, checked on a slice against an unsharded recompute (HIP path only)
        ids0, _ = make_inputs(a, primary, b_local, rank, dev)
        parity = sharded_parity(model, a, primary, world, rank, dev, b_local, ids0, model.embedding.preprocess(ids0), backend)
        del ids0
    wall = {"setup_build_model": time.perf_counter() - T_START}
    t_mark = time.perf_counter()
    r1 = measure(model, a, primary, world, rank, dev, b_local, a.steps, a.warmup, opt_box, probe_steps=a.probe_steps,
                 sustained_steps=a.sustained_steps)
    wall["primary_leg"] = time.perf_counter() - t_mark
    t_mark = time.perf_counter()
    # the other C3 bag-length list (SURVEY.md section 8d lists both), same tables and model, shorter run.  The
    # shapes change (ids, plan workspace), so the leg gets its own warm-up of at least 5 steps: the caching
    # allocator re-carves its blocks during the first steps after a shape change
    sec_steps = max(3, a.steps // 2)
    r2 = measure(model, a, secondary, world, rank, dev, b_local, sec_steps, max(5, a.warmup), opt_box,
                 probe_steps=a.probe_steps)
    wall["secondary_leg"] = time.perf_counter() - t_mark
    elapsed, k1_s, elapsed2, k1_s2 = r1["elapsed"], r1["k1_s"], r2["elapsed"], r2["k1_s"]
    c2 = None
    want_c2 = not a.no_c2 and world == 1 and not a.force_sharded and not a.criteo_vocab and not a.graph
    host = None
    if a.host_inputs > 0 and world == 1 and not a.force_sharded:
        # ids start in host memory: a small pool of batches cycles through the loader threads, which
        # concatenate them into page-locked memory and upload on their own streams
        from keras_rs_amd.data import ThreadedDataLoader

        rng = np.random.default_rng(7)
        pool = [{f"cat_{t:02d}_id": rng.integers(0, a.vocab, (b_local, primary[t]), dtype=np.int32)
                 for t in range(a.tables)} for _ in range(4)]

        def cycle():
            while True:
                yield from pool

        loader = ThreadedDataLoader(model.embedding.preprocess, cycle(), num_workers=a.host_inputs, buffer_size=4)
        el_h = measure(model, a, primary, world, rank, dev, b_local, a.steps, a.warmup, opt_box, loader=loader)["elapsed"]
        loader.stop()
        id_bytes = 4 * b_local * sum(primary)
        host = {"ms_per_step": el_h / a.steps * 1e3, "value": a.batch * sum(primary) / (el_h / a.steps),
                "unit": "lookups/s", "loader_threads": a.host_inputs, "id_bytes_per_step": id_bytes,
                "note": "ids generated on the host, ThreadedDataLoader -> preprocess -> pinned upload; PCIe-inclusive"}
    full = None
    if a.full_model and world == 1 and not a.force_sharded:
        model = None  # frees the first model's tables before the second set is built
        torch.cuda.empty_cache()
        sys.path.insert(0, os.path.join(ROOT, "examples"))
        import dlrm_dcn_v2 as ex

        fm = ex.build_model(b_local, a.vocab, primary, embedding_dim=a.dim, projection=a.projection,
                            cross_layers=a.cross_layers)
        x, y = ex.synthetic_batch(b_local, 13, a.vocab, primary, dev)
        x["large_emb_inputs"] = fm.embedding_layer.preprocess(x["large_emb_inputs"])
        box = [None]
        for _ in range(max(a.warmup, 2)):
            ex.train_step(fm, box, x, y)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            ex.train_step(fm, box, x, y)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        full = {"ms_per_step": dt * 1e3, "value": a.batch * sum(primary) / dt, "unit": "lookups/s",
                "model": "bottom MLP 13-512-256-128 (relu), 26 embeddings, 3 x FeatureCross(3456, 512), top MLP "
                         "3456-1024-1024-512-256-1 (relu / sigmoid), BCE, Adagrad everywhere"}
    sharded = world > 1 or a.force_sharded
    # ---- sharded runs over RCCL: the same K steps again, replayed from a HIP graph (forward, backward, both all-to-alls, the
    # gradient all-to-all, the dense all-reduce, the fused updates: one hipGraphLaunch per step; keras_rs_amd/graphs.py) --
    # BEHIND the eager legs, whose numbers are already in hand: if capture or replay hangs on first contact with real links,
    # a watchdog prints the eager line and leaves.  The faster of the two legs is `value`; both are in the line.
    graph_leg, graph_note = None, None
    want_graph = (sharded and backend == "nccl" and torch.distributed.is_initialized() and a.exchange == "static"
                  and not a.graph and not a.no_graph_leg and (parity is None or parity.get("ok", True)))
    finish = {"fn": None}
    if want_graph:
        import threading

        done = threading.Event()

        def watchdog():
            if not done.wait(timeout=float(os.environ.get("KRS_BENCH_GRAPH_TIMEOUT", "240"))):
                if rank == 0 and finish["fn"] is not None:
                    finish["fn"]({"attempted": True, "ok": False,
                                  "error": "capture / replay did not finish within the watchdog's limit: the eager legs stand"})
                os._exit(0)

        threading.Thread(target=watchdog, daemon=True).start()
    def describe(hots):
        return "multi-hot ml_perf lengths (sum L = %d)" % sum(hots) if sum(hots) > len(hots) else "hotness L = 1"

    def k1_name(hots):
        return "embed_bag_fwd_vec (K1, krs_embed_bag_fwd)" if sum(hots) > len(hots) else \
            "embed_gather_hot1 (K1 one-hot form, krs_embed_bag_fwd)"

    lookups = a.batch * sum(primary)
    out = {}
    if rank == 0:
        if a.criteo_vocab:
            shape_name = "C5 Criteo-1TB scale" if a.criteo_vocab >= 40_000_000 else "C3' (Criteo vocabularies, capped)"
            rows_desc = "min(Criteo-1TB vocabulary, %d) (%d rows in all)" % (a.criteo_vocab, sum(a.vocabs))
        else:
            shape_name, rows_desc = "C3 DLRM-small", "%d" % a.vocab
        ids_desc = "power-law ids (id = perm(floor(V u^%g)))" % a.id_skew if a.id_skew > 0 else "uniform ids"
        sharded = world > 1 or a.force_sharded
        if world > 1 and backend == "gloo":
            backend_desc = ("gloo, staged through the host" + (
                ": %d ranks share %d GPU(s), a functional rig, NOT a measurement of the links"
                % (world, torch.cuda.device_count()) if world > torch.cuda.device_count() else ""))
        else:
            backend_desc = {"nccl": "nccl (RCCL)", None: None}.get(backend, backend)
        out = {
            "metric": "embedding lookups/sec + DCN fwd+bwd step time, 26-table DLRM batch 65 536",
            "value": lookups / (elapsed / a.steps),
            "unit": "lookups/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",   # --gpus N keeps the GLOBAL batch (BASELINE.json: "batch 65 536 at 1/2/4/8 GPU")
            "vs_baseline": None,
            "dtype": "bf16",
            "data": "synthetic",
            "ranks": world, "backend": backend_desc,
            "config": {
                "workload": ("%s: %d tables x %s rows x %d (bf16), global batch %d, %s, %s, "
                             "DotInteraction(F=%d) + %d x FeatureCross(d=%d, projection=%d), fused %s on tables"
                             % (shape_name, a.tables, rows_desc, a.dim, a.batch, describe(primary), ids_desc, a.tables + 1,
                                a.cross_layers, (a.tables + 1) * a.dim, a.projection,
                                "row-wise Adagrad (opt-in variant, NOT the reference optimizer)" if a.rowwise_adagrad
                                else "Adagrad")),
                "global_batch": a.batch,
                "parallelism": ("single GPU" if world == 1 and not a.force_sharded else
                                ("rank 0's step of a %d-way job on ONE GPU (virtual world: 1/%d shard, batch / %d samples, ids "
                                 "routed to %d owners, device copies for the links; `value` = global batch / this step: a "
                                 "projection WITHOUT link time)" % ((a.virtual_world,) * 4)) if a.virtual_world > 1 else
                                "sharded code path on ONE GPU (dry run: %s)" % (
                                    "collectives through a one-rank RCCL communicator" if backend else
                                    "device copies stand in for the links") if world == 1
                                else f"tables MOD row-sharded over {world} GPUs, dense part DP"),
            },
            "step_stats": step_stats(r1["step_ms"]),
            "wall_seconds": wall,      # where the run's own wall clock went (host seconds per leg)
            # host time to ENQUEUE one step (the loop returns before the device has finished): below ms_per_step = the host
            # runs ahead of the GPU and the step is GPU-bound; equal to it = the host is the limit (e.g. a wait inside the step)
            "host_enqueue_ms_per_step": r1["enqueue_s"] / a.steps * 1e3,
        }
        if r1.get("sustained"):
            out["sustained"] = r1["sustained"]
        if a.virtual_world > 1:
            out["virtual_world"] = a.virtual_world
        if a.graph:
            out["launch"] = "every timed step is one replay of a HIP graph captured from the eager step (keras_rs_amd.graphs)"
        if sharded and "exchange" in r1:
            ex = r1["exchange"]
            out["a2a_bytes_per_step"] = ex.get("bytes_per_step")
            out["exchange"] = {k: v for k, v in ex.items() if k in ("mode", "bytes", "capacity", "need", "received")}
            if ex.get("mode") == "static" and ex.get("capacity") and ex.get("need"):
                # link bytes of the blocks as sized (capacity) against what this batch's data needed (the largest block
                # need over all ranks, every block sized to it): the cost of the headroom in bytes, side by side
                from keras_rs_amd import _lib as _L
                import ctypes as _C

                def block_bytes(cl, cs):
                    w = int(_L.lib().krs_shard_static_block_words(_C.c_int64(int(cl)), _C.c_int64(int(cs)), _C.c_int(0)))
                    es_p = ex["bytes"]["partials_fwd"] // max(1, vworld * ex["capacity"][1] * a.dim)
                    off = (vworld - 1) / vworld if vworld > 1 else 1.0
                    return int(off * (4 * vworld * w + 2 * vworld * int(cs) * a.dim * es_p))

                out["exchange"]["bytes_at_capacity"] = block_bytes(*ex["capacity"])
                out["exchange"]["bytes_at_need"] = block_bytes(*ex["need"])
            # static exchange: lookups beyond a block's capacity are DROPPED (the reference's id dropping) -- `value` would
            # then count dropped lookups as work, so the line says so at the top level
            if "phases" in r1:
                ph = dict(r1["phases"])
                gpu_ms = step_stats(r1["step_ms"])["median_ms"]
                ph["dense_and_rest"] = {"ms_per_step": gpu_ms - sum(v["ms_per_step"] for v in ph.values()),
                                        "note": "median GPU step minus the phases above: DotInteraction + cross stack forward / "
                                                "backward, dense optimizer, launch gaps (the probe steps keep every span on one stream)"}
                out["phases"] = ph
            out["overflow_steps"] = int(r1.get("overflow_steps", 0)) + int(r2.get("overflow_steps", 0))
            if out["overflow_steps"]:
                out["invalid"] = ("the static exchange dropped lookups in %d step(s) (capacity %s, largest per-owner need %s): "
                                  "`value` counts dropped lookups as work" % (out["overflow_steps"], ex.get("capacity"), ex.get("need")))
        second = {"workload": "same tables / model, " + describe(secondary),
                  "value": a.batch * sum(secondary) / (elapsed2 / sec_steps), "unit": "lookups/s",
                  "ms_per_step": elapsed2 / sec_steps * 1e3, "steps": sec_steps, "warmup": max(5, a.warmup),
                  "step_stats": step_stats(r2["step_ms"])}
        if k1_s is not None:
            n1 = "embed_gather_hot1 (K1 owner-side row gather of the sharded path, rank 0)" if sharded else k1_name(primary)
            n2 = n1 if sharded else k1_name(secondary)
            out["embed_fwd_lookups_per_s"] = vworld * b_local * sum(primary) / k1_s
            def in_step(res):       # K1's span inside the probe steps (the unsharded layer: one krs_embed_bag_fwd per step)
                e = (res.get("probe") or {}).get("k1")
                return None if sharded or not e or not e["calls"] else e["ms_total"] / e["calls"] * 1e-3

            out["roofline"] = k1_roofline(a, primary, b_local, k1_s, n1, sharded, in_step(r1))
            second["embed_fwd_lookups_per_s"] = vworld * b_local * sum(secondary) / k1_s2
            second["roofline"] = k1_roofline(a, secondary, b_local, k1_s2, n2, sharded, in_step(r2))
        for res, tgt, hots in ((r1, out, primary), (r2, second, secondary)):
            rs = roofline_step(a, hots, b_local, res)
            if rs:
                tgt["roofline_step"] = rs
        for key, single in (("roofline_dominant", False), ("roofline_dominant_kernel", True)):
            dom = dominant_roofline(out.get("roofline_step"), single)
            if dom:
                out[key] = dom
        out["also"] = second
        if host is not None:
            out["host_inputs"] = host
        if full is not None:
            out["full_model"] = full
        if parity is not None:
            out["parity"] = parity
            if not parity.get("ok", True):
                out["invalid"] = ("the sharded step failed its self-check against the unsharded recompute (parity: fwd_max_ulp %s, "
                                  "update_max_ulp %s)" % (parity.get("fwd_max_ulp"), parity.get("update_max_ulp")))

    def emit(graph_info=None):
        full = dict(out)
        if graph_info is not None:
            full["graph_leg"] = graph_info
        written = write_detail(full, a.detail)
        line = compact_line(full, written[0] if written else None)
        # (RCCL writes its version banner through C stdio: push it out now, so that the JSON line is the LAST line)
        import ctypes

        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        if a.print_detail:
            print("KRS_BENCH_DETAIL " + json.dumps(full))     # an EARLIER stdout line: the last one stays the compact one
        print(json.dumps(line))
        sys.stdout.flush()

    finish["fn"] = emit
    if want_graph:
        if rank == 0:
            # (the eager legs' line goes to STDERR first: should the capture take the process down -- which no handler can catch --
            #  the measurement is at least in the log)
            write_detail(dict(out, graph_leg={"attempted": True, "ok": None, "note": "written before the capture"}), a.detail)
            print("KRS_EAGER_LINE " + json.dumps(_pick(out, ("value", "ms_per_step", "n_gpus", "overflow_steps"))),
                  file=sys.stderr, flush=True)
        info = {"attempted": True, "ok": False}
        try:
            a.graph, a._graph_used = True, False
            rg = measure(model, a, primary, world, rank, dev, b_local, a.steps, max(a.warmup, 3), opt_box)
            poll = getattr(model.embedding, "poll_exchange_stats", None)
            grew = bool(poll()) if poll is not None else False
            g_ms = rg["elapsed"] / a.steps * 1e3
            info = {"attempted": True, "ok": not grew, "ms_per_step": g_ms, "value": lookups / (rg["elapsed"] / a.steps),
                    "step_stats": step_stats(rg["step_ms"]), "host_enqueue_ms_per_step": rg["enqueue_s"] / a.steps * 1e3,
                    "launch": "every timed step is ONE replay of a HIP graph captured from the eager step, collectives included "
                              "(keras_rs_amd.graphs.GraphedStep)", "capacity_grew_during_replays": grew}
            if rank == 0 and info["ok"] and rg["elapsed"] < elapsed and "invalid" not in out:
                # the replayed leg is the faster one: it becomes `value`; the eager leg stays in the line
                out["eager_leg"] = {"ms_per_step": out["ms_per_step"], "value": out["value"], "step_stats": out["step_stats"],
                                    "host_enqueue_ms_per_step": out["host_enqueue_ms_per_step"]}
                out.update(value=info["value"], ms_per_step=g_ms, step_stats=info["step_stats"],
                           host_enqueue_ms_per_step=info["host_enqueue_ms_per_step"], launch=info["launch"])
                info["promoted_to_value"] = True
        except Exception as e:   # noqa: BLE001 -- a failed capture must not cost the eager legs their line
            info = {"attempted": True, "ok": False, "error": repr(e)[:400]}
        done.set()
        if rank == 0:
            emit(info)
        # (destroy_process_group() waits forever while graphs that hold RCCL kernels are alive in this process, ROCm 7.2:
        #  every rank leaves without the teardown)
        sys.stdout.flush()
        os._exit(0)
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank != 0:
        return
    t_mark = time.perf_counter()
    if want_c2:
        # the C2 leg (BASELINE.json configs[1]) in a process of its own: its graph capture is then the first of a process, and a
        # failure costs this leg only.  It runs once this process is done with the GPU (tables freed) and BEFORE the host-CPU
        # leg: side by side they share the host's cores, and a launch-bound step of ~40 launches measured 1.89 ms instead
        # of 0.65 (and the CPU leg 2.9 M lookups/s instead of 3.5 M) -- gpurun_out of round 6, first call
        import subprocess

        model = r1 = r2 = None     # (tables and probe events freed before the leg's process asks for the GPU)
        torch.cuda.empty_cache()
        try:
            pr = subprocess.run([sys.executable, os.path.abspath(__file__), "--c2-leg", "--no-cpu-baseline"], capture_output=True,
                                text=True, timeout=420)
            out["also_c2"] = json.loads([ln for ln in pr.stdout.strip().splitlines() if ln.startswith("{")][-1])
        except Exception as e:   # noqa: BLE001
            out["also_c2"] = {"error": "the C2 leg's process failed: " + repr(e)[:300]}
        out["wall_seconds"]["c2_leg_process"] = time.perf_counter() - t_mark
        t_mark = time.perf_counter()
    if not a.no_cpu_baseline and world == 1 and not a.criteo_vocab:   # the host-CPU leg is timed on rank 0 of the single-GPU run only
        out["cpu_baseline"] = cpu_baseline(a, primary)
        out["wall_seconds"]["cpu_baseline"] = time.perf_counter() - t_mark
    out["wall_seconds"]["total_since_import"] = time.perf_counter() - T_START
    emit()


if __name__ == "__main__":
    main()
