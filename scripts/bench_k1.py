"""Micro-benchmark of K1 (krs_embed_bag_fwd) at the C3 shape; development aid."""
import argparse
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import json
import time

import numpy as np
import torch

from keras_rs_amd.embedding_ops import FusedBags

ap = argparse.ArgumentParser()
ap.add_argument("--tables", type=int, default=26)
ap.add_argument("--vocab", type=int, default=1_000_000)
ap.add_argument("--dim", type=int, default=128)
ap.add_argument("--batch", type=int, default=65536)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--multihot", action="store_true")
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--bpgs", default="")
ap.add_argument("--stacked", action="store_true")
ap.add_argument("--copyref", action="store_true")
ap.add_argument("--idmode", default="rand")
ap.add_argument("--fmajor_out", action="store_true")
ap.add_argument("--scale", action="store_true")
ap.add_argument("--hotrows", default="", help="comma list of KRS_EMBED_OPT_HOTROWS values to A/B (0,64,128)")
a = ap.parse_args()

dev = torch.device("cuda:0")
dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
g = torch.Generator(device=dev).manual_seed(1337)
if a.stacked:
    big = torch.empty(a.tables * a.vocab, a.dim, dtype=dt, device=dev)
    tables = [big[t * a.vocab:(t + 1) * a.vocab] for t in range(a.tables)]
    for t in tables:
        t.copy_((torch.rand(a.vocab, a.dim, device=dev, generator=g) * 0.1 - 0.05).to(dt))
else:
    tables = [(torch.rand(a.vocab, a.dim, device=dev, generator=g) * 0.1 - 0.05).to(dt) for _ in range(a.tables)]
HOTS = [3, 2, 1, 2, 6, 1, 1, 1, 1, 7, 3, 8, 1, 6, 9, 5, 1, 1, 1, 12, 100, 27, 10, 3, 1, 1]
hots = (HOTS * 4)[: a.tables] if a.multihot else [1] * a.tables
gi = torch.Generator(device=dev).manual_seed(1338)
if a.idmode.startswith("pow"):
    # power-law skew (C5-like): id = floor(V * u^e), e.g. --idmode pow4: the lowest 1 % of the rows
    # take 32 % of the lookups, the lowest 0.01 % take 10 %
    e = float(a.idmode[3:] or 4)
    ids = torch.cat([(torch.rand(a.batch * h, device=dev, generator=gi) ** e * a.vocab).to(torch.int32).clamp_(0, a.vocab - 1)
                     for h in hots])
elif a.idmode == "seq":
    ids = torch.cat([(torch.arange(a.batch * h, device=dev, dtype=torch.int32) % a.vocab) for h in hots])
else:
    ids = torch.cat([torch.randint(0, a.vocab, (a.batch * h,), device=dev, generator=gi, dtype=torch.int32)
                     for h in hots])
if a.fmajor_out:
    fb = FusedBags(tables, [(t, "sum", t * a.batch * a.dim) for t in range(a.tables)])
    out = torch.empty(a.batch * a.tables, a.dim, dtype=dt, device=dev)
else:
    fb = FusedBags(tables, [(t, "sum", t * a.dim) for t in range(a.tables)])
    out = torch.empty(a.batch, a.tables * a.dim, dtype=dt, device=dev)
if a.copyref:
    src = torch.empty(1 << 30, dtype=torch.uint8, device=dev); dst = torch.empty_like(src)
    for _ in range(3): dst.copy_(src)
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): dst.copy_(src)
    e1.record(); torch.cuda.synchronize()
    print(json.dumps({"copy_GBps": 2 * 10 * (1 << 30) / (e0.elapsed_time(e1) * 1e-3) / 1e9}))
import ctypes as _C
from keras_rs_amd import _lib as _L
_variants = [(b, h) for b in (a.bpgs.split(",") if a.bpgs else [""]) for h in (a.hotrows.split(",") if a.hotrows else [""])]
for bpg, hr in _variants:
    if bpg:
        os.environ["KRS_BPG"] = bpg
    if hr != "":
        _L.check(_L.lib().krs_embed_set_option(_C.c_int(3), _C.c_int(int(hr))), "set_option")
    for _ in range(3):
        fb.forward(ids, a.batch, hots=hots, out=out)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.iters + 1)]
    ev[0].record()
    for i in range(a.iters):
        fb.forward(ids, a.batch, hots=hots, out=out, want_scale=a.scale)
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(a.iters)]) * 1e-3
    nnz = ids.numel()
    es = 2 if a.dtype == "bf16" else 4
    bytes_ = nnz * (a.dim * es + 4) + a.batch * a.tables * (a.dim * es)
    print(json.dumps({"lib": os.path.basename(os.environ.get("KRS_LIB", "default")), "bpg": bpg, "hotrows": hr, "stacked": a.stacked, "fmajor_out": a.fmajor_out, "vocab": a.vocab, "idmode": a.idmode, "nnz": nnz, "median_us": float(np.median(ts) * 1e6),
                      "min_us": float(ts.min() * 1e6), "lookups_per_s": nnz / float(np.median(ts)),
                      "GBps": bytes_ / float(np.median(ts)) / 1e9, "frac_of_8TBps": bytes_ / float(np.median(ts)) / 8e12}))
    
_L.lib().krs_embed_set_option(_C.c_int(3), _C.c_int(0))
# ---- K2 timings (plan = sort; apply = fused Adagrad) ----
def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters

if not a.fmajor_out:
    fb.slots = [torch.full(t.shape, 0.1, dtype=torch.float32, device=dev) for t in tables]
    fb.lrs = [0.0034] * a.tables
    grad = (torch.rand(a.batch, a.tables * a.dim, device=dev) * 1e-3).to(dt)
    nnz = ids.numel()
    from keras_rs_amd import _lib as L
    import ctypes as C
    for pv in (1, 0, 1, 0):          # KRS_EMBED_OPT_PLAN: 1 = global sort, 0 = table-segmented sort
        L.check(L.lib().krs_embed_set_option(C.c_int(2), C.c_int(pv)), "set_option")
        print(json.dumps({"plan_variant": pv, "k2_plan_us": timeit(lambda: fb.plan_backward(ids, a.batch, hots=hots, global_order=False)) * 1e6}))
    t_plan = timeit(lambda: fb.plan_backward(ids, a.batch, hots=hots, global_order=False))
    ws = fb.plan_backward(ids, a.batch, hots=hots, global_order=False)
    for _ in range(2):
        t_ada = timeit(lambda: fb.backward_fused("adagrad", ws, grad, a.batch, nnz, hots=hots))
        t_sgd = timeit(lambda: fb.backward_fused("sgd", ws, grad, a.batch, nnz, hots=hots))
        print(json.dumps({"k2_plan_us": t_plan * 1e6, "k2_adagrad_us": t_ada * 1e6, "k2_sgd_us": t_sgd * 1e6}))

# ---- does the plan hide under the apply kernel?  (two streams) ----
if not a.fmajor_out:
    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
    for variant in (0,):
        for order in ("apply_first", "plan_first"):
            ts = []
            for rep in range(6):
                torch.cuda.synchronize()
                e0, eA, eB = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                e0.record()
                sA.wait_stream(torch.cuda.current_stream()); sB.wait_stream(torch.cuda.current_stream())
                def run_apply():
                    with torch.cuda.stream(sA):
                        fb.backward_fused("adagrad", ws, grad, a.batch, nnz, hots=hots)
                        eA.record()
                def run_plan():
                    with torch.cuda.stream(sB):
                        fb.plan_backward(ids, a.batch, hots=hots, global_order=False)
                        eB.record()
                (run_apply(), run_plan()) if order == "apply_first" else (run_plan(), run_apply())
                torch.cuda.synchronize()
                ts.append((e0.elapsed_time(eA), e0.elapsed_time(eB)))
            ts = np.array(ts[2:])
            print(json.dumps({"overlap": order, "apply_variant": variant, "apply_done_us": float(np.median(ts[:, 0]) * 1e3),
                              "plan_done_us": float(np.median(ts[:, 1]) * 1e3)}))
