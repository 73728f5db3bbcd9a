"""Measurement only: what do the per-lookup gradient gathers cost in the K2 apply kernel (fused Adagrad, C3 multi-hot)?
Run once per library build (KRS_LIB=...): the product build, -DKRS_K2_EXP=1 (every gather reads sample 0: cache hits -- the
kernel without its gather traffic) and -DKRS_K2_EXP=2 (addresses of a FEATURE-MAJOR gradient slab [feature][sample][dim]: a
table's 16.8 MB slice contiguous -- the layout the round-3 review proposed).  The two experiment builds compute wrong
updates; only their time is read.   scripts/exp/build_variants.sh does the builds (KRS_VARIANT_SRC=embed_bag_bwd)."""
import json
import os
import sys

import torch

sys.path.insert(0, ".")
from keras_rs_amd.embedding_ops import FusedBags  # noqa: E402

dev = torch.device("cuda:0")
T, V, D, B = 26, 1_000_000, 128, 65536
HOTS = [3, 2, 1, 2, 6, 1, 1, 1, 1, 7, 3, 8, 1, 6, 9, 5, 1, 1, 1, 12, 100, 27, 10, 3, 1, 1]
g = torch.Generator(device=dev).manual_seed(1337)
tables = [(torch.rand(V, D, device=dev, generator=g) * 0.1 - 0.05).to(torch.bfloat16) for _ in range(T)]
ids = torch.cat([torch.randint(0, V, (B * h,), device=dev, generator=g, dtype=torch.int32) for h in HOTS])
fb = FusedBags(tables, [(t, "sum", t * D) for t in range(T)])
fb.slots = [torch.full(t.shape, 0.1, dtype=torch.float32, device=dev) for t in tables]
fb.lrs = [0.0034] * T
grad = (torch.rand(B, T * D, device=dev) * 1e-3).to(torch.bfloat16)
nnz = ids.numel()
ws = fb.plan_backward(ids, B, hots=HOTS, global_order=False)


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


out = {"lib": os.environ.get("KRS_LIB", "product build")}
for rep in range(2):
    out[f"k2_adagrad_us_{rep}"] = timeit(lambda: fb.backward_fused("adagrad", ws, grad, B, nnz, hots=HOTS))
    out[f"k2_sgd_us_{rep}"] = timeit(lambda: fb.backward_fused("sgd", ws, grad, B, nnz, hots=HOTS))
print(json.dumps(out))
