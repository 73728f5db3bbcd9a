"""K2 apply (fused Adagrad) at the C3 shape on a SUBSET of the feature list, for counter passes per table class
(round-4 review, next #4: where do the bytes above the algorithmic count come from?).
    python scripts/exp/k2_split.py --subset all|hot100|hot1|mid [--iters 3]
all = the 26 features (sum L = 214); hot100 = the one table whose bags hold 100 lookups (6.5 M lookups on 1 M rows);
hot1 = the 13 one-hot tables (65,536 lookups each: every gradient row is read once); mid = the other 12.
Prints the algorithmic bytes of ONE apply launch and its event time."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from keras_rs_amd.embedding_ops import FusedBags

ap = argparse.ArgumentParser()
ap.add_argument("--subset", default="all")
ap.add_argument("--iters", type=int, default=3)
a = ap.parse_args()
HOTS = [3, 2, 1, 2, 6, 1, 1, 1, 1, 7, 3, 8, 1, 6, 9, 5, 1, 1, 1, 12, 100, 27, 10, 3, 1, 1]
sel = {"all": range(26), "hot100": [20], "hot1": [i for i, h in enumerate(HOTS) if h == 1],
       "mid": [i for i, h in enumerate(HOTS) if 1 < h < 100]}[a.subset]
hots = [HOTS[i] for i in sel]
T, V, D, B = len(hots), 1_000_000, 128, 65536
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1337)
tables = [(torch.rand(V, D, device=dev, generator=g) * 0.1 - 0.05).to(torch.bfloat16) for _ in range(T)]
slots = [torch.full((V, D), 0.1, dtype=torch.float32, device=dev) for _ in range(T)]
fb = FusedBags(tables, [(t, "sum", t * D) for t in range(T)], slots=slots, lrs=[0.0034] * T)
ids = torch.cat([torch.randint(0, V, (B * h,), device=dev, generator=g, dtype=torch.int32) for h in hots])
grad = (torch.rand(B, T * D, device=dev, generator=g) * 1e-3).to(torch.bfloat16)
nnz = ids.numel()
uniq, base = 0, 0
for h in hots:
    m = torch.zeros(V, dtype=torch.bool, device=dev)
    m[ids[base: base + B * h].long()] = True
    uniq += int(m.sum())
    base += B * h
ws = fb.plan_backward(ids, B, hots=hots, global_order=False)
fb.backward_fused("adagrad", ws, grad, B, nnz, hots=hots)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.iters + 1)]
ev[0].record()
for i in range(a.iters):
    fb.backward_fused("adagrad", ws, grad, B, nnz, hots=hots)
    ev[i + 1].record()
torch.cuda.synchronize()
us = sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(a.iters))[a.iters // 2]
alg = B * T * D * 2 + nnz * 12 + uniq * (2 * D * 2 + 2 * D * 4)
print(json.dumps({"subset": a.subset, "tables": T, "lookups": nnz, "unique_rows": uniq, "lookups_per_row": nnz / uniq,
                  "algorithmic_bytes": alg, "gradient_rows_bytes": B * T * D * 2, "gathered_gradient_bytes": nnz * D * 2,
                  "row_and_slot_bytes": uniq * (2 * D * 2 + 2 * D * 4), "apply_us": us, "algorithmic_GBps": alg / us / 1e3}))
