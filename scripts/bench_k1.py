"""Micro-benchmark of K1 (krs_embed_bag_fwd) at the C3 shape; development aid."""
import argparse
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
a = ap.parse_args()

dev = torch.device("cuda:0")
dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
g = torch.Generator(device=dev).manual_seed(1337)
tables = [(torch.rand(a.vocab, a.dim, device=dev, generator=g) * 0.1 - 0.05).to(dt) for _ in range(a.tables)]
HOTS = [3, 2, 1, 2, 6, 1, 1, 1, 1, 7, 3, 8, 1, 6, 9, 5, 1, 1, 1, 12, 100, 27, 10, 3, 1, 1]
hots = (HOTS * 4)[: a.tables] if a.multihot else [1] * a.tables
gi = torch.Generator(device=dev).manual_seed(1338)
ids = torch.cat([torch.randint(0, a.vocab, (a.batch * h,), device=dev, generator=gi, dtype=torch.int32)
                 for h in hots])
fb = FusedBags(tables, [(t, "sum", t * a.dim) for t in range(a.tables)])
out = torch.empty(a.batch, a.tables * a.dim, dtype=dt, device=dev)
for _ in range(3):
    fb.forward(ids, a.batch, hots=hots, out=out)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.iters + 1)]
ev[0].record()
for i in range(a.iters):
    fb.forward(ids, a.batch, hots=hots, out=out)
    ev[i + 1].record()
torch.cuda.synchronize()
ts = np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(a.iters)]) * 1e-3
nnz = ids.numel()
es = 2 if a.dtype == "bf16" else 4
bytes_ = nnz * (a.dim * es + 4) + a.batch * a.tables * (a.dim * es)
print(json.dumps({"kernel": "krs_embed_bag_fwd", "nnz": nnz, "median_us": float(np.median(ts) * 1e6),
                  "min_us": float(ts.min() * 1e6), "lookups_per_s": nnz / float(np.median(ts)),
                  "GBps": bytes_ / float(np.median(ts)) / 1e9, "frac_of_8TBps": bytes_ / float(np.median(ts)) / 8e12}))
