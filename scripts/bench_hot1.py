"""A/B of the one-hot gather variants of K1 (krs_embed_set_option(KRS_EMBED_OPT_HOT1, v)) at the C3 shape:
interleaved rounds in one process, median per variant, every variant checked as an exact gather."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from keras_rs_amd import _lib as L
from keras_rs_amd.embedding_ops import FusedBags

dev = torch.device("cuda:0")
T, V, D, B = 26, 1_000_000, 128, 65536
g = torch.Generator(device=dev).manual_seed(1337)
tables = [(torch.rand(V, D, device=dev, generator=g) * 0.1 - 0.05).to(torch.bfloat16) for _ in range(T)]
ids = torch.randint(0, V, (T, B), device=dev, generator=g, dtype=torch.int32)
fb = FusedBags(tables, [(t, "sum", t * D) for t in range(T)])
slab = torch.empty(B, D + T * D, dtype=torch.bfloat16, device=dev)
out = slab[:, D:]
names = {0: "feature-major, 8 loads in flight", 1: "feature-major, 16 in flight", 2: "sample-major, 8 in flight",
         3: "sample-major, 16 in flight"}
times = {v: [] for v in names}
blocker = torch.empty(2, 1 << 29, dtype=torch.uint8, device=dev)
for v in names:
    L.check(L.lib().krs_embed_set_option(C.c_int(0), C.c_int(v)), "set_option")
    slab.zero_()
    fb.forward(ids.reshape(-1), B, hots=[1] * T, out=out)
    for t in (0, 11, 25):
        assert torch.equal(out[:, t * D:(t + 1) * D], tables[t][ids[t].long()]), (v, t)
    assert torch.count_nonzero(slab[:, :D]) == 0
for rnd in range(6):
    for v in names:
        L.check(L.lib().krs_embed_set_option(C.c_int(0), C.c_int(v)), "set_option")
        fb.forward(ids.reshape(-1), B, hots=[1] * T, out=out)
        blocker[1].copy_(blocker[0])          # launches below are queued behind a busy GPU: event gaps = kernel time
        evs = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fb.forward(ids.reshape(-1), B, hots=[1] * T, out=out)
            e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize()
        times[v] += [a.elapsed_time(b) * 1e3 for a, b in evs]
alg = T * B * (D * 2 + 4) * 2
for v, n in names.items():
    us = float(np.median(times[v]))
    print(json.dumps({"variant": v, "what": n, "median_us": round(us, 1), "min_us": round(min(times[v]), 1),
                      "GBps": round(alg / us / 1e3, 1), "frac_of_8TBps": round(alg / us / 1e3 / 8000, 3)}))
