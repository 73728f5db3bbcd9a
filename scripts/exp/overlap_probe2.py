"""Does an HBM-bound elementwise pass (the dz / dx0 kernel of a cross layer) hide under a weight-gradient GEMM of the
NEXT-upper layer when that GEMM leaves half of every CU free?  C3 shapes.  KRS_GEMM_TN128=1 selects the 128 x 128
twin (4 waves, <= 128 VGPRs, 64 KB of LDS, two workgroups per CU at most).  One JSON line per case."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from keras_rs_amd import dense_ops as D

dev = torch.device("cuda:0")
B, d, P = 65536, 3456, 512
g = torch.Generator(device=dev).manual_seed(1)
bf = lambda *s: (torch.rand(*s, device=dev, generator=g) - 0.5).to(torch.bfloat16)  # noqa: E731
gr, u, x0, x, dx0acc, dh = bf(B, d), bf(B, d), bf(B, d), bf(B, d), bf(B, d), bf(B, P)
dd = torch.empty(d, P, dtype=torch.float32, device=dev)


def elem():
    D.cross_epilogue_bwd(gr, u, x0, x, 0.0, want_dxd=False, want_dbias=True, dx0_into=dx0acc)


def wgrad():
    D.gemm(x, dh, a_is_km=True, out_dtype=torch.float32, out=dd)


def timed(fns, streams, reps=8):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for s in streams:
            s.wait_stream(torch.cuda.current_stream())
        for fn, s in zip(fns, streams):
            with torch.cuda.stream(s):
                fn()
        for s in streams:
            torch.cuda.current_stream().wait_stream(s)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts[2:]) * 1e3)


sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
tag = "tn128" if os.environ.get("KRS_GEMM_TN128") else "pp256"
print(json.dumps({"gemm": tag, "elementwise_alone_us": timed([elem], [sA]), "wgrad_alone_us": timed([wgrad], [sA]),
                  "both_two_streams_us": timed([elem, wgrad], [sA, sB]), "both_gemm_first_us": timed([wgrad, elem], [sA, sB])}))
