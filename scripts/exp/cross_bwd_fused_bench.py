"""Measurement only: krs_gemm_cross_bwd at the C3 shape (M = 65536, N = 3456, K = 512) in the forms the step launches:
R + dx0 accumulate (EPI 4), R + u_upper (EPI 7), no R (EPI 5), no R / no dx0 (EPI 8).  KRS_LIB selects the build."""
import json
import os
import sys

import torch

sys.path.insert(0, ".")
from keras_rs_amd import dense_ops as D  # noqa: E402

dev = "cuda:0"
B, d, p = 65536, 3456, 512
gen = torch.Generator(device=dev).manual_seed(5)
rnd = lambda *sh: (torch.rand(*sh, device=dev, generator=gen) - 0.5).to(torch.bfloat16)  # noqa: E731
A, Bt, R, x0, u, uup, acc = rnd(B, p), rnd(d, p) * 0.2, rnd(B, d), rnd(B, d), rnd(B, d), rnd(B, d), rnd(B, d)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / n * 1e3, 1)


out = {"lib": os.environ.get("KRS_LIB", "product build").split("/")[-2] if os.environ.get("KRS_LIB") else "product build"}
ref = D.gemm_cross_bwd(A, Bt, R, x0, u, dx0_into=acc.clone())
out["checksum"] = [float(t.float().sum()) for t in ref[:3]]
for rep in range(2):
    out[f"epi4_us_{rep}"] = timed(lambda: D.gemm_cross_bwd(A, Bt, R, x0, u, dx0_into=acc))
    out[f"epi7_us_{rep}"] = timed(lambda: D.gemm_cross_bwd(A, Bt, R, x0, u, u_upper=uup))
    out[f"epi5_us_{rep}"] = timed(lambda: D.gemm_cross_bwd(A, Bt, None, x0, u))
    out[f"epi8_us_{rep}"] = timed(lambda: D.gemm_cross_bwd(A, Bt, None, x0, u, want_dx0=False))
print(json.dumps(out))
