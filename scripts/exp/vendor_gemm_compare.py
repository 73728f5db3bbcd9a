"""Measurement only (NOT product code, nothing in keras_rs_amd imports this): the six products of one C3 FeatureCross layer
(B = 65536, d = 3456, p = 512, bf16 in, fp32 accumulate) on krs_gemm against torch.matmul -- the vendor GEMM library of this
ROCm build (hipBLASLt / rocBLAS behind torch) -- as an outside yardstick for "how far from what this chip gives such shapes".
Same random operands, same layouts the step uses, HIP events around 20 launches each, alternating, two rounds.

    python scripts/exp/vendor_gemm_compare.py            (on a GPU box)
"""
import sys

import torch

sys.path.insert(0, ".")
from keras_rs_amd import dense_ops as D  # noqa: E402

dev = "cuda:0"
B, d, p = 65536, 3456, 512
g = torch.Generator(device=dev).manual_seed(3)


def rnd(*sh):
    return (torch.rand(*sh, device=dev, generator=g) - 0.5).to(torch.bfloat16)


x, dz = rnd(B, d), rnd(B, d)
h, dh = rnd(B, p), rnd(B, p)
U, Ut = rnd(d, p), rnd(p, d)      # U [d, p] and its K-contiguous copy for h = x U  (Bt [N = p, K = d])
K, Kt = rnd(p, d), rnd(d, p)      # K [p, d] and its K-contiguous copy for y = h K  (Bt [N = d, K = p])


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
    return e0.elapsed_time(e1) / n * 1e3


out_bd, out_bp = torch.empty(B, d, device=dev, dtype=torch.bfloat16), torch.empty(B, p, device=dev, dtype=torch.bfloat16)
out_pd, out_dp = torch.empty(p, d, device=dev, dtype=torch.bfloat16), torch.empty(d, p, device=dev, dtype=torch.bfloat16)
# (the weight gradients leave krs_gemm in fp32 -- split-K slabs reduced in a fixed order; the vendor call writes bf16: less to store)
cases = [
    ("h  = x U        [B,d]x[d,p]", lambda: D.gemm(x, Ut, b_is_nk=True), lambda: torch.matmul(x, U, out=out_bp)),
    ("z  = h K        [B,p]x[p,d]", lambda: D.gemm(h, Kt, b_is_nk=True), lambda: torch.matmul(h, K, out=out_bd)),
    ("dK = h^T dz     [p,B]x[B,d]", lambda: D.gemm(h, dz, a_is_km=True, out_dtype=torch.float32),
     lambda: torch.matmul(h.t(), dz, out=out_pd)),
    ("dh = dz K^T     [B,d]x[d,p]", lambda: D.gemm(dz, K, b_is_nk=True), lambda: torch.matmul(dz, K.t(), out=out_bp)),
    ("dU = x^T dh     [d,B]x[B,p]", lambda: D.gemm(x, dh, a_is_km=True, out_dtype=torch.float32),
     lambda: torch.matmul(x.t(), dh, out=out_dp)),
    ("dx = dh U^T     [B,p]x[p,d]", lambda: D.gemm(dh, U, b_is_nk=True), lambda: torch.matmul(dh, U.t(), out=out_bd)),
]
flops = 2.0 * B * d * p
print(f"torch {torch.__version__}  hip {torch.version.hip}  preferred BLAS: {torch.backends.cuda.preferred_blas_library()}")
for rep in range(2):
    print(f"-- round {rep}")
    for name, ours, vendor in cases:
        a, b = timed(ours), timed(vendor)
        print(f"{name}   krs_gemm {a:7.1f} us ({flops / a / 1e6:6.0f} TF/s)   torch.matmul {b:7.1f} us ({flops / b / 1e6:6.0f} TF/s)")
