"""Per-call timing of the FeatureCross GEMMs / elementwise kernels at the C3 shape (development aid)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from keras_rs_amd import dense_ops as D

dev = torch.device("cuda:0")
B, d, p = 65536, 3456, 512
dt = torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: (torch.rand(*s, device=dev, generator=g) - 0.5).to(dt)
x0, x, gy = rn(B, d), rn(B, d), rn(B, d)
U, V = rn(d, p) * 0.05, rn(p, d) * 0.05
bias = torch.zeros(d, device=dev)
Ut, Vt = U.t().contiguous(), V.t().contiguous()
h, _ = D.gemm(x, Ut, b_is_nk=True)
y, u = D.gemm(h, Vt, b_is_nk=True, bias=bias, x0=x0, x=x, want_u=True)
dz, dx0, _, db = D.cross_epilogue_bwd(gy, u, x0, x, 0.0, want_dxd=False)
dh, _ = D.gemm(dz, V, b_is_nk=True)


def t(fn, n=10):
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


F = 2.0 * B * d * p
MB = 1e6
cases = [
    ("fwd1  h = x U        (K=3456,N=512)", lambda: D.gemm(x, Ut, b_is_nk=True), F, (B * d + B * p) * 2),
    ("fwd2  y = cross(hV)  (K=512,N=3456)", lambda: D.gemm(h, Vt, b_is_nk=True, bias=bias, x0=x0, x=x, want_u=True), F,
     (B * p + 4 * B * d) * 2),
    ("bwd   elementwise dz, dx0", lambda: D.cross_epilogue_bwd(gy, u, x0, x, 0.0, want_dxd=False), 0, 6 * B * d * 2),
    ("dK    = h^T dz       (split-K)", lambda: D.gemm(h, dz, a_is_km=True, out_dtype=torch.float32), F, (B * p + B * d) * 2),
    ("dh    = dz V^T       (K=3456,N=512)", lambda: D.gemm(dz, V, b_is_nk=True), F, (B * d + B * p) * 2),
    ("dU    = x^T dh       (split-K)", lambda: D.gemm(x, dh, a_is_km=True, out_dtype=torch.float32), F, (B * d + B * p) * 2),
    ("dx    = dh U^T + g   (K=512,N=3456)", lambda: D.gemm(dh, U, b_is_nk=True, r=gy, beta=1.0), F, (B * p + 2 * B * d) * 2),
]
tot = 0
for name, fn, fl, by in cases:
    us = t(fn)
    tot += us
    print(f"{name:42s} {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s  {by / us / 1e3:7.1f} GB/s (min HBM bytes)")
print(f"layer fwd+bwd total {tot:.1f} us")
# the K = 512 product without its epilogue traffic (plain bf16 store): what the epilogues cost
us = t(lambda: D.gemm(h, Vt, b_is_nk=True))
print(f"{'      y = h V      plain (K=512,N=3456)':42s} {us:8.1f} us  {F / us / 1e6:7.1f} TF/s")
us = t(lambda: D.cross_epilogue_fwd(u, x0, x, 0.0))
print(f"{'      cross epilogue alone (4 x [B,d])':42s} {us:8.1f} us  {4 * B * d * 2 / us / 1e3:7.1f} GB/s")
