"""Measurement only: which kernels the vendor GEMM library picks for the long-K products of a C3 cross layer (run under
rocprofv3 --kernel-trace --stats; the kernel names spell out macro tile, MFMA shape, depth, workgroup and prefetch options)."""
import torch
dev = "cuda:0"
B, d, p = 65536, 3456, 512
g = torch.Generator(device=dev).manual_seed(3)
rnd = lambda *sh: (torch.rand(*sh, device=dev, generator=g) - 0.5).to(torch.bfloat16)  # noqa: E731
x, dz, h, dh, U, K = rnd(B, d), rnd(B, d), rnd(B, p), rnd(B, p), rnd(d, p), rnd(p, d)
for _ in range(5):
    torch.matmul(x, U)          # h  (NN)
    torch.matmul(dz, K.t())     # dh (NT)
    torch.matmul(h, K)          # z
    torch.matmul(dh, U.t())     # dx
torch.cuda.synchronize()
