"""Per-call timing of DotInteraction fwd / bwd at the C3 shape (development aid)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from keras_rs_amd import dense_ops as D

dev = torch.device("cuda:0")
B, F, Dm = 65536, 27, 128
dt = torch.bfloat16
buf = (torch.rand(B, F * Dm, device=dev) - 0.5).to(dt)
feats = [buf[:, f * Dm:(f + 1) * Dm] for f in range(F)]
out = D.dot_interaction_fwd(feats)
g = (torch.rand_like(out.float()) - 0.5).to(dt)


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


by = B * F * Dm * 2
print(f"fwd {t(lambda: D.dot_interaction_fwd(feats)):.1f} us  (bytes {by + out.numel() * 2:.3e})")
us = t(lambda: D.dot_interaction_bwd(feats, g))
print(f"bwd {us:.1f} us  -> {(2 * by + out.numel() * 2) / us / 1e3:.0f} GB/s")
