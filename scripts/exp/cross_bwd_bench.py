"""Measurement only: the elementwise backward of a cross layer (krs_cross_epilogue_bwd) at the C3 shape, in the two forms the
step uses -- dz only (top layer of a stack: g, x0 in; dz out) and the full pass (g, u, x0, dL/dx0 in; dz, dL/dx0 out)."""
import json
import os
import sys

import torch

sys.path.insert(0, ".")
from keras_rs_amd import dense_ops as D  # noqa: E402

dev = "cuda:0"
B, d = 65536, 3456
gen = torch.Generator(device=dev).manual_seed(5)
g, u, x0, acc = ((torch.rand(B, d, device=dev, generator=gen) - 0.5).to(torch.bfloat16) for _ in range(4))


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


out = {"lib": os.environ.get("KRS_LIB", "product build")}
for rep in range(2):
    out[f"dz_only_us_{rep}"] = timed(lambda: D.cross_epilogue_bwd(g, None, x0, x0, 0.0, want_dxd=False, want_dbias=True, want_dx0=False))
    out[f"full_us_{rep}"] = timed(lambda: D.cross_epilogue_bwd(g, u, x0, x0, 0.0, want_dxd=False, want_dbias=True, dx0_into=acc))
print(json.dumps(out))
