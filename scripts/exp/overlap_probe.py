"""Can the HBM-bound table update (K2 apply, fused Adagrad) run UNDER the weight-gradient GEMMs of the cross
stack (operand-path bound, ~2.5 TB/s of HBM)?  Times, at the C3 shapes: apply alone, the 6 weight-gradient GEMMs
alone, both on two streams -- plain streams, and streams created with disjoint CU masks
(hipExtStreamCreateWithCUMask) in several splits.  One JSON line per case."""

import ctypes as C
import json
import sys

import numpy as np
import torch

from keras_rs_amd import dense_ops as D
from keras_rs_amd.embedding_ops import FusedBags

dev = torch.device("cuda:0")
B, T, V, DIM, P = 65536, 26, 1_000_000, 128, 512
d = (T + 1) * DIM
HOTS = [3, 2, 1, 2, 6, 1, 1, 1, 1, 7, 3, 8, 1, 6, 9, 5, 1, 1, 1, 12, 100, 27, 10, 3, 1, 1]
g = torch.Generator(device=dev).manual_seed(1)
tables = [(torch.rand(V, DIM, device=dev, generator=g) * 0.1 - 0.05).to(torch.bfloat16) for _ in range(T)]
ids = torch.cat([torch.randint(0, V, (B * h,), device=dev, generator=g, dtype=torch.int32) for h in HOTS])
fb = FusedBags(tables, [(t, "sum", t * DIM) for t in range(T)])
fb.slots = [torch.full(t.shape, 0.1, dtype=torch.float32, device=dev) for t in tables]
fb.lrs = [0.0034] * T
grad = (torch.rand(B, T * DIM, device=dev) * 1e-3).to(torch.bfloat16)
nnz = ids.numel()
ws = fb.plan_backward(ids, B, hots=HOTS)

bf = lambda *s: (torch.rand(*s, device=dev, generator=g) - 0.5).to(torch.bfloat16)  # noqa: E731
h, dz, x, dh = bf(B, P), bf(B, d), bf(B, d), bf(B, P)
dk = torch.empty(P, d, dtype=torch.float32, device=dev)
dd = torch.empty(d, P, dtype=torch.float32, device=dev)


def apply():
    fb.backward_fused("adagrad", ws, grad, B, nnz, hots=HOTS)


def wgrads():
    for _ in range(3):
        D.gemm(h, dz, a_is_km=True, out_dtype=torch.float32, out=dk)
        D.gemm(x, dh, a_is_km=True, out_dtype=torch.float32, out=dd)


hip = C.CDLL("libamdhip64.so")


def masked_stream(lo, hi, total=256):
    """A stream restricted to CU-mask bits lo..hi-1."""
    words = (total + 31) // 32
    m = (C.c_uint32 * words)()
    for b in range(lo, hi):
        m[b // 32] |= 1 << (b % 32)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), C.c_uint32(words), m)
    if rc:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask -> {rc}")
    return torch.cuda.ExternalStream(s.value, device=dev)


def timed(fa, sa, fb_, sb, reps=5):
    """Wall time (events on the default stream) of fa on stream sa and fb_ on stream sb, started together."""
    main = torch.cuda.current_stream()
    out = []
    for i in range(reps + 2):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main)
        for f, s in ((fa, sa), (fb_, sb)):
            if f is None:
                continue
            s.wait_stream(main)
            with torch.cuda.stream(s):
                f()
        for f, s in ((fa, sa), (fb_, sb)):
            if f is not None:
                main.wait_stream(s)
        e1.record(main)
        torch.cuda.synchronize()
        if i >= 2:
            out.append(e0.elapsed_time(e1))
    return float(np.median(out))


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
res = {"apply_alone_ms": timed(apply, s1, None, None), "wgrads_alone_ms": timed(wgrads, s1, None, None)}
print(json.dumps(res), flush=True)
print(json.dumps({"case": "two plain streams", "both_ms": timed(wgrads, s1, apply, s2)}), flush=True)
print(json.dumps({"case": "two plain streams, apply first", "both_ms": timed(apply, s1, wgrads, s2)}), flush=True)
hi_s = torch.cuda.Stream(priority=-1)
print(json.dumps({"case": "gemm on a high-priority stream", "both_ms": timed(wgrads, hi_s, apply, s2)}), flush=True)
for n_gemm in (224, 192, 160, 128):
    try:
        sg, sa = masked_stream(0, n_gemm), masked_stream(n_gemm, 256)
    except Exception as e:  # noqa: BLE001
        print(json.dumps({"case": f"cu mask {n_gemm}/{256 - n_gemm}", "error": str(e)}), flush=True)
        continue
    r = {"case": f"cu mask: gemm bits 0..{n_gemm - 1}, apply bits {n_gemm}..255",
         "gemm_alone_masked_ms": timed(wgrads, sg, None, None), "apply_alone_masked_ms": timed(apply, sa, None, None),
         "both_ms": timed(wgrads, sg, apply, sa)}
    print(json.dumps(r), flush=True)
# interleaved mask: every 4th bit to the apply
words = 8
ma, mg = (C.c_uint32 * words)(), (C.c_uint32 * words)()
for b in range(256):
    (ma if b % 4 == 3 else mg)[b // 32] |= 1 << (b % 32)
sa_, sg_ = C.c_void_p(), C.c_void_p()
if not hip.hipExtStreamCreateWithCUMask(C.byref(sa_), C.c_uint32(words), ma) and \
        not hip.hipExtStreamCreateWithCUMask(C.byref(sg_), C.c_uint32(words), mg):
    sa, sg = torch.cuda.ExternalStream(sa_.value, device=dev), torch.cuda.ExternalStream(sg_.value, device=dev)
    print(json.dumps({"case": "cu mask interleaved 3:1", "gemm_alone_masked_ms": timed(wgrads, sg, None, None),
                      "apply_alone_masked_ms": timed(apply, sa, None, None), "both_ms": timed(wgrads, sg, apply, sa)}),
          flush=True)
sys.exit(0)
