"""Which ops of the whole-model step launch fill / copy kernels?  torch.profiler over three steps, fills and copies with
their input shapes and Python stacks (scripts/exp: development only)."""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "examples"))
import torch
from torch.profiler import ProfilerActivity, profile

import dlrm_dcn_v2 as ex

dev = torch.device("cuda", 0)
hots = [3, 2, 1, 2, 6, 1, 1, 1, 1, 7, 3, 8, 1, 6, 9, 5, 1, 1, 1, 12, 100, 27, 10, 3, 1, 1]
B = 65536
fm = ex.build_model(B, 1_000_000, hots)
x, y = ex.synthetic_batch(B, 13, 1_000_000, hots, dev)
x["large_emb_inputs"] = fm.embedding_layer.preprocess(x["large_emb_inputs"])
box = [None]
for _ in range(4):
    ex.train_step(fm, box, x, y)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    for _ in range(3):
        ex.train_step(fm, box, x, y)
    torch.cuda.synchronize()
rows = []
for e in prof.events():
    if e.name.startswith("aten::") and any(k in e.name for k in ("fill_", "zero_", "zeros", "copy_", "_to_copy", "cat", "add", "mul", "sum", "clone", "contiguous", "full")):
        dt = getattr(e, "device_time_total", 0) or getattr(e, "cuda_time_total", 0)
        if dt > 0:
            rows.append((dt, e.name, str(e.input_shapes)[:80], [s for s in (e.stack or []) if "keras_rs_amd" in s or "dlrm_dcn" in s or "bench" in s][:3]))
rows.sort(key=lambda r: -r[0])
agg = {}
for dt, name, shp, st in rows:
    k = (name, shp, tuple(st))
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1
    a[1] += dt
for (name, shp, st), (n, dt) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
    print(f"{dt / 3:9.1f} us/step  x{n / 3:.1f}  {name}  {shp}")
    for s in st:
        print("             ", s)
