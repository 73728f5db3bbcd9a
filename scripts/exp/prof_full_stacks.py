"""Which Python lines launch the ATen kernels inside a step of the WHOLE ml_perf model (examples/dlrm_dcn_v2.py at the C3 shape):
torch.profiler with stacks over three eager steps, one line per (aten op, innermost repo frames) -- development aid."""
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
N = 3
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    for _ in range(N):
        ex.train_step(fm, box, x, y)
    torch.cuda.synchronize()
rows = []
for ev in prof.key_averages(group_by_input_shape=True, group_by_stack_n=8):
    if not ev.key.startswith("aten::") or ev.device_time_total <= 0:
        continue
    frames = [f for f in (ev.stack or []) if ("keras_rs_amd" in f or "examples" in f or "prof_full" in f)]
    rows.append((ev.self_device_time_total / N, ev.count / N, ev.key, str(ev.input_shapes)[:90],
                 " <- ".join(f.split("/")[-1].strip() for f in frames[:3])))
print("| aten op | calls per step | self device us per step | input shapes | python frames (innermost first) |")
print("|---|---|---|---|---|")
for us, cnt, name, shapes, where in sorted(rows, reverse=True):
    if us > 0:
        print(f"| {name} | {cnt:.1f} | {us:.1f} | {shapes} | {where} |")
