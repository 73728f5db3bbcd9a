"""Which Python lines launch the ATen kernels inside a bench step (development aid): torch.profiler with stacks over three
eager steps of the C3 model (or the sharded per-rank step with --force-sharded --batch 8192), one line per
(aten op, innermost frames under /root/repo or the snapshot) with its count per step and device time."""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0], "--no-cpu-baseline"] + sys.argv[1:]
import torch
from torch.profiler import ProfilerActivity, profile

import bench

a = bench.parse()
dev = torch.device("cuda", 0)
hots = (bench.ML_PERF_HOTS * 8)[: a.tables]
model = bench.Model(a, hots, 1, 0)
model.embedding.build(None)
box = [None]
ids, dense = bench.make_inputs(a, hots, a.batch, 0, dev)
pre = model.embedding.preprocess(ids)
n_slots = a.tables + 1
scale = 1.0 / (a.batch * n_slots * a.dim)
g_xl = torch.full((a.batch, n_slots * a.dim), scale, dtype=torch.bfloat16, device=dev)
g_inter = torch.full((a.batch, n_slots * (n_slots - 1) // 2), 0.1 * scale, dtype=torch.bfloat16, device=dev)


def step():
    xl, inter = model(dense, pre)
    torch.autograd.backward([xl, inter], [g_xl, g_inter])
    if box[0] is None:
        from keras_rs_amd.optim import Adagrad

        box[0] = Adagrad([p for layer in model.cross for p in layer.parameters()], lr=0.0034, initial_accumulator_value=0.1,
                         prepare_casts=True)
    box[0].step()
    box[0].zero_grad(set_to_none=True)


for _ in range(4):
    step()
torch.cuda.synchronize()
N = 3
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    for _ in range(N):
        step()
    torch.cuda.synchronize()
rows = []
for ev in prof.key_averages(group_by_input_shape=True, group_by_stack_n=8):
    if not ev.key.startswith("aten::") or ev.device_time_total <= 0:
        continue
    frames = [f for f in (ev.stack or []) if ("keras_rs_amd" in f or "bench" in f or "prof_step" in f or "examples" in f)]
    rows.append((ev.self_device_time_total / N, ev.count / N, ev.key, str(ev.input_shapes)[:90],
                 " <- ".join(f.split("/")[-1].strip() for f in frames[:3])))
print("| aten op | calls per step | self device us per step | input shapes | python frames (innermost first) |")
print("|---|---|---|---|---|")
for us, cnt, name, shapes, where in sorted(rows, reverse=True):
    if us > 0:
        print(f"| {name} | {cnt:.1f} | {us:.1f} | {shapes} | {where} |")
