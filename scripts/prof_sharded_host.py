"""Host-side view of the sharded step at the per-rank batch of an 8-GPU run (8192): the step is
launch-bound there, so the table is sorted by CPU time (development aid)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0], "--no-cpu-baseline", "--force-sharded", "--batch", "8192"]
import torch
from torch.profiler import ProfilerActivity, profile

import bench

a = bench.parse()
dev = torch.device("cuda", 0)
hots = (bench.ML_PERF_HOTS * 8)[: a.tables]
model = bench.Model(a, hots, 1, 0)
model.embedding.build(None)
box = [None]
bench.measure(model, a, hots, 1, 0, dev, a.batch, 1, 3, box)
el, _ = bench.measure(model, a, hots, 1, 0, dev, a.batch, 30, 3, box)
print("ms_per_step %.3f" % (el / 30 * 1e3))
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    bench.measure(model, a, hots, 1, 0, dev, a.batch, 5, 0, box)
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=45, max_name_column_width=60))
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=40, max_name_column_width=90))
