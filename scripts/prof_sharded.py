"""torch.profiler view of the sharded (world-of-1 dry run) step (development aid)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0], "--no-cpu-baseline", "--force-sharded"]
import torch
from torch.profiler import ProfilerActivity, profile

import bench

a = bench.parse()
dev = torch.device("cuda", 0)
hots = (bench.ML_PERF_HOTS * 8)[: a.tables]
model = bench.Model(a, hots, 1, 0)
model.embedding.build(None)
box = [None]
bench.measure(model, a, hots, 1, 0, dev, a.batch, 1, 2, box)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    bench.measure(model, a, hots, 1, 0, dev, a.batch, 3, 0, box)
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=60, max_name_column_width=70))
