"""torch.profiler view of the full DLRM-DCN-v2 example step (development aid)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
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
for _ in range(3):
    ex.train_step(fm, box, x, y)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(2):
        ex.train_step(fm, box, x, y)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=25, max_name_column_width=90))
