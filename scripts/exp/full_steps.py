"""A few steps of the whole DLRM-DCN-v2 model at the C3 shape (for rocprofv3 --kernel-trace)."""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "examples"))
import torch

import dlrm_dcn_v2 as ex

dev = torch.device("cuda", 0)
hots = [3, 2, 1, 2, 6, 1, 1, 1, 1, 7, 3, 8, 1, 6, 9, 5, 1, 1, 1, 12, 100, 27, 10, 3, 1, 1]
B = 65536
fm = ex.build_model(B, 1_000_000, hots)
x, y = ex.synthetic_batch(B, 13, 1_000_000, hots, dev)
x["large_emb_inputs"] = fm.embedding_layer.preprocess(x["large_emb_inputs"])
box = [None]
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    ex.train_step(fm, box, x, y)
torch.cuda.synchronize()
