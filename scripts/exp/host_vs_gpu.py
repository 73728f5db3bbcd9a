"""How much of the sharded step at B_local = 8192 is host time?  Enqueue time per step (no synchronisation inside)
against the synchronised step time, plus a cProfile of the enqueue path."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.argv = [sys.argv[0], "--no-cpu-baseline", "--force-sharded", "--batch", sys.argv[1] if len(sys.argv) > 1 else "8192",
            "--exchange", sys.argv[2] if len(sys.argv) > 2 else "static"]
import torch

import bench

a = bench.parse()
dev = torch.device("cuda", 0)
hots = (bench.ML_PERF_HOTS * 8)[: a.tables]
model = bench.Model(a, hots, 1, 0)
model.embedding.build(None)
box = [None]
bench.measure(model, a, hots, 1, 0, dev, a.batch, 2, 3, box)
r = bench.measure(model, a, hots, 1, 0, dev, a.batch, 30, 3, box)
print("synchronised ms_per_step %.3f, host enqueue %.3f ms per step (unprofiled)" % (r["elapsed"] / 30 * 1e3, r["enqueue_s"] / 30 * 1e3))
# host-only: profile the python side of 20 steps
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
bench.measure(model, a, hots, 1, 0, dev, a.batch, 20, 0, box)
pr.disable()
print("wall per step incl. final sync %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
