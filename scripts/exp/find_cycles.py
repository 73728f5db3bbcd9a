"""Which reference cycles does a training step leave behind?  (bench.py switches the cyclic collector off inside its timed
region: cyclic garbage that holds device tensors then grows the allocator's reservation step by step.)"""
import gc
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench

a = bench.parse()
a.batch, a.vocab, a.no_cpu_baseline = 8192, 100000, True
a.vocabs = [a.vocab] * a.tables
hots = (bench.ML_PERF_HOTS * 8)[: a.tables]
dev = torch.device("cuda", 0)
model = bench.Model(a, hots, 1, 0)
model.embedding.build(None)
ids, dense = bench.make_inputs(a, hots, a.batch, 0, dev)
pre = model.embedding.preprocess(ids)
g_xl = torch.full((a.batch, (a.tables + 1) * a.dim), 1e-6, dtype=torch.bfloat16, device=dev)
g_in = torch.full((a.batch, (a.tables + 1) * a.tables // 2), 1e-7, dtype=torch.bfloat16, device=dev)


def step():
    xl, inter = model(dense, pre)
    torch.autograd.backward([xl, inter], [g_xl, g_in])
    for p in model.cross.parameters():
        p.grad = None


for _ in range(3):
    step()
gc.collect()
gc.disable()
gc.set_debug(gc.DEBUG_SAVEALL)
for _ in range(2):
    step()
torch.cuda.synchronize()
n = gc.collect()
print("unreachable objects after 2 steps:", n)
tens = [o for o in gc.garbage if isinstance(o, torch.Tensor)]
print("tensors in garbage:", [(tuple(t.shape), t.dtype) for t in tens][:20])
kinds = {}
for o in gc.garbage:
    kinds[type(o).__name__] = kinds.get(type(o).__name__, 0) + 1
print(sorted(kinds.items(), key=lambda kv: -kv[1])[:25])
for t in tens[:4]:
    print("--- referrers of", tuple(t.shape))
    for r in gc.get_referrers(t):
        if r is gc.garbage or isinstance(r, types.FrameType):
            continue
        desc = type(r).__name__
        if isinstance(r, dict):
            desc += " keys=" + str(list(r.keys())[:8])
        elif isinstance(r, (tuple, list)):
            desc += " len=%d types=%s" % (len(r), [type(x).__name__ for x in r][:8])
        print("   ", desc)
