import os, sys, json
sys.path.insert(0, os.getcwd())
import torch
from keras_rs_amd.embedding_ops import FusedBags
HOTS = [3, 2, 1, 2, 6, 1, 1, 1, 1, 7, 3, 8, 1, 6, 9, 5, 1, 1, 1, 12, 100, 27, 10, 3, 1, 1]
T, V, D, B = 26, 1_000_000, 128, 65536
dev = torch.device("cuda:0"); g = torch.Generator(device=dev).manual_seed(1)
tables = [(torch.rand(V, D, device=dev, generator=g) * 0.1 - 0.05).to(torch.bfloat16) for _ in range(T)]
fb = FusedBags(tables, [(t, "sum", t * D) for t in range(T)])
ids = torch.cat([torch.randint(0, V, (B * h,), device=dev, generator=g, dtype=torch.int32) for h in HOTS])
grad = (torch.rand(B, T * D, device=dev, generator=g) * 1e-3).to(torch.bfloat16)
nnz = ids.numel()
out = [torch.zeros(V, D, device=dev) for _ in range(T)]
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
ws = fb.plan_backward(ids, B, hots=HOTS, global_order=False)
d_us = t(lambda: fb.backward_dense(ws, grad, B, nnz, hots=HOTS, out=out))
wsg = fb.plan_backward(ids, B, hots=HOTS)
s_us = t(lambda: fb.backward_sparse(wsg, grad, B, nnz, hots=HOTS))
print(json.dumps({"lib": os.environ.get("KRS_LIB", "default")[-30:], "dense_us": d_us, "sparse_us": s_us}))
