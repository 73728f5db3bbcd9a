"""Worker of tests/test_dp_gloo.py: two ranks, different data, gradients averaged by GradAllReduce."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keras_rs_amd.dp import GradAllReduce  # noqa: E402

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.manual_seed(0)
w1 = torch.nn.Parameter(torch.randn(6, 4))
w2 = torch.nn.Parameter(torch.randn(4, 3))
frozen = torch.nn.Parameter(torch.randn(3), requires_grad=False)
sync = GradAllReduce([w1, w2, frozen])
for step in range(2):
    x = torch.randn(5, 6, generator=torch.Generator().manual_seed(10 * step + rank))
    loss = ((x @ w1).relu() @ w2 + frozen).square().sum()
    loss.backward()
    sync.wait()
    # expectation: every rank recomputes all ranks' gradients on copies of the weights
    exp1, exp2 = torch.zeros_like(w1), torch.zeros_like(w2)
    for r in range(world):
        a, b = w1.detach().clone().requires_grad_(True), w2.detach().clone().requires_grad_(True)
        xr = torch.randn(5, 6, generator=torch.Generator().manual_seed(10 * step + r))
        ((xr @ a).relu() @ b + frozen).square().sum().backward()
        exp1 += a.grad / world
        exp2 += b.grad / world
    torch.testing.assert_close(w1.grad, exp1, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(w2.grad, exp2, rtol=1e-6, atol=1e-6)
    with torch.no_grad():
        w1 -= 0.01 * w1.grad
        w2 -= 0.01 * w2.grad
    w1.grad = w2.grad = None
assert not sync._pending
sync.remove()
dist.barrier()
if rank == 0:
    print("DP_OK")
dist.destroy_process_group()
