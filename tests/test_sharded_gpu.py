"""World-size-2 (and -8) run of the sharded embedding on the HIP kernels: all ranks share the test box's one
GPU, the collectives go over gloo (host-staged), results are compared with the single-GPU layer."""

import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("kind,exchange,world", [("sgd", "exact", 2), ("adagrad", "exact", 2), ("adam", "exact", 2),
                                                 ("sgd", "static", 2), ("adagrad", "static", 2),
                                                 ("adagrad", "static_prefetch", 2), ("sgd", "static_prefetch", 8),
                                                 # the world of the scaling run's last point: eight owners, the HIP
                                                 # routing / unpack / combine kernels with n_shards = 8
                                                 ("sgd", "static", 8), ("adam", "exact", 8)])
def test_sharded_world2_on_hip_matches_single_gpu_layer(kind, exchange, world):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "_sharded_hip_worker.py"), kind, exchange]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0 and f"SHARDED_HIP_OK {kind}" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
