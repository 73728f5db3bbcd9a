"""N > 1 path on CPU: world_size-2 gloo run of the MOD-sharded embedding exchange
(keras_rs_amd/sharded.py) against the unsharded oracle, forward and fused-optimizer backward."""

import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("kind,weighted,world,exchange", [
    ("sgd", "w", 2, "exact"), ("adagrad", "w", 2, "exact"), ("adagrad", "now", 2, "exact"), ("sgd", "now", 3, "exact"),
    ("adam", "w", 2, "exact"), ("ftrl", "now", 2, "exact"), ("adagrad", "wragged", 2, "exact"),
    ("sgd", "nowragged", 2, "exact"),
    # static-capacity exchange (no size ever reaches the host): sized automatically, from the TableConfig limits, and
    # from a capacity the first steps overflow (ids dropped + flagged, limits learnt from the running statistics)
    ("adagrad", "w", 2, "static"), ("sgd", "now", 3, "static"), ("adam", "w", 2, "static_cfg"),
    ("adagrad", "w", 2, "static_tiny"), ("sgd", "now", 3, "static_tiny"), ("adagrad", "wragged", 2, "static"),
    # round 4: the id side of the lookup issued ahead of the call (prefetch()), the segment gradients gathered straight
    # out of the slab gradient (no copy), capacities that shrink to the settled statistics
    ("adagrad", "w", 2, "static_prefetch"), ("sgd", "now", 3, "static_prefetch"), ("adagrad", "now", 2, "static_shrink"),
    # the world the scaling run ends on (8 owners, every bag split over up to 8 partial sums)
    ("adagrad", "w", 8, "static"), ("adagrad", "nowragged", 8, "exact")])
def test_sharded_embedding_matches_unsharded_oracle(kind, weighted, world, exchange):
    # weighted: user weights on every feature; "now": none (mean / sqrtn scales are still folded in);
    # "...ragged": bags of varying length (some empty) given as Ragged values + row offsets
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "_sharded_worker.py"), kind, weighted, exchange]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0 and f"SHARDED_OK {kind}" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
