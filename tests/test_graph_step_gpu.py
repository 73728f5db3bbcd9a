"""A training step replayed from a HIP graph (keras_rs_amd.graphs.GraphedStep) leaves the same bits in every table,
optimizer slot and dense weight as the same number of eager steps: the C ABI only enqueues kernels and memsets on the
stream it is given, the compute-dtype copies of the weights live at fixed addresses, and flags / exchange statistics are
read between replays.  Both embedding layers: the single-GPU DistributedEmbedding and the sharded layer at world 1
(static-capacity exchange: no size reaches the host).  Each case runs in its own process (tests/_graph_worker.py)."""

import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# "sharded_rccl" (round 4): the sharded layer's three all-to-alls and the dense all-reduce run through a one-rank RCCL
# communicator INSIDE the capture -- RCCL's kernels and torch's stream hand-offs replay with the step
@pytest.mark.parametrize("which", ["single", "sharded", "sharded_rccl"])
def test_replayed_steps_equal_eager_steps(which):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_graph_worker.py"), which], capture_output=True,
                       text=True, timeout=240 if which == "sharded_rccl" else 600, cwd=ROOT)
    assert r.returncode == 0 and f"GRAPH_OK {which}" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
