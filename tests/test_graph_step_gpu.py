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


@pytest.mark.parametrize("which,optimizer", [("single", "adam"), ("single", "sched"), ("sharded", "adam"), ("sharded", "sched")])
def test_replayed_steps_with_step_dependent_optimizer_constants_equal_eager_steps(which, optimizer):
    """Round-5 review, weak #8: Adam's bias correction and scheduled learning rates are computed on the host once per update.
    Frozen into a captured launch they made every replay apply step 1's constants, silently.  They now live in device memory
    (embedding_ops.StepConstants; krs_store_f32 / krs_embed_bag_bwd_fused_adam_dyn), GraphedStep refreshes them before every
    replay, and a bare torch.cuda.graph capture of such a step raises.  2 eager + 3 replayed updates leave the bits (tables,
    both Adam moments, dense weights) and the update count of 5 eager updates."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_graph_worker.py"), which, optimizer], capture_output=True,
                       text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0 and f"GRAPH_OK {which} {optimizer}" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


@pytest.mark.parametrize("which,optimizer", [("single", "adam"), ("sharded", "sched")])
def test_a_bare_capture_of_step_dependent_constants_is_refused(which, optimizer):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_graph_worker.py"), which, optimizer, "bare"], capture_output=True,
                       text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0 and f"GRAPH_REFUSED {optimizer}" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_a_forward_only_capture_of_the_sharded_layer_joins_its_plan_stream():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_graph_worker.py"), "sharded", "adagrad", "fwdonly"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0 and "GRAPH_FWD_ONLY_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
