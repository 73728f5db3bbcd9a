"""bench.py's N > 1 path end to end on the test box's one GPU: two ranks launched as the driver launches them
(torch.distributed.run), sharing cuda:0, collectives over gloo staged through the host (--dist-backend gloo is the
debugging switch for exactly this; production uses nccl = RCCL).  Checks that the run completes and that rank 0
prints the contract's JSON line for the sharded + data-parallel configuration."""

import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_bench_run_over_gloo():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo",
           "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--batch", "4096", "--vocab", "50000"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "strong"
    assert d["unit"] == "lookups/s" and d["value"] > 0 and d["ms_per_step"] > 0 and d["higher_is_better"] is True
    assert "row-sharded over 2 GPUs" in d["config"]["parallelism"] and d["config"]["global_batch"] == 4096
    assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1.5
