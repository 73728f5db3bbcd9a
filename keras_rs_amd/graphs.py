"""A training step replayed from a HIP graph (torch.cuda.CUDAGraph on ROCm = hipGraph): for steps that are bound by the
host's enqueue rate -- the per-rank step of a strongly-scaled job (batch 8192 per GPU: ~250 launches in 2.4 ms) -- one
`hipGraphLaunch` replaces the Python / ctypes work of the whole step.

What makes the path capturable: the C ABI only ever enqueues kernels and memsets on the stream it is given (no copies to
the host, no allocation, no synchronisation: include/krs.h), the static-capacity exchange has no size that reaches the
host (keras_rs_amd/sharded.py), out-of-range-id flags and exchange statistics are copied to page-locked memory by nodes
of the graph and looked at BETWEEN replays (`check_ids(wait=True)`, `poll_exchange_stats()`), and the compute-dtype
copies of the dense weights live at fixed addresses (dense_ops.refresh_casts).

Limits, all of the capture mechanism: shapes and ids buffers are fixed (new ids are copied into the captured input
tensors), scalars the host computes per step are frozen into the kernel arguments, and a capacity that grows means a
new capture.  The optimizer constants that change per step -- scheduled learning rates, Adam's bias correction -- are
therefore NOT kernel arguments: they live in device memory (embedding_ops.StepConstants), the captured launches read
them when they run, and `GraphedStep.__call__` rewrites them before every replay (`before_each_replay` hooks: a kernel
whose arguments carry the new values, krs_store_f32, enqueued on the replay's stream).  A step-dependent constant met
during a capture that is not GraphedStep's raises instead of replaying stale values (round-5 review, weak #8).

Validated (tests/test_graph_step_gpu.py, `bench.py --graph`): replays leave the same bits as eager steps, for the
single-GPU layer and for the sharded layer at world 1; the sharded per-rank step (batch 8192) was captured twice in one
process.  Known problem, ROCm 7.2: a SECOND capture of the large-batch single-GPU step (batch 65536, which forks into the
plan and weight-gradient streams) in the same process crashed inside hipStreamEndCapture -- with new side streams and with
the first graph kept alive alike; `bench.py --graph` therefore graphs its primary leg only.
Collectives (round 4): the sharded layer's all-to-alls and the dense all-reduce run INSIDE the capture -- validated through a
one-rank RCCL communicator (`tests/test_graph_step_gpu.py::sharded_rccl`: replays leave the bits of eager steps;
`bench.py --force-sharded --rccl-self --graph`: 2.61 ms per replayed step against 2.81-3.00 ms eager at the per-rank batch of
8192, host enqueue 0.24 ms).  RCCL's kernels and torch's hand-offs between its collective stream and the step's stream are
ordinary graph nodes; with N > 1 every rank captures and replays the same graph (one GPU per box here: not run)."""

from __future__ import annotations

import torch


_capturing: "GraphedStep | None" = None


def before_each_replay(fn) -> None:
    """Registers `fn` (host work of one optimizer update: StepConstants.advance) to run before every replay of the step
    being captured.  Only valid while GraphedStep is capturing: any other capture has nobody to call it."""
    if _capturing is None:
        from keras_rs_amd import _lib as L

        raise L.KrsError("a fused optimizer with step-dependent constants (Adam, or a learning-rate schedule) is being "
                         "captured into a HIP graph outside keras_rs_amd.graphs.GraphedStep: every replay would apply the "
                         "constants of the captured step; capture the step with GraphedStep, which refreshes them")
    if all(h != fn for h in _capturing.hooks):      # (bound methods of one object compare equal)
        _capturing.hooks.append(fn)


def join_at_capture_end(side_stream) -> None:
    """A side stream forked inside the step that the step itself may not join (the sharded layer's plan stream is joined by
    the BACKWARD pass: a captured forward-only step with gradients enabled would end its capture with the stream still
    forked, which hipStreamEndCapture refuses).  GraphedStep makes the capturing stream wait for it after the
    step; outside its capture this is a no-op (an eager stream needs no join)."""
    if _capturing is not None and torch.cuda.is_current_stream_capturing() and all(s is not side_stream for s in _capturing.joins):
        _capturing.joins.append(side_stream)


def count_update(owner) -> None:
    """One fused update of `owner` (an object with a `step` count) whose constants do not depend on the count: counted now, or
    -- inside GraphedStep's capture, which runs nothing -- once per replay, so that the checkpointed `iterations` follows the
    updates actually applied.  (A bare torch.cuda.graph capture counts the captured step once, as before.)"""
    if _capturing is not None and torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
        if not any(getattr(h, "_krs_counts", None) is owner for h in _capturing.hooks):
            def bump(owner=owner):
                owner.step += 1
            bump._krs_counts = owner
            _capturing.hooks.append(bump)
    else:
        owner.step += 1


class GraphedStep:
    """`step` (a callable without arguments that runs forward, backward and the optimizer on fixed tensors) captured once
    and replayed.  `warmup` eager calls run first on a side stream, as torch.cuda.graph asks: they build the layers, the
    optimizer state and every cached descriptor."""

    def __init__(self, step, warmup: int = 3):
        self.graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            # let ProcessGroupNCCL's watchdog (a 100 ms polling loop) retire the warm-up steps' collectives before the
            # capture opens: an event query of its that meets the capture invalidates it (hipErrorCapturedEvent, seen once
            # in 8 runs; the caller then falls back to the eager step)
            import time

            time.sleep(0.5)
        # capture_error_mode="thread_local": only THIS thread's calls are held to the capture rules.  With the default
        # ("global") an event query from any other thread while the capture is open is an error that aborts the process --
        # and ProcessGroupNCCL's watchdog thread polls the events of the eager warm-up steps' collectives for a few
        # milliseconds more (seen once in ~8 runs of `bench.py --force-sharded --rccl-self`: the bench died inside
        # WorkNCCL::finishedGPUExecutionInternal before it could print its line; round 5).
        global _capturing
        self.hooks: list = []          # host work per replay (step-dependent optimizer constants)
        self.joins: list = []          # side streams forked inside the step: joined before the capture ends
        _capturing = self
        try:
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                step()
                for side_stream in self.joins:
                    torch.cuda.current_stream().wait_stream(side_stream)
        finally:
            _capturing = None

    def __call__(self) -> None:
        for hook in self.hooks:
            hook()
        self.graph.replay()
