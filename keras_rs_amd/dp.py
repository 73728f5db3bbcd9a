"""Data-parallel gradient averaging for the dense part of the path (FeatureCross / Dense weights; SURVEY.md
section 8e step 5).  The embedding tables need none of this: every row has one owner (keras_rs_amd/sharded.py).

`GradAllReduce(params)` hooks every parameter: the moment autograd has accumulated a parameter's gradient,
an asynchronous all-reduce of it is launched (RCCL runs it on its own stream), so the reductions of the last
layers overlap the rest of the backward pass -- the remaining cross layers and the embedding update, which is
the long tail of the step.  `wait()` before the optimizer step completes them.  The reference leaves this to
the Keras distribution API (`keras.distribution.DataParallel`, examples/ml_perf/main.py:107-133)."""

from __future__ import annotations

from typing import Iterable

import torch
import torch.distributed as dist


class GradAllReduce:
    def __init__(self, params: Iterable[torch.nn.Parameter], group=None, average: bool = True,
                 run_at_world1: bool = False):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.average = average
        # run_at_world1: issue the collectives on a one-rank communicator too (bench --rccl-self: the RCCL call path
        # on a one-GPU box); otherwise a single rank has nothing to reduce
        self._active = self.world > 1 or (run_at_world1 and dist.is_initialized())
        # RCCL averages inside the collective; gloo has no AVG, the division follows in wait()
        self._avg_op = self._active and average and dist.get_backend(group) == "nccl"
        self._pending: list = []
        self._hooks = []
        if self._active:
            for p in params:
                if p.requires_grad:
                    self._hooks.append(p.register_post_accumulate_grad_hook(self.launch))
                    p._krs_hooks_rejoin_wgrad_stream = True     # launch() waits for autograd's weight-gradient stream

    def launch(self, p: torch.nn.Parameter) -> None:
        """Starts the reduction of `p.grad` (what the hook does; callable directly for gradients that
        were produced before the hooks existed)."""
        if not self._active or p.grad is None:
            return
        if p.grad.is_cuda:
            # a cross layer's weight gradients come off a second stream (autograd.WGRAD_SIDE_STREAM)
            from keras_rs_amd.autograd import wgrad_stream_sync

            wgrad_stream_sync()
        op = dist.ReduceOp.AVG if self._avg_op else dist.ReduceOp.SUM
        self._pending.append((dist.all_reduce(p.grad, op=op, group=self.group, async_op=True), p))

    def wait(self) -> None:
        """Completes the reductions launched during this backward pass; gradients are averages afterwards."""
        for work, p in self._pending:
            work.wait()
            if self.average and not self._avg_op:
                p.grad.div_(self.world)
        self._pending.clear()

    def remove(self) -> None:
        for h in self._hooks:
            h.remove()
        self._hooks.clear()
