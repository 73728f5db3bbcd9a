"""Dense-weight optimizer of the DLRM / DCN step on the C ABI (krs_dense_adagrad): every parameter of the list is
updated by ONE launch instead of four multi-tensor passes.  Same arithmetic and state layout as
torch.optim.Adagrad(lr_decay=0, weight_decay=0): state["sum"] starts at initial_accumulator_value."""

from __future__ import annotations

import ctypes as C

import torch

from keras_rs_amd import _lib as L


class Adagrad(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 0.01, initial_accumulator_value: float = 0.0, eps: float = 1e-10,
                 prepare_casts: bool = False):
        """prepare_casts: right behind the update, prepare the compute-dtype copies (plain + K-contiguous) that the
        FeatureCross / Dense layers want of their kernels for the NEXT forward, for all weights in ONE launch instead
        of one launch per weight inside the layers.  A copy is used once, by the first forward after this step, and
        only if the weight's storage and torch version are what they were here -- so the loop must not modify the
        weights behind torch's back (`p.data...` writes) between `step()` and that forward; off by default."""
        if lr < 0 or eps < 0 or initial_accumulator_value < 0:
            raise ValueError("Adagrad: lr, eps and initial_accumulator_value must be non-negative")
        super().__init__(params, dict(lr=lr, initial_accumulator_value=initial_accumulator_value, eps=eps))
        self.prepare_casts = bool(prepare_casts)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            ps, gs, accs = [], [], []
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.dtype != torch.float32 or p.grad.dtype != torch.float32 or p.grad.is_sparse:
                    raise L.KrsError("keras_rs_amd.optim.Adagrad: dense fp32 parameters and gradients only")
                L.require_device(p, "Adagrad parameter")
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["sum"] = torch.full_like(p, group["initial_accumulator_value"], memory_format=torch.contiguous_format)
                st["step"] += 1
                if not p.is_contiguous():
                    raise L.KrsError("keras_rs_amd.optim.Adagrad: parameters must be contiguous")
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                ps.append(p)
                gs.append(g)
                accs.append(st["sum"])
            n = len(ps)
            if not n:
                continue
            arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])  # noqa: E731
            sizes = (C.c_int64 * n)(*[t.numel() for t in ps])
            rc = L.lib().krs_dense_adagrad(arr(ps), arr(gs), arr(accs), sizes, C.c_int(n), C.c_float(group["lr"]),
                                           C.c_float(group["eps"]), L.stream_ptr())
            L.check(rc, "krs_dense_adagrad")
            # weights created with a constraint (Layer.add_weight(constraint=...)) are projected behind their update, as
            # a Keras optimizer does; the in-place copy moves the version counter, so their cached casts are rebuilt
            from keras_rs_amd.layers.base import apply_constraint

            for p in ps:
                apply_constraint(p)
            # the bf16 copies (plain + K-contiguous) the GEMMs of the next step want, for every weight a layer has asked
            # for, in one launch: the weights have just changed, and the C-ABI update is invisible to torch's versions
            if self.prepare_casts:
                from keras_rs_amd import dense_ops as D

                D.refresh_casts(ps)
        return loss
