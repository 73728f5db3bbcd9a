"""Host side of K1/K2: descriptor management and the fused embedding-bag calls.

`FusedBags` owns the device-resident krs_table / krs_feature descriptor arrays
for one group of same-width tables and launches

    krs_embed_bag_fwd            forward  (one launch for all features)
    krs_embed_bag_bwd_plan       sort of the lookups by row
    krs_embed_bag_bwd_{dense,fused_sgd,fused_adagrad}

through the C ABI.  It is the counterpart of the per-feature python loop at
keras_rs/src/layers/embedding/base_distributed_embedding.py:910-928.
"""

from __future__ import annotations

import ctypes as C
import math
from typing import Sequence

import numpy as np
import torch

from keras_rs_amd import _lib as L
from keras_rs_amd import probe


class FusedBags:
    """Descriptors for `tables` (list of [V, D] tensors of one dtype and width) and
    `features` = [(table_index, combiner, out_col)], bags numbered feature-major."""

    def __init__(self, tables: Sequence[torch.Tensor], features: Sequence[tuple],
                 slots: Sequence[torch.Tensor | None] | None = None,
                 lrs: Sequence[float] | None = None):
        assert len(tables) > 0
        self.tables = list(tables)
        self.slots = list(slots) if slots is not None else [None] * len(tables)
        self.lrs = list(lrs) if lrs is not None else [0.0] * len(tables)
        self.features = [(int(t), L.COMBINERS[c] if isinstance(c, str) else int(c), int(col))
                         for t, c, col in features]
        self.dim = int(tables[0].shape[1])
        self.dtype = tables[0].dtype
        for t in tables:
            if t.dim() != 2 or t.shape[1] != self.dim or t.dtype != self.dtype:
                raise L.KrsError("FusedBags: tables must share embedding_dim and dtype")
        self.row_bases = np.concatenate([[0], np.cumsum([t.shape[0] for t in tables])]).astype(np.int64)
        self.total_rows = int(self.row_bases[-1])
        self._tab_key = None
        self._tab_dev = None
        self._feat_cache: dict = {}
        self._feat_host: dict = {}
        self._tab_host = None

    # ---- descriptors ------------------------------------------------------
    def table_desc(self, weights=None, slots=None) -> torch.Tensor:
        """krs_table array on the device (rebuilt when a storage pointer moved)."""
        weights = self.tables if weights is None else weights
        slots = self.slots if slots is None else slots
        # everything a descriptor carries is in the key: a caller may re-point `self.tables` at a tensor that the
        # caching allocator placed at the previous step's address with a different row count (the sharded layer's
        # transient tables do: their height is the data-dependent number of segments)
        key = tuple(w.data_ptr() for w in weights) + tuple(0 if s is None else s.data_ptr() for s in slots) \
            + tuple(self.lrs) + tuple(int(w.shape[0]) for w in weights) + tuple(int(b) for b in self.row_bases)
        cacheable = weights is self.tables
        if cacheable and key == self._tab_key:
            return self._tab_dev
        arr = np.zeros(len(weights), dtype=L.TABLE_DT)
        for i, w in enumerate(weights):
            L.require_device(w, "embedding table")
            if not w.is_contiguous():
                raise L.KrsError("embedding tables must be contiguous")
            arr[i] = (w.data_ptr(), 0 if slots[i] is None else slots[i].data_ptr(),
                      self.row_bases[i], w.shape[0], self.lrs[i])
        dev = L.struct_to_device(arr, weights[0].device)
        if cacheable:
            self._tab_key, self._tab_dev, self._tab_host = key, dev, arr
        return dev

    def store_lrs(self, lrs: Sequence[float]) -> None:
        """New learning rates for the tables (a schedule's value for the coming update).  Once the descriptor array is on the
        device they are written INTO it by a kernel whose arguments carry the values (krs_store_f32: no upload, no wait, ordered
        on the current stream) -- the form that also works between the replays of a captured step; before that they only
        replace `self.lrs` and ride along with the first upload."""
        lrs = [float(v) for v in lrs]
        if len(lrs) != len(self.tables):
            raise L.KrsError("store_lrs: one learning rate per table")
        if lrs == self.lrs:
            return
        self.lrs = lrs
        if self._tab_dev is None or self._tab_key is None:
            return
        n = len(self.tables)
        vals = (C.c_float * n)(*lrs)
        rc = L.lib().krs_store_f32(C.c_void_p(self._tab_dev.data_ptr() + L.TABLE_DT.fields["lr"][1]), C.c_int64(L.TABLE_DT.itemsize),
                                   vals, C.c_int(n), L.stream_ptr())
        L.check(rc, "krs_store_f32")
        self._tab_key = self._tab_key[:2 * n] + tuple(lrs) + self._tab_key[3 * n:]
        if self._tab_host is not None:
            self._tab_host["lr"] = lrs

    def feature_desc(self, batch: int, hots: Sequence[int] | None, device) -> torch.Tensor:
        # CSR form (hots is None): the descriptors carry no batch-dependent field
        key = (batch if hots is not None else None, None if hots is None else tuple(hots), str(device))
        dev = self._feat_cache.get(key)
        if dev is None:
            arr = np.zeros(len(self.features), dtype=L.FEATURE_DT)
            base = 0
            for i, (t, comb, col) in enumerate(self.features):
                hot = 0 if hots is None else int(hots[i])
                arr[i] = (base, t, hot, comb, col)
                base += batch * hot
            dev = L.struct_to_device(arr, device)
            self._feat_cache[key] = dev
            self._feat_host[key] = arr
        return dev

    # ---- K1 ---------------------------------------------------------------
    def forward(self, ids: torch.Tensor, batch: int, hots: Sequence[int] | None = None,
                offsets: torch.Tensor | None = None, weights: torch.Tensor | None = None,
                out: torch.Tensor | None = None, out_dtype: torch.dtype | None = None,
                want_scale: bool = False, err_flag: torch.Tensor | None = None):
        """ids: flat [nnz] int32/int64 (feature-major).  Dense mode: hots[f] ids per bag of
        feature f; CSR mode: offsets [n_feats*batch+1].  Returns (out [batch, ld], bag_scale|None)."""
        L.require_device(ids, "ids")
        n_feats = len(self.features)
        dev = ids.device
        if (hots is None) == (offsets is None):
            raise L.KrsError("FusedBags.forward: give exactly one of hots (dense) / offsets (CSR)")
        if out is None:
            ld = max(col for _, _, col in self.features) + self.dim
            out = torch.empty((batch, ld), dtype=out_dtype or self.dtype, device=dev)
        if out.stride(-1) != 1:
            raise L.KrsError("out must be row-major")
        scale = torch.empty(n_feats * batch, dtype=torch.float32, device=dev) if want_scale else None
        if weights is not None and weights.dtype != torch.float32:
            weights = weights.float()
        tdesc, fdesc = self.table_desc(), self.feature_desc(batch, hots, dev)
        with probe.span("k1"):
            rc = L.lib().krs_embed_bag_fwd(
                L.ptr(tdesc), L.ptr(fdesc), C.c_int(n_feats),
                L.ptr(ids), C.c_int(L.itype(ids)),
                L.ptr(offsets), C.c_int(L.itype(offsets) if offsets is not None else L.I32),
                L.ptr(weights), C.c_int64(ids.numel()), C.c_int(batch), C.c_int(self.dim),
                C.c_int(L.fdtype(self.tables[0])),
                L.ptr(out), C.c_int(L.fdtype(out)), C.c_int64(out.stride(0)),
                L.ptr(scale), L.ptr(err_flag), L.stream_ptr())
        L.check(rc, "krs_embed_bag_fwd")
        return out, scale

    # ---- K2 ---------------------------------------------------------------
    def plan_backward(self, ids: torch.Tensor, batch: int, hots: Sequence[int] | None = None,
                      offsets: torch.Tensor | None = None, err_flag: torch.Tensor | None = None,
                      global_order: bool = True, ws: torch.Tensor | None = None):
        """Sorts the lookups by global row (krs_embed_bag_bwd_plan / _plan_tables).  Returns the opaque
        workspace tensor the apply calls consume; depends only on the ids, not on gradients.
        global_order (default): the global sort -- out-of-range ids form ONE trailing run, every apply form accepts
        the plan.  global_order=False: dense bags take the faster table-segmented sort, which leaves them at the end
        of their table's run; backward_dense / backward_fused skip them wherever they are (the autograd functions
        ask for this form); for such a plan krs_embed_bag_bwd_sparse reports n_unique = -1 and backward_sparse raises KrsError."""
        L.require_device(ids, "ids")
        nnz = ids.numel()
        nbytes = L.lib().krs_embed_bag_bwd_workspace_bytes(C.c_int64(nnz))
        # ws: a workspace the caller keeps across steps (autograd.EmbedBagFusedFn: the plan runs on a side stream, and a
        # fresh tensor per step there is a fresh hipMalloc per step for as long as the host runs ahead of the device)
        if ws is None or ws.numel() < max(int(nbytes), 1) or ws.device != ids.device:
            ws = torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=ids.device)
        tdesc, fdesc = self.table_desc(), self.feature_desc(batch, hots, ids.device)
        if hots is not None and not global_order:
            th = self._tab_host
            fh = self._feat_host[(batch, tuple(hots), str(ids.device))]
            with probe.span("k2_plan"):
                rc = L.lib().krs_embed_bag_bwd_plan_tables(
                    L.ptr(tdesc), th.ctypes.data_as(C.c_void_p), C.c_int(len(th)),
                    L.ptr(fdesc), fh.ctypes.data_as(C.c_void_p), C.c_int(len(self.features)),
                    L.ptr(ids), C.c_int(L.itype(ids)), C.c_int(batch), C.c_int64(nnz), C.c_int64(self.total_rows),
                    L.ptr(ws), C.c_size_t(ws.numel()), L.ptr(err_flag), L.stream_ptr())
            L.check(rc, "krs_embed_bag_bwd_plan_tables")
            return ws
        with probe.span("k2_plan"):
            rc = L.lib().krs_embed_bag_bwd_plan(
                L.ptr(tdesc), L.ptr(fdesc),
                C.c_int(len(self.features)), L.ptr(ids), C.c_int(L.itype(ids)),
                L.ptr(offsets), C.c_int(L.itype(offsets) if offsets is not None else L.I32),
                C.c_int(batch), C.c_int64(nnz), C.c_int64(self.total_rows),
                L.ptr(ws), C.c_size_t(ws.numel()), L.ptr(err_flag), L.stream_ptr())
        L.check(rc, "krs_embed_bag_bwd_plan")
        return ws

    def _apply_common(self, grad, batch, hots):
        L.require_device(grad, "grad")
        if grad.stride(-1) != 1:
            grad = grad.contiguous()
        return grad, self.feature_desc(batch, hots, grad.device)

    def backward_dense(self, ws, grad, batch, nnz, hots=None, weights=None, bag_scale=None,
                       out: Sequence[torch.Tensor] | None = None):
        """Dense per-table gradients [V, D] fp32 (the reference autodiff result)."""
        grad, fdesc = self._apply_common(grad, batch, hots)
        if out is None:
            out = [torch.zeros(t.shape, dtype=torch.float32, device=grad.device) for t in self.tables]
        gdesc = self.table_desc(weights=list(out), slots=[None] * len(out))
        rc = L.lib().krs_embed_bag_bwd_dense(
            L.ptr(gdesc), C.c_int(len(out)), L.ptr(fdesc), C.c_int(len(self.features)),
            L.ptr(weights), L.ptr(bag_scale), L.ptr(grad), C.c_int(L.fdtype(grad)),
            C.c_int64(grad.stride(0)), C.c_int(batch), C.c_int(self.dim), C.c_int64(nnz),
            L.ptr(ws), L.stream_ptr())
        L.check(rc, "krs_embed_bag_bwd_dense")
        return list(out)

    def backward_fused(self, kind, ws, grad, batch, nnz, hots=None, weights=None, bag_scale=None, hyper=None):
        """In-place SGD / Adagrad / Adam / FTRL on the touched rows of the tables (and their slots:
        [V, D] fp32 for Adagrad, [2, V, D] fp32 for Adam (m, v) and FTRL (accumulator, linear)).
        hyper: Adam (beta_1, beta_2, epsilon, bias_correction); FTRL (learning_rate_power, l1, l2, beta)."""
        grad, fdesc = self._apply_common(grad, batch, hots)
        if kind != "sgd" and any(s is None for s in self.slots):
            raise L.KrsError(f"fused {kind} needs a slot buffer per table")
        head = (L.ptr(self.table_desc()), C.c_int(len(self.tables)), L.ptr(fdesc),
                C.c_int(len(self.features)), L.ptr(weights), L.ptr(bag_scale), L.ptr(grad),
                C.c_int(L.fdtype(grad)), C.c_int64(grad.stride(0)), C.c_int(batch), C.c_int(self.dim),
                C.c_int(L.fdtype(self.tables[0])), C.c_int64(nnz))
        tail = (L.ptr(ws), L.stream_ptr())
        if kind in ("sgd", "adagrad", "adagrad_rowwise"):
            fn = {"sgd": L.lib().krs_embed_bag_bwd_fused_sgd, "adagrad": L.lib().krs_embed_bag_bwd_fused_adagrad,
                  "adagrad_rowwise": L.lib().krs_embed_bag_bwd_fused_adagrad_rowwise}[kind]
            with probe.span("k2_apply"):
                rc = fn(*head, *tail)
        elif kind in ("adam", "ftrl"):
            if hyper is None or len(hyper) != 4:
                raise L.KrsError(f"fused {kind} needs its four hyper-parameters")
            if kind == "adam" and isinstance(hyper[3], torch.Tensor):
                # bias correction kept in device memory (StepConstants): the launch reads it when it RUNS
                L.require_device(hyper[3], "Adam bias correction")
                with probe.span("k2_apply"):
                    rc = L.lib().krs_embed_bag_bwd_fused_adam_dyn(*head, *(C.c_float(float(h)) for h in hyper[:3]),
                                                                  L.ptr(hyper[3]), *tail)
                L.check(rc, "krs_embed_bag_bwd_fused_adam_dyn")
                return
            fn = {"adam": L.lib().krs_embed_bag_bwd_fused_adam, "ftrl": L.lib().krs_embed_bag_bwd_fused_ftrl}[kind]
            with probe.span("k2_apply"):
                rc = fn(*head, *(C.c_float(float(h)) for h in hyper), *tail)
        else:
            raise L.KrsError(f"unknown fused optimizer {kind!r}")
        L.check(rc, f"krs_embed_bag_bwd_fused_{kind}")

    def backward_sparse(self, ws, grad, batch, nnz, hots=None, weights=None, bag_scale=None):
        """(unique global rows [U] int64, summed gradients [U, D] fp32)."""
        grad, fdesc = self._apply_common(grad, batch, hots)
        dev = grad.device
        rows = torch.empty(max(nnz, 1), dtype=torch.int64, device=dev)
        vals = torch.empty((max(nnz, 1), self.dim), dtype=torch.float32, device=dev)
        n_u = torch.zeros(1, dtype=torch.int64, device=dev)
        rc = L.lib().krs_embed_bag_bwd_sparse(
            L.ptr(fdesc), C.c_int(len(self.features)), L.ptr(weights), L.ptr(bag_scale), L.ptr(grad),
            C.c_int(L.fdtype(grad)), C.c_int64(grad.stride(0)), C.c_int(batch), C.c_int(self.dim),
            C.c_int64(nnz), L.ptr(ws), L.ptr(rows), L.ptr(vals), L.ptr(n_u), L.stream_ptr())
        L.check(rc, "krs_embed_bag_bwd_sparse")
        u = int(n_u.item())
        if u < 0:
            raise L.KrsError("krs_embed_bag_bwd_sparse: the workspace holds a table-segmented plan; "
                             "plan_backward(global_order=True) is what the compact form needs")
        return rows[:u], vals[:u]


class StepConstants:
    """The optimizer constants of a fused group that change from step to step -- scheduled learning rates and Adam's
    bias-correction factor sqrt(1 - beta_2^t) / (1 - beta_1^t) -- evaluated on the HOST once per update (the reference does
    the same: callable learning rates, jax/config_conversion.py:136-176) and kept in DEVICE memory, where the fused update
    reads them when it runs: the learning rates inside the krs_table descriptors, the Adam factor in a float of its own.
    `advance()` is that once-per-update host work.  An eager step calls it from its backward pass; a step captured into a HIP
    graph does not (the capture would freeze one step's values into the graph) -- keras_rs_amd.graphs.GraphedStep calls it
    before every replay instead, on the replay's stream, so replay k runs with the constants of update k."""

    def __init__(self, owner, bags_of, lrs_at, adam_betas=None):
        self.owner = owner            # object with the `step` count (fused updates applied so far; checkpointed by the layers)
        self.bags_of = bags_of        # () -> FusedBags whose descriptors hold the learning rates
        self.lrs_at = lrs_at          # None (constant rates) or step -> [lr per table]
        self.adam_betas = adam_betas  # None or (beta_1, beta_2)
        self.bias_correction = None   # device float32[1]

    def advance(self) -> None:
        o = self.owner
        if self.lrs_at is not None:
            self.bags_of().store_lrs(self.lrs_at(o.step))
        o.step += 1
        if self.adam_betas is not None:
            if self.bias_correction is None:
                if torch.cuda.is_current_stream_capturing():
                    raise L.KrsError("fused Adam: run one eager step before capturing (the bias-correction float is "
                                     "allocated on first use)")
                self.bias_correction = torch.ones(1, dtype=torch.float32, device=self.bags_of().tables[0].device)
            b1, b2 = self.adam_betas
            bc = math.sqrt(1.0 - b2 ** o.step) / (1.0 - b1 ** o.step)
            rc = L.lib().krs_store_f32(L.ptr(self.bias_correction), C.c_int64(4), (C.c_float * 1)(bc), C.c_int(1), L.stream_ptr())
            L.check(rc, "krs_store_f32")

    def on_backward(self) -> None:
        """Called where a step's backward pass reaches the fused update."""
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            from keras_rs_amd import graphs

            if self.adam_betas is not None and self.bias_correction is None:
                raise L.KrsError("fused Adam: run one eager step before capturing")
            graphs.before_each_replay(self.advance)      # raises outside GraphedStep's capture
        else:
            self.advance()
