"""Row-sharded embedding tables across the GPUs of one node (config C4 / C5).

Layout (the reference's own convention, sharding_strategy="MOD" at
keras_rs/src/layers/embedding/jax/embedding_utils.py:194, reassembly code at
tensorflow/distributed_embedding.py:316-328): global row r of every sharded table lives on rank
r % N at local row r // N.  Tables of one embedding width and one fused optimizer form a GROUP; each
rank keeps ONE stacked buffer [sum_t ceil(V_t / N), D] per group, table t starting at row
off_t = sum_{s<t} ceil(V_s / N), so the composite id c = off_t * N + r gives the owner (c % N) and the
stacked local row (c // N) of a lookup in one integer.

Small tables are not worth an exchange: with `replicate_below=V0` every table with fewer than V0 rows is
REPLICATED -- each rank holds it whole as an ordinary trainable weight, looks it up locally (one fused
launch, dense [V, D] gradients from K2) and its gradient joins the data-parallel all-reduce of the dense
weights (keras_rs_amd/dp.py), exactly what the reference's model does with the tables under
`embedding_threshold` (examples/ml_perf/main.py:135-141, model.py:128-148: plain keras Embeddings trained by
the model optimizer).  Without it a 3-row Criteo table would send all of its 65,536 x L lookups to <= 3 ranks.

Per step and group (one process per GPU, torch.distributed over RCCL/xGMI; xGMI is point-to-point, so the
exchange is an all-to-all whose pairs each use their own link).  What crosses the links is one PARTIALLY
POOLED vector per (bag, owner) pair that has lookups -- not one vector per lookup: with the ml_perf bag
lengths (214 lookups per sample) that is 41 vectors per sample at N = 2 and 79 at N = 8.
  fwd  1. krs_shard_route (ONE call, csrc/shard_route.hip): range check, composite id -> owner / local row,
          stable grouping by owner, SEGMENTS (runs of one bag inside an owner's bucket), per-lookup weight =
          user weight x combiner scale, the packed send buffer [per owner: rows | weights | segment lengths],
          per-owner counts
       2. the counts travel in one tiny all-to-all; both sides' counts reach the host through page-locked
          memory behind a sequence flag (krs_publish_i64): the ONE host wait of the lookup
       3. ONE all-to-all-v of the packed buffer; owner: krs_shard_unpack -> rows, weights, CSR segment
          offsets; K1 in CSR form -> one partial vector per segment
       4. ONE all-to-all-v of the partials back; home: krs_shard_combine sums the <= N partials of every bag
          straight into the output slab
  bwd  d(partial of a segment) = the output gradient of its bag (a row gather by the segment's gradient row),
       ONE all-to-all-v to the owners, K2 fused optimizer on the shard in CSR form.  Every row has exactly
       one owner, so table gradients need no cross-GPU reduction.
Gradient convention: the backward applies to every row the SUM over all ranks of d(loss_rank)/d(row).  With a
loss that is a mean over the LOCAL batch and dense weights averaged over ranks (dp.GradAllReduce), pass
`grad_average=True`: the contributions are scaled by 1 / world, so tables and dense weights both see the
gradient of the global-batch mean (the reference's SparseCore update is a global-batch update).

Partials travel in `partial_dtype` (default: the compute dtype; bf16 at C3, one extra rounding of each
partial compared with the single-GPU path -- the sums differ by <= 1 bf16 ulp per partial; "float32" keeps
the single-GPU summation exact at twice the bytes).

The exchange logic is independent of the compute kernels: `kernels` is an object with the methods of
HipShardKernels.  The product default runs the HIP kernels; the CPU/gloo tests in
tests/test_sharded_gloo.py inject an oracle-backed implementation to check the plumbing with world_size 2 / 3.
"""

from __future__ import annotations

import ctypes as C
import dataclasses
import math
import os
import time
import weakref
from typing import Any

import numpy as np
import torch
import torch.distributed as dist

from keras_rs_amd import _lib as L
from keras_rs_amd import probe
from keras_rs_amd.layers import base
from keras_rs_amd.layers.distributed_embedding import (DistributedEmbedding, FusedOptimizer,
                                                       resolve_fused_optimizer)
from keras_rs_amd.layers.distributed_embedding_config import FeatureConfig, TableConfig


class HipShardKernels:
    """The compute side of the sharded path on MI355X (K1 / K2 / K6 through the C ABI)."""

    device_step_constants = True     # scheduled learning rates / Adam's bias correction are read from device memory

    def __init__(self):
        self._shard_bags: dict = {}
        self._desc_cache: dict = {}

    def _bags_for(self, table, slot, lr):
        """One FusedBags (device descriptors) per shard storage, re-used across steps."""
        from keras_rs_amd.embedding_ops import FusedBags

        key = (table.data_ptr(), 0 if slot is None else slot.data_ptr())
        fb = self._shard_bags.get(key)
        if fb is None:
            fb = self._shard_bags[key] = FusedBags([table], [(0, "sum", 0)], slots=[slot], lrs=[lr])
        return fb

    def _transient_bags(self, table, feats):
        """FusedBags for a per-step `table` (an output gradient): one object per feature list is kept and
        re-pointed, so that only the 32-byte table descriptor is uploaded per call."""
        from keras_rs_amd.embedding_ops import FusedBags

        key = ("transient", tuple(feats), table.shape[1], table.dtype)
        fb = self._shard_bags.get(key)
        if fb is None:
            fb = self._shard_bags[key] = FusedBags([table], list(feats))
        fb.tables = [table]
        fb.row_bases[:] = (0, table.shape[0])
        fb.total_rows = int(table.shape[0])
        return fb

    # ---- K6 -------------------------------------------------------------------------------------
    def route(self, desc: np.ndarray, ids, offsets, weights, batch: int, n_shards: int, emit_w: bool, err_flag=None):
        """krs_shard_route.  desc: SHARD_FEATURE_DT array.  Returns dict(packed, seg_grow, bag_seg, counts):
        device tensors, counts = int64 [3, n_shards] (lookups, segments, packed words per owner)."""
        dev = ids.device
        key = (desc.tobytes(), str(dev))
        ddev = self._desc_cache.get(key)
        if ddev is None:
            ddev = self._desc_cache[key] = L.struct_to_device(desc, dev)
        nnz, n_feats = ids.numel(), len(desc)
        n_bags = batch * n_feats
        packed = torch.empty(max(nnz * (2 + int(emit_w)), 1), dtype=torch.int32, device=dev)
        seg_bag = torch.empty(max(nnz, 1), dtype=torch.int32, device=dev)
        seg_grow = torch.empty(max(nnz, 1), dtype=torch.int32, device=dev)
        bag_seg = torch.empty((max(n_bags, 1), n_shards), dtype=torch.int32, device=dev)
        counts = torch.empty((3, n_shards), dtype=torch.int64, device=dev)
        wsb = int(L.lib().krs_shard_route_workspace_bytes(C.c_int64(nnz), C.c_int64(n_bags), C.c_int(n_shards)))
        ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=dev)
        if weights is not None and weights.dtype != torch.float32:
            weights = weights.float()
        rc = L.lib().krs_shard_route(
            L.ptr(ddev), desc.ctypes.data_as(C.c_void_p), C.c_int(n_feats), L.ptr(ids), C.c_int(L.itype(ids)),
            L.ptr(offsets), C.c_int(L.itype(offsets) if offsets is not None else L.I32), L.ptr(weights),
            C.c_int64(nnz), C.c_int(batch), C.c_int(n_shards), C.c_int(int(emit_w)), L.ptr(packed), L.ptr(seg_bag),
            L.ptr(seg_grow), L.ptr(bag_seg), L.ptr(counts), L.ptr(err_flag), L.ptr(ws), C.c_size_t(ws.numel()),
            L.stream_ptr())
        L.check(rc, "krs_shard_route")
        return dict(packed=packed, seg_grow=seg_grow, bag_seg=bag_seg, counts=counts)

    def unpack(self, packed, lookups, segments, weighted: bool):
        """krs_shard_unpack: (rows int32 [sum lookups], w fp32 | None, offsets int32 [sum segments + 1])."""
        dev = packed.device
        n = len(lookups)
        n_cnt, n_seg = int(sum(lookups)), int(sum(segments))
        rows = torch.empty(max(n_cnt, 1), dtype=torch.int32, device=dev)
        w = torch.empty(max(n_cnt, 1), dtype=torch.float32, device=dev) if weighted else None
        off = torch.empty(n_seg + 1, dtype=torch.int32, device=dev)
        wsb = int(L.lib().krs_shard_unpack_workspace_bytes(C.c_int64(n_seg)))
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        arr = lambda v: (C.c_int64 * n)(*[int(x) for x in v])  # noqa: E731
        rc = L.lib().krs_shard_unpack(L.ptr(packed), C.c_int(n), arr(lookups), arr(segments), C.c_int(int(weighted)),
                                      L.ptr(rows), L.ptr(w), L.ptr(off), L.ptr(ws), C.c_size_t(wsb), L.stream_ptr())
        L.check(rc, "krs_shard_unpack")
        return rows[:n_cnt], (None if w is None else w[:n_cnt]), off

    def route_static(self, desc: np.ndarray, ids, offsets, weights, batch: int, n_shards: int, emit_w: bool,
                     cap_l: int, cap_s: int, err_flag=None):
        """krs_shard_route_static: fixed-size blocks.  Returns dict(packed [n_shards, W] int32, seg_grow
        [n_shards*cap_s], bag_seg (segment SLOTS), counts [3, n_shards] on the device)."""
        dev = ids.device
        key = (desc.tobytes(), str(dev))
        ddev = self._desc_cache.get(key)
        if ddev is None:
            ddev = self._desc_cache[key] = L.struct_to_device(desc, dev)
        nnz, n_feats = ids.numel(), len(desc)
        n_bags = batch * n_feats
        words = int(L.lib().krs_shard_static_block_words(C.c_int64(cap_l), C.c_int64(cap_s), C.c_int(int(emit_w))))
        packed = torch.empty((n_shards, words), dtype=torch.int32, device=dev)
        seg_bag = torch.empty(max(nnz, 1), dtype=torch.int32, device=dev)
        seg_grow = torch.empty(n_shards * cap_s, dtype=torch.int32, device=dev)
        bag_seg = torch.empty((max(n_bags, 1), n_shards), dtype=torch.int32, device=dev)
        counts = torch.empty((3, n_shards), dtype=torch.int64, device=dev)
        wsb = int(L.lib().krs_shard_route_workspace_bytes(C.c_int64(nnz), C.c_int64(n_bags), C.c_int(n_shards)))
        ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=dev)
        if weights is not None and weights.dtype != torch.float32:
            weights = weights.float()
        rc = L.lib().krs_shard_route_static(
            L.ptr(ddev), desc.ctypes.data_as(C.c_void_p), C.c_int(n_feats), L.ptr(ids), C.c_int(L.itype(ids)),
            L.ptr(offsets), C.c_int(L.itype(offsets) if offsets is not None else L.I32), L.ptr(weights),
            C.c_int64(nnz), C.c_int(batch), C.c_int(n_shards), C.c_int(int(emit_w)), C.c_int64(cap_l), C.c_int64(cap_s),
            L.ptr(packed), L.ptr(seg_bag), L.ptr(seg_grow), L.ptr(bag_seg), L.ptr(counts), L.ptr(err_flag), L.ptr(ws),
            C.c_size_t(ws.numel()), L.stream_ptr())
        L.check(rc, "krs_shard_route_static")
        return dict(packed=packed, seg_grow=seg_grow, bag_seg=bag_seg, counts=counts)

    def unpack_static(self, packed, cap_l: int, cap_s: int, weighted: bool):
        """krs_shard_unpack_static: (rows int32 [n*cap_l], w | None, offsets int32 [n*cap_s + 1], stats int64 [4])."""
        dev = packed.device
        n = packed.shape[0]
        rows = torch.empty(n * cap_l, dtype=torch.int32, device=dev)
        w = torch.empty(n * cap_l, dtype=torch.float32, device=dev) if weighted else None
        off = torch.empty(n * cap_s + 1, dtype=torch.int32, device=dev)
        stats = torch.empty(4, dtype=torch.int64, device=dev)
        wsb = int(L.lib().krs_shard_unpack_workspace_bytes(C.c_int64(n * cap_s)))
        ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=dev)
        rc = L.lib().krs_shard_unpack_static(L.ptr(packed), C.c_int(n), C.c_int64(cap_l), C.c_int64(cap_s),
                                             C.c_int(int(weighted)), L.ptr(rows), L.ptr(w), L.ptr(off), L.ptr(stats),
                                             L.ptr(ws), C.c_size_t(wsb), L.stream_ptr())
        L.check(rc, "krs_shard_unpack_static")
        return rows, w, off, stats

    def combine(self, partials, bag_seg, batch: int, n_feats: int, dim: int, out):
        """krs_shard_combine into `out` (a [batch, n_feats*dim] row-major window of the slab)."""
        rc = L.lib().krs_shard_combine(L.ptr(partials), L.ptr(bag_seg), C.c_int(batch), C.c_int(n_feats),
                                       C.c_int(bag_seg.shape[1]), C.c_int(dim), C.c_int(L.fdtype(out)), L.ptr(out),
                                       C.c_int64(out.stride(0)), L.stream_ptr())
        L.check(rc, "krs_shard_combine")
        return out

    # ---- K1 / K2 on the shard ---------------------------------------------------------------------
    def gather_rows(self, table: torch.Tensor, rows: torch.Tensor) -> torch.Tensor:
        """table[rows] (K1 one-hot form).  `table` is a per-step tensor here (an output gradient)."""
        n = rows.numel()
        if n == 0:
            return torch.empty((0, table.shape[1]), dtype=table.dtype, device=table.device)
        fb = self._transient_bags(table, [(0, "sum", 0)])
        out, _ = fb.forward(rows, n, hots=(1,))
        fb.tables = []   # do not keep the step's tensor alive
        return out

    def pool_segments(self, table, rows, offsets, weights, out_dtype):
        """Owner side: one vector per CSR segment of `rows`, sum of weight * table[row]."""
        n_seg = offsets.numel() - 1
        if n_seg == 0:
            return torch.empty((0, table.shape[1]), dtype=out_dtype, device=table.device)
        out, _ = self._bags_for(table, None, 0.0).forward(rows, n_seg, offsets=offsets, weights=weights,
                                                          out_dtype=out_dtype)
        return out

    def plan_segments(self, table, slot, rows, offsets, ws=None):
        """Owner side: the backward plan (sort of the received rows + segment list) of a later apply_segments call.  It
        depends on the rows alone, so the layer runs it at FORWARD time on a side stream, under the dense part of the step
        (as autograd.EmbedBagFusedFn does for the single-GPU layer).  Returns the workspace, or None for nothing to do."""
        n_seg = offsets.numel() - 1
        if rows.numel() == 0 or n_seg == 0:
            return None
        return self._bags_for(table, slot, 0.0).plan_backward(rows, n_seg, offsets=offsets, ws=ws)

    def prepare_plan(self, table, slot, rows, offsets):
        """Uploads (and caches) the descriptor blocks plan_segments will read -- called on the MAIN stream before the side
        stream forks, so that a cold cache cannot race the two streams."""
        fb = self._bags_for(table, slot, 0.0)
        fb.table_desc()
        fb.feature_desc(offsets.numel() - 1, None, rows.device)

    def apply_segments(self, table, slot, rows, offsets, weights, seg_grads, lr, kind, hyper=None, grad_scale=1.0, ws=None):
        """Owner side: fused optimizer step; lookup i of segment s carries weights[i] * grad_scale * seg_grads[s].
        ws: the plan of plan_segments for these rows (None: planned here)."""
        n_seg = offsets.numel() - 1
        if rows.numel() == 0 or n_seg == 0:
            return
        fb = self._bags_for(table, slot, 0.0)
        fb.slots = [slot]
        if lr is not None:     # (None: the descriptor already holds this update's rate -- StepConstants wrote it)
            fb.store_lrs([float(lr)])
        if ws is None:
            ws = fb.plan_backward(rows, n_seg, offsets=offsets)
        scale = None
        if grad_scale != 1.0:
            scale = torch.full((n_seg,), float(grad_scale), dtype=torch.float32, device=rows.device)
        fb.backward_fused(kind, ws, seg_grads, n_seg, rows.numel(), weights=weights, bag_scale=scale, hyper=hyper)


class _HostCounts:
    """Device counters -> host through page-locked memory and a sequence flag (krs_publish_i64): the host
    polls the flag instead of synchronising the stream."""

    def __init__(self):
        self.buf = torch.zeros(256, dtype=torch.int64).pin_memory()
        self.seq = 0

    def read(self, dev_counts: torch.Tensor) -> list:
        n = dev_counts.numel()
        self.seq += 1
        rc = L.lib().krs_publish_i64(L.ptr(dev_counts), C.c_int(n), C.c_void_p(self.buf.data_ptr()),
                                     C.c_int64(self.seq), L.stream_ptr())
        L.check(rc, "krs_publish_i64")
        flag = self.buf[n:n + 1]
        t0 = time.monotonic()
        while int(flag.item()) != self.seq:
            if time.monotonic() - t0 > 120.0:
                raise L.KrsError("sharded lookup: the size exchange did not arrive within 120 s")
        return self.buf[:n].tolist()


@dataclasses.dataclass
class _ShardGroup:
    """Sharded tables of one embedding width and one fused optimizer: one stacked shard, one exchange."""

    dim: int
    fused: FusedOptimizer
    paths: list
    table_of_feature: list
    table_configs: list
    local_rows: list = dataclasses.field(default_factory=list)
    row_off: list = dataclasses.field(default_factory=list)
    step: int = 0
    pname: str = ""
    sname: str = ""


def _release_plan_of(layer_ref, gi, owns) -> None:
    layer = layer_ref()
    if layer is not None:
        layer._release_plan(gi, owns)


class _ShardedLookupFn(torch.autograd.Function):
    """Outputs: the slab [B, lead + n*dim] (see DistributedEmbedding.slab_lead_cols) and its n feature views."""

    @staticmethod
    def forward(ctx, layer, gi, ids, batch, hots, offsets, weights, lead, anchor):
        from keras_rs_amd.autograd import _split_columns

        ctx.set_materialize_grads(False)  # unused outputs arrive as None, not as zero tensors
        # (grad mode is off inside a Function's forward: whether a backward will follow is what the anchor says)
        layer._backward_follows = bool(ctx.needs_input_grad[8])
        try:
            slab, saved = layer._forward_impl(gi, ids, batch, hots, offsets, weights, lead)
        finally:
            layer._backward_follows = False
        ctx.layer, ctx.gi, ctx.saved, ctx.lead = layer, gi, saved, lead
        if saved.get("plan") is not None:
            # a graph that is dropped without a backward gives the kept plan workspace back (else every later step
            # would allocate a fresh one on the side stream)
            weakref.finalize(ctx, _release_plan_of, weakref.ref(layer), gi, saved["plan"][2])
        g = layer._sgroups[gi]
        return (slab,) + _split_columns(slab, len(g.paths), g.dim, lead)

    @staticmethod
    def backward(ctx, g_slab, *gs):
        layer, g = ctx.layer, ctx.layer._sgroups[ctx.gi]
        from keras_rs_amd.autograd import _sum_slab_and_feature_grads

        out_dtype, device = ctx.saved["out_meta"]
        lead, n = ctx.lead, len(g.paths)
        if (lead > 0 and lead % g.dim == 0 and g_slab is not None and all(x is None for x in gs) and g_slab.is_contiguous()
                and g_slab.dtype == ctx.saved["pdt"] and g_slab.shape[1] == lead + n * g.dim):
            # the gradient of the whole slab [B, lead + n*dim] IS a table of B * (n + lead/dim) rows of `dim`: the
            # segments' gradient rows are gathered straight out of it (no copy of its feature columns)
            layer._backward_impl(ctx.gi, None, ctx.saved, slab_grad=g_slab, lead_slots=lead // g.dim)
        else:
            grad = _sum_slab_and_feature_grads(g_slab, gs, lead, ctx.saved["batch"], n, g.dim, out_dtype, device)
            layer._backward_impl(ctx.gi, grad, ctx.saved)
        return (None,) * 9     # (the anchor gets no gradient: it only makes autograd call this function)


class ShardedDistributedEmbedding(base.Layer):
    """DistributedEmbedding whose tables are MOD row-sharded over the ranks of `process_group`.

    feature_configs: flat dict {name: FeatureConfig}.  Sharded tables run their TableConfig.optimizer (SGD /
    Adagrad / Adam / Ftrl, see resolve_fused_optimizer) inside the backward, as on the single-GPU 'sparsecore'
    placement; tables with fewer than `replicate_below` rows (and every table whose placement is
    'default_device') are replicated trainable weights with dense gradients.  call(inputs) takes raw
    {name: ids} (dense [batch, hot] / [batch] arrays, embed_reduce.Ragged or numpy object arrays of rows) or
    the result of preprocess()."""

    def __init__(self, feature_configs: dict[str, FeatureConfig], *, process_group=None, kernels=None,
                 slab_lead_cols: int = 0, replicate_below: int = 0, grad_average: bool = False,
                 partial_dtype=None, exchange: str = "exact", capacity="auto", capacity_headroom: float = 1.25,
                 update_stats: bool = True, capacity_settle_steps: int = 0, virtual_world: int = 0, **kwargs: Any):
        super().__init__(**kwargs)
        # (base_distributed_embedding.py:461-464: whether the per-partition limits follow the running statistics; here
        #  the capacities of the static exchange.  False = they stay what they were sized to, overflows are only counted)
        self.update_stats = bool(update_stats)
        if exchange not in ("exact", "static"):
            raise ValueError(f"exchange must be 'exact' or 'static', got {exchange!r}")
        # "exact": the all-to-alls carry exactly the lookups of the step; their sizes reach the host through ONE
        #   wait per lookup (krs_publish_i64).
        # "static": the reference's SparseCore contract (static buffers sized by max_ids_per_partition /
        #   max_unique_ids_per_partition, ids beyond them dropped, limits learnt from running statistics:
        #   distributed_embedding_config.py:54-61, jax/embedding_utils.py:187-197, jax/distributed_embedding.py:657-664):
        #   every (home, owner) pair exchanges a fixed-size block, no count ever reaches the host, the step has no
        #   host wait.  capacity = "auto": expected per-owner load of a uniform id distribution x capacity_headroom
        #   (MOD interleaving balances any distribution over ROWS; needs dense bags); "table_config": the sums of
        #   TableConfig.max_ids_per_partition / max_unique_ids_per_partition over the group's features; or an
        #   explicit (lookups, segments) pair per (home, owner) block.  Lookups that do not fit are dropped and
        #   counted (`overflow_steps`), and the capacity grows to the need every rank has seen (same step on all ranks).
        #   What dropping means for the result (as on SparseCore, jax/embedding_utils.py:187-197): the dropped lookups
        #   contribute nothing and are NOT renormalised away -- the mean / sqrtn scale of a bag is computed at home from
        #   ALL its lookups and folded into the per-lookup weights before the route, so a bag that lost lookups keeps the
        #   full-bag denominator (its output is biased towards zero for that step, not a mean over the survivors); the
        #   backward drops the same lookups.  Capacities react two steps late and only while `update_stats` is on.
        #   `overflow_steps` counts the steps in which anything was dropped (0 = the exact result); bench.py puts it at
        #   the top level of its line and marks the line invalid when it is not 0.
        self.exchange = exchange
        self._capacity_spec = capacity
        self.capacity_headroom = float(capacity_headroom)
        # Shrinking the blocks to the settled statistics is OPT-IN (0 = never shrink, the default): a heavier or more
        # skewed batch arriving after a shrink overflows -- its lookups are dropped for the two steps the statistics take
        # to react -- while the padding it saves only matters on the links.  Steps replayed from a graph never shrink
        # (poll_exchange_stats): the captured buffers keep their size, a smaller `_caps` would only turn a later need
        # that still fits the captured blocks into a false overflow.
        self.capacity_settle_steps = int(capacity_settle_steps)
        self.capacity_shrinks = 0
        self._settle: dict = {}          # caps key -> [steps that fit in a row, their largest need_l, need_s]
        self._caps: dict = {}            # (group, batch, hots) -> [cap_lookups, cap_segments]
        self._stats_q: dict = {}           # group -> list of (step, event, pinned stats, caps key)
        self._stat_bufs: dict = {}         # group -> ring of four page-locked buffers the statistics land in
        self._graph_stats: dict = {}       # group -> (page-locked statistics of a CAPTURED step, caps key)
        self._stat_step: dict = {}
        self.overflow_steps = 0
        self.slab_lead_cols = int(slab_lead_cols)  # as DistributedEmbedding: room for layers.concat_features
        self._pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        # virtual_world = N (measurement only, no process group): ONE process holds rank 0's 1/N shard and runs the step with
        # the shapes of an N-way job -- ids routed to N owners, N blocks per exchange, N partials per bag -- while every
        # "all-to-all" is a device copy of its own send buffer (block k comes back as if rank k had sent what this rank sent it:
        # the same load under MOD interleaving; a local row is valid in every shard, they have one shape).  The per-rank kernel
        # times of an N-way job on one GPU (bench.py --virtual-world); table reads / writes across "ranks" mean nothing here.
        self.virtual = int(virtual_world) > 1 and not dist.is_initialized()
        if self.virtual:
            self.world = int(virtual_world)
        self.kernels = kernels or HipShardKernels()
        self.replicate_below = int(replicate_below)
        self.grad_average = bool(grad_average)
        self._partial_dtype = partial_dtype
        self._feature_configs = feature_configs
        self._paths = list(feature_configs.keys())
        self._sgroups: list[_ShardGroup] = []
        self._where: dict[str, tuple] = {}       # path -> ("shard", group, index) | ("rep",)
        rep_cfgs: dict[str, FeatureConfig] = {}
        rep_tables: dict[int, TableConfig] = {}
        for p, fc in feature_configs.items():
            tc = fc.table
            if tc.placement == "default_device" or tc.vocabulary_size < self.replicate_below:
                # a replicated twin of the table: same name / shape / initializer, ordinary trainable weight
                if id(tc) not in rep_tables:
                    rep_tables[id(tc)] = dataclasses.replace(tc, placement="default_device")
                rep_cfgs[p] = dataclasses.replace(fc, table=rep_tables[id(tc)])
                self._where[p] = ("rep",)
                continue
            fo = resolve_fused_optimizer(tc.optimizer)
            if fo is None:
                raise NotImplementedError(
                    f"Table '{tc.name}': a sharded table runs its optimizer inside the backward (SGD / Adagrad / "
                    f"Adam / Ftrl, the option set of the reference's SparseCore path); got {tc.optimizer!r}. "
                    "Smaller tables can be replicated instead (replicate_below / placement='default_device').")
            # one group = one stacked shard = one exchange: same width, same optimizer INCLUDING its learning rate
            # (or schedule object); C3 / C5 have one group
            g = next((g for g in self._sgroups if g.dim == tc.embedding_dim and g.fused == fo), None)
            if g is None:
                g = _ShardGroup(tc.embedding_dim, fo, [], [], [])
                self._sgroups.append(g)
            ti = next((i for i, t in enumerate(g.table_configs) if t is tc), None)
            if ti is None:
                ti = len(g.table_configs)
                g.table_configs.append(tc)
            self._where[p] = ("shard", self._sgroups.index(g), len(g.paths))
            g.paths.append(p)
            g.table_of_feature.append(ti)
        # A table with very few rows has few OWNERS under MOD sharding (3 rows: ranks 0-2 own everything) -- with the static
        # exchange its lookups all land in a few (home, owner) blocks sized for a uniform spread and are DROPPED beyond the
        # capacity (profiles/r5z_bench_c5_gpus8_one_gpu_gloo_tiny_tables_sharded_overflow.json: `overflow_steps` 1 with the
        # Criteo tables of 3 ... 155 rows left sharded over 8 ranks).  The reference's model replicates such tables
        # (examples/ml_perf/main.py:135-141, embedding_threshold); say so loudly when one stays sharded.
        if self.world > 1:
            tiny = sorted({tc.name for g in self._sgroups for tc in g.table_configs if tc.vocabulary_size < 64 * self.world})
            if tiny:
                import warnings

                warnings.warn(
                    f"ShardedDistributedEmbedding: table(s) {tiny} have fewer than 64 x world ({64 * self.world}) rows and stay "
                    f"MOD-sharded over {self.world} ranks: their lookups reach only a few owners, and the static exchange drops "
                    "what exceeds a block's capacity (overflow_steps).  Replicate them: replicate_below=<rows> or "
                    "TableConfig(placement='default_device').", stacklevel=2)
            self.tiny_sharded_tables = tiny
        else:
            self.tiny_sharded_tables = []
        for gi, g in enumerate(self._sgroups):
            g.local_rows = [math.ceil(tc.vocabulary_size / self.world) for tc in g.table_configs]
            g.row_off = [0]
            for n_loc in g.local_rows:
                g.row_off.append(g.row_off[-1] + n_loc)
            if g.row_off[-1] >= 2 ** 31:
                raise NotImplementedError("a rank's stacked shard must stay below 2^31 rows")
            g.pname, g.sname = f"shard{gi}", f"shard{gi}_slot"
            self.register_parameter(g.pname, None)
        self._replicated = None
        if rep_cfgs:
            self._replicated = DistributedEmbedding(rep_cfgs, dtype=self.dtype_policy, device=self._device,
                                                    name=f"{self.name}_replicated")
        self._anchor = None
        self._xstream = None              # exchange stream of prefetch()
        # The owner-side backward plan (sort of the received rows) depends on the ids alone: on the GPU it runs at FORWARD
        # time on a side stream, under the pool / all-to-all / dense part of the step, instead of in front of the table
        # update in the backward (KRS_SHARD_PLAN_AHEAD=0 for an A/B).  One workspace per group is kept across steps.
        self.plan_ahead = bool(int(os.environ.get("KRS_SHARD_PLAN_AHEAD", "1")))
        self._plan_stream = None
        self._plan_ws: dict = {}          # group -> workspace tensor kept across steps
        self._plan_busy: dict = {}        # group -> a pending backward still owns that workspace
        self.plans_ahead = 0              # lookups whose plan ran ahead (tests / diagnostics)
        self._prefetched: dict = {}       # group -> what prefetch() ran ahead for the next call
        self.prefetch_hits = 0
        self.slab_grad_gathers = 0        # backward passes that gathered the segment gradients out of the slab gradient
        self._grow_luts: dict = {}        # (batch, features, lead slots, device) -> row of (sample, feature) in the slab gradient
        self._host_counts = None
        self._collectives_at_world1 = False   # bench --rccl-self: run the collectives through a one-rank communicator
        self._err_dev = self._err_host = self._err_event = None
        self.last_exchange: dict = {}    # host-side counts of the last lookup (tests / load-balance diagnostics)
        self._step_constants: dict = {}  # group -> embedding_ops.StepConstants (scheduled rates / Adam bias correction)

    # single-group conveniences (the common case: one width, one optimizer) -- kept for callers / tests
    @property
    def dim(self) -> int:
        return self._sgroups[0].dim

    @property
    def shard(self):
        return getattr(self, self._sgroups[0].pname)

    # ---------------------------------------------------------------- tables
    def build(self, *_):
        if self.built:
            return
        for g in self._sgroups:
            rows = g.row_off[-1]
            shard = torch.zeros((rows, g.dim), dtype=self.variable_dtype, device=self._device)
            for t, tc in enumerate(g.table_configs):
                # rank r holds global rows r, r+N, r+2N, ...: initialise the full table deterministically
                # only when it is small; otherwise draw the local rows directly
                n_local = len(range(self.rank, tc.vocabulary_size, self.world))
                init = base.get_initializer(tc.initializer)
                if tc.vocabulary_size * g.dim <= (1 << 24):
                    full = init((tc.vocabulary_size, g.dim), self.variable_dtype, self._device)
                    shard[g.row_off[t]: g.row_off[t] + n_local] = full[self.rank::self.world]
                else:
                    shard[g.row_off[t]: g.row_off[t] + n_local] = init((n_local, g.dim), self.variable_dtype,
                                                                     self._device)
            p = torch.nn.Parameter(shard, requires_grad=False)
            setattr(self, g.pname, p)
            self._weight_order.append(p)
            slot = g.fused.new_slot((rows, g.dim), self._device)
            if slot is not None:
                self.register_buffer(g.sname, slot, persistent=True)   # optimizer state is module state
        if self._replicated is not None:
            self._replicated.build(None)
        self._anchor = torch.zeros((), device=self._device, requires_grad=True)
        self.built = True

    def _slot(self, g: _ShardGroup):
        return self._buffers.get(g.sname)

    def get_extra_state(self):
        return {"iterations": [int(g.step) for g in self._sgroups]}

    def set_extra_state(self, state) -> None:
        for g, s in zip(self._sgroups, (state or {}).get("iterations", [])):
            g.step = int(s)

    def _all_gather_rows(self, mine: torch.Tensor) -> list:
        if self.virtual:
            raise L.KrsError("ShardedDistributedEmbedding(virtual_world=N) holds one rank's shard only: tables cannot be assembled")
        if self.world == 1:
            return [mine]
        mine = mine.contiguous()
        staged = mine.is_cuda and dist.get_backend(self._pg) == "gloo"   # gloo gathers host tensors only
        src = mine.cpu() if staged else mine
        parts = [torch.empty_like(src) for _ in range(self.world)]
        dist.all_gather(parts, src, group=self._pg)
        return [q.to(mine.device) for q in parts] if staged else parts

    def get_embedding_tables(self) -> dict[str, torch.Tensor]:
        """Unsharded [V, D] tables by name (all-gather + un-interleave), base:810-825 contract."""
        if not self.built:
            self.build()
        self.check_ids(wait=True)
        out = {}
        for g in self._sgroups:
            parts = self._all_gather_rows(getattr(self, g.pname).data)
            for t, tc in enumerate(g.table_configs):
                full = torch.empty((tc.vocabulary_size, g.dim), dtype=parts[0].dtype, device=parts[0].device)
                for r in range(self.world):
                    n_local = len(range(r, tc.vocabulary_size, self.world))
                    full[r::self.world] = parts[r][g.row_off[t]: g.row_off[t] + n_local]
                out[tc.name] = full
        if self._replicated is not None:
            out.update(self._replicated.get_embedding_tables())
        return out

    def set_embedding_tables(self, tables: dict) -> None:
        if not self.built:
            self.build()
        with torch.no_grad():
            for g in self._sgroups:
                shard = getattr(self, g.pname)
                for t, tc in enumerate(g.table_configs):
                    if tc.name in tables:
                        full = torch.as_tensor(np.asarray(tables[tc.name].detach().cpu() if isinstance(
                            tables[tc.name], torch.Tensor) else tables[tc.name])).to(shard.dtype).to(shard.device)
                        mine = full[self.rank::self.world]
                        shard[g.row_off[t]: g.row_off[t] + mine.shape[0]] = mine
        if self._replicated is not None:
            self._replicated.set_embedding_tables(tables)

    # ---------------------------------------------------------------- out-of-range ids (as DistributedEmbedding)
    def _err_flag(self, device):
        if device.type != "cuda":
            return None
        if self._err_dev is None:
            self._err_dev = torch.zeros(1, dtype=torch.int32, device=device)
            self._err_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        return self._err_dev

    def check_ids(self, wait: bool = False) -> None:
        """IndexError if an earlier lookup met an id outside [0, vocabulary_size) of its table (such lookups are
        dropped before the exchange: they reach no row of any table on any rank)."""
        if self._replicated is not None:
            self._replicated.check_ids(wait)
        if base.stream_capturing():
            return
        ev = self._err_event
        if ev is None:
            if not (wait and getattr(self, "_err_in_graph", False)):
                return
            torch.cuda.current_stream(self._err_dev.device).synchronize()   # replays of a captured step
        elif wait:
            ev.synchronize()
        elif not ev.query():
            return
        self._err_event = None
        if int(self._err_host.item()) & L.FLAG_ID_OUT_OF_RANGE:
            self._err_dev.zero_()
            self._err_host.zero_()
            raise IndexError("ShardedDistributedEmbedding: an embedding id was out of range for its table "
                             "(ids are never clamped; the lookup was dropped)")

    # ---------------------------------------------------------------- inputs
    def preprocess(self, inputs: dict, weights: dict | None = None, training: bool = False):
        """{feature: ids} -> per group one feature-major id buffer (+ CSR offsets when any feature is ragged)."""
        from keras_rs_amd.layers.distributed_embedding import _ragged_numpy_to_csr
        from keras_rs_amd.layers.embed_reduce import Ragged

        if not self.built:
            self.build()
        pre: dict = {"groups": []}
        for g in self._sgroups:
            dev = getattr(self, g.pname).device
            parts, wparts, hots, lens, batch = [], [], [], [], None
            ragged = False
            for p in g.paths:
                x = inputs[p]
                w = None if weights is None else weights[p]
                x, w = _ragged_numpy_to_csr(x, w)
                if isinstance(x, Ragged):
                    ragged = True
                    vals = x.values if isinstance(x.values, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x.values))
                    offs = np.asarray(x.row_offsets.cpu() if isinstance(x.row_offsets, torch.Tensor) else x.row_offsets,
                                      dtype=np.int64)
                    b = len(offs) - 1
                    t = vals.reshape(-1)
                    hots.append(None)
                    lens.append(np.diff(offs))
                    if w is not None:
                        w = w.values if isinstance(w, Ragged) else w
                else:
                    t = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))
                    if t.dim() == 1:
                        t = t.reshape(-1, 1)
                    b = t.shape[0]
                    hots.append(int(t.shape[1]))
                    lens.append(np.full(b, t.shape[1], dtype=np.int64))
                batch = b if batch is None else batch
                if b != batch:
                    raise ValueError("all features must share the batch size")
                parts.append(t.reshape(-1).to(torch.int64 if t.dtype == torch.int64 else torch.int32))
                if weights is not None:
                    w = w if isinstance(w, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(w))
                    wparts.append(w.reshape(-1).float())
            dt = torch.int64 if any(q.dtype == torch.int64 for q in parts) else torch.int32
            ids = torch.cat([q.to(dt) for q in parts]).to(dev, non_blocking=True)
            w = torch.cat(wparts).to(dev, non_blocking=True) if weights is not None else None
            offsets = None
            if ragged:
                offsets = torch.from_numpy(np.concatenate([[0], np.cumsum(np.concatenate(lens))]).astype(np.int64)).to(dev)
            pre["groups"].append({"ids": ids, "hots": None if ragged else tuple(hots), "batch": batch,
                                  "offsets": offsets, "weights": w})
        if self._replicated is not None:
            rp = [p for p in self._paths if self._where[p][0] == "rep"]
            pre["replicated"] = self._replicated.preprocess({p: inputs[p] for p in rp},
                                                            None if weights is None else {p: weights[p] for p in rp})
        return {"preprocessed_inputs_per_placement": {"sparsecore": pre}}

    # ---------------------------------------------------------------- step
    def call(self, inputs, weights=None, training: bool = False):
        self.check_ids()
        if not (isinstance(inputs, dict) and "preprocessed_inputs_per_placement" in inputs):
            inputs = self.preprocess(inputs, weights, training)
        pre = inputs["preprocessed_inputs_per_placement"]["sparsecore"]
        out: dict = {}
        single = len(self._sgroups) == 1 and self._replicated is None
        for gi, (g, fi) in enumerate(zip(self._sgroups, pre["groups"])):
            lead = self.slab_lead_cols if single else 0
            slab, *outs = _ShardedLookupFn.apply(self, gi, fi["ids"], fi["batch"], fi["hots"], fi["offsets"],
                                                 fi["weights"], lead, self._anchor)
            for i, o in enumerate(outs):
                o._krs_slab = (slab, lead + i * g.dim, len(outs), lead)
            out.update(zip(g.paths, outs))
        if self._replicated is not None:
            out.update(self._replicated(pre["replicated"]))
        if self._err_dev is not None:
            self._err_host.copy_(self._err_dev, non_blocking=True)
            if base.stream_capturing():
                self._err_event, self._err_in_graph = None, True
            else:
                self._err_event = torch.cuda.Event()
                self._err_event.record()
        return {p: out[p] for p in self._paths}

    def _a2a(self, send: torch.Tensor, send_counts: list | None = None, recv_counts: list | None = None) -> torch.Tensor:
        """all-to-all of the leading dimension; without counts every pair exchanges send.shape[0] / world rows."""
        n_recv = send.shape[0] if recv_counts is None else sum(recv_counts)
        recv = torch.empty((n_recv,) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
        if self.virtual or (self.world == 1 and not (dist.is_initialized() and self._collectives_at_world1)):
            recv.copy_(send)     # dry run on one GPU: a device copy stands in for the links
        elif send.is_cuda and dist.get_backend(self._pg) == "gloo":
            # gloo has no device all-to-all: stage through the host (debugging / single-GPU test rigs only;
            # production runs use the nccl = RCCL backend)
            host = torch.empty(recv.shape, dtype=recv.dtype)
            dist.all_to_all_single(host, send.contiguous().cpu(), output_split_sizes=recv_counts,
                                   input_split_sizes=send_counts, group=self._pg)
            recv.copy_(host)
        else:
            dist.all_to_all_single(recv, send.contiguous(), output_split_sizes=recv_counts,
                                   input_split_sizes=send_counts, group=self._pg)
        return recv

    def _exchange_sizes(self, mine: torch.Tensor):
        """mine: [3, N] int64 per-owner counts on the device.  Returns (mine, theirs) as host lists of lists:
        theirs[k][s] = what rank s sends here.  One host wait (page-locked flag) on the GPU path."""
        n = self.world
        if n > 1:
            theirs = self._a2a(mine.t().contiguous(), [1] * n, [1] * n).t().contiguous()   # rows = per-rank triples
        else:
            theirs = mine
        both = torch.stack([mine, theirs]).reshape(-1)
        if both.is_cuda and isinstance(self.kernels, HipShardKernels):
            if self._host_counts is None:
                self._host_counts = _HostCounts()
            flat = self._host_counts.read(both)
        else:
            flat = both.cpu().tolist()
        a = np.asarray(flat, dtype=np.int64).reshape(2, 3, n)
        return a[0].tolist(), a[1].tolist()

    def _route_desc(self, g: _ShardGroup, batch: int, hots) -> np.ndarray:
        desc = np.zeros(len(g.paths), dtype=L.SHARD_FEATURE_DT)
        base_pos = 0
        for i, p in enumerate(g.paths):
            tc = g.table_configs[g.table_of_feature[i]]
            hot = 0 if hots is None else int(hots[i])
            desc[i] = (base_pos, g.row_off[g.table_of_feature[i]] * self.world, hot, L.COMBINERS[tc.combiner],
                       tc.vocabulary_size, 0)
            base_pos += batch * hot
        return desc

    # ---------------------------------------------------------------- static-capacity exchange
    def _capacity(self, gi, g, batch, hots, nnz):
        """[cap_lookups, cap_segments] of one (home, owner) block for this group and batch shape, or None when the
        static form cannot be sized (ragged bags without an explicit capacity) -> exact exchange."""
        key = (gi, batch, None if hots is None else tuple(hots))
        cap = self._caps.get(key)
        if cap is not None:
            return cap, key
        n, spec = self.world, self._capacity_spec
        up4 = lambda v: int(-(-int(v) // 4) * 4)   # noqa: E731
        n_bags = batch * len(g.paths)
        if isinstance(spec, (tuple, list)):
            cap = [up4(spec[0]), up4(spec[1])]
        elif spec == "table_config":
            tcs = [g.table_configs[t] for t in g.table_of_feature]
            for tc in tcs:
                for field in ("max_ids_per_partition", "max_unique_ids_per_partition"):
                    v = getattr(tc, field, None)
                    if v is None or int(v) <= 0:
                        raise ValueError(f"capacity='table_config': table '{tc.name}' has {field}={v!r}; the static "
                                         "exchange sizes its blocks from these limits (distributed_embedding_config.py:"
                                         "54-61) and needs a positive value on every sharded table")
            cap = [up4(sum(tc.max_ids_per_partition for tc in tcs)), up4(sum(tc.max_unique_ids_per_partition for tc in tcs))]
        elif spec == "auto" and hots is not None:
            h = self.capacity_headroom
            exp_l = nnz / n
            exp_s = sum(batch * (1.0 - (1.0 - 1.0 / n) ** hot) for hot in hots)
            cap = [min(up4(nnz), up4(h * exp_l + 64)), min(up4(n_bags), up4(h * exp_s + 64))]
        else:
            return None, key
        cap = [max(cap[0], 4), max(cap[1], 4)]
        self._caps[key] = cap
        return cap, key

    def _note_stats(self, gi, stats, key):
        """Running statistics of the static exchange (the reference's update_stats): `stats` = device int64[4] of
        this step's unpack (max need_l / need_s over ALL ranks' blocks: identical on every rank).  They are copied to
        page-locked memory behind an event and looked at two steps later -- long complete, so no wait -- on every
        rank at the same step: a need above the capacity grows it (x 1.125) for the steps that follow."""
        if stats.is_cuda and base.stream_capturing():
            # a captured step: the copy is a node of every replay, nothing is polled from inside the step;
            # poll_exchange_stats() between replays looks at the latest figures (a capacity that grows changes buffer
            # shapes: the step must then be captured again)
            ring = self._stat_bufs.get(gi)
            if not ring:
                raise L.KrsError("ShardedDistributedEmbedding: run the step eagerly once before capturing it "
                                 "(page-locked memory cannot be allocated while a stream is capturing)")
            self._graph_stats[gi] = (ring[0], key)
            ring[0].copy_(stats, non_blocking=True)
            return
        step = self._stat_step.get(gi, 0)
        self._stat_step[gi] = step + 1
        q = self._stats_q.setdefault(gi, [])
        if stats.is_cuda:
            ring = self._stat_bufs.setdefault(gi, [])
            if len(ring) < 4:     # (a buffer is read two steps after it was filled: four never collide)
                ring.append(torch.empty(4, dtype=torch.int64).pin_memory())
            host = ring[step % 4] if len(ring) == 4 else ring[-1]
            host.copy_(stats, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        else:
            host, ev = stats.clone(), None
        q.append((step, ev, host, key))
        while q and q[0][0] <= step - 2:
            _, ev0, h0, key0 = q.pop(0)
            if ev0 is not None:
                ev0.synchronize()
            self._apply_stats(h0, key0)

    def _apply_stats(self, h0, key0, allow_shrink: bool = True) -> bool:
        """One step's statistics (host copy) against the capacity they ran with; True when the capacity grew.
        `allow_shrink=False` (statistics of a replayed graph): fitting steps never resize."""
        need_l, need_s = int(h0[0]), int(h0[1])
        cap = self._caps.get(key0)
        self.last_exchange.update(need=(need_l, need_s), received=(int(h0[2]), int(h0[3])))
        if cap is None:
            return False
        if need_l <= cap[0] and need_s <= cap[1]:
            # Settled statistics shrink the blocks (every slot of a block crosses the link, padding included): once
            # `capacity_settle_steps` steps in a row fit, the capacity drops to the largest need they showed + 6 % + 64
            # (rounded to 64) -- every rank sees the same maxima at the same step, so all ranks resize together.
            if allow_shrink and self.update_stats and self.capacity_settle_steps > 0 and self._capacity_spec == "auto":
                w = self._settle.setdefault(key0, [0, 0, 0])
                w[0], w[1], w[2] = w[0] + 1, max(w[1], need_l), max(w[2], need_s)
                if w[0] >= self.capacity_settle_steps:
                    up64 = lambda v: int(-(-int(v) // 64) * 64)   # noqa: E731
                    new = [min(cap[0], up64(w[1] * 1.0625 + 64)), min(cap[1], up64(w[2] * 1.0625 + 64))]
                    self._settle[key0] = [0, 0, 0]
                    if new != cap:
                        cap[0], cap[1] = new
                        self.capacity_shrinks += 1
            return False
        self._settle.pop(key0, None)
        self.overflow_steps += 1
        if not self.update_stats:
            return False
        up64 = lambda v: int(-(-int(v) // 64) * 64)   # noqa: E731
        if need_l > cap[0]:
            cap[0] = up64(need_l * 1.125)
        if need_s > cap[1]:
            cap[1] = up64(need_s * 1.125)
        return True

    def flush_exchange_stats(self) -> int:
        """Looks at the statistics of the steps still in the two-step queue (waits for them): `overflow_steps` is then
        final for everything run so far.  Returns it."""
        for q in self._stats_q.values():
            while q:
                _, ev0, h0, key0 = q.pop(0)
                if ev0 is not None:
                    ev0.synchronize()
                self._apply_stats(h0, key0)
        return self.overflow_steps

    def poll_exchange_stats(self) -> bool:
        """For steps replayed from a graph (torch.cuda.graph around the step): waits for the device, looks at the
        statistics of the last replay and grows the capacities they exceed.  True = a capacity grew -- lookups of that
        replay were dropped (`overflow_steps`), and the step has to be captured again, its buffers changed size."""
        if not self._graph_stats:
            return False
        torch.cuda.current_stream(self._device).synchronize()
        grew = False
        for host, key in self._graph_stats.values():
            grew = self._apply_stats(host, key, allow_shrink=False) or grew
        return grew

    def _route_exchange(self, g, cap_l, cap_s, ids, batch, hots, offsets, weights, emit_w):
        """The part of a static-capacity lookup that depends on the ids alone: route -> all-to-all of the fixed-size
        id blocks -> unpack (what prefetch() runs ahead of the step)."""
        k, n = self.kernels, self.world
        off_rank = (n - 1) / n if n > 1 else 1.0
        # (probe.span: per-phase event pairs when bench.py is probing -- `phases` of its line; otherwise one global read)
        with probe.span("route"):
            r = k.route_static(self._route_desc(g, batch, hots), ids, offsets, weights, batch, n, emit_w, cap_l, cap_s,
                               self._err_flag(ids.device))
        with probe.span("a2a_ids", off_rank * 4 * n * r["packed"].shape[1]):
            recv_packed = self._a2a(r["packed"])                                # [n, W]: equal splits, no counts
        with probe.span("unpack"):
            rows, w, off, stats = k.unpack_static(recv_packed, cap_l, cap_s, emit_w)
        return r, rows, w, off, stats

    def prefetch(self, inputs, weights=None, training: bool = False):
        """Runs the id side of a LATER call now, on the layer's exchange stream: krs_shard_route -> the all-to-all of the
        id blocks -> krs_shard_unpack depend on the ids only (not on the tables), so the next step's can run under this
        step's backward pass instead of at the head of the next forward.  Static exchange only (the exact form sizes its
        all-to-all with a host wait).  Takes raw inputs or the result of preprocess(); returns the preprocessed inputs to
        hand to the call.  Every rank must call it at the same point of its step (the collectives of one communicator
        run in issue order).  The analogue in the reference: the host-side preprocessing + enqueue of batch i+1 while the
        SparseCore step of batch i runs (jax/distributed_embedding.py:405-462, embedding_lookup.py:134-147)."""
        if not (isinstance(inputs, dict) and "preprocessed_inputs_per_placement" in inputs):
            inputs = self.preprocess(inputs, weights, training)
        if self.exchange != "static" or base.stream_capturing():
            return inputs
        pre = inputs["preprocessed_inputs_per_placement"]["sparsecore"]
        for gi, (g, fi) in enumerate(zip(self._sgroups, pre["groups"])):
            ids, batch, hots = fi["ids"], fi["batch"], fi["hots"]
            cap, _ = self._capacity(gi, g, batch, hots, ids.numel())
            if cap is None:
                continue
            emit_w = fi["weights"] is not None or any(g.table_configs[t].combiner != "sum" for t in g.table_of_feature)
            cap_l, cap_s = cap
            if ids.is_cuda:
                main = torch.cuda.current_stream(ids.device)
                xs = self._xstream
                if xs is None:
                    xs = self._xstream = torch.cuda.Stream(device=ids.device)
                self._err_flag(ids.device)      # (allocated on the main stream, before the fork)
                xs.wait_stream(main)            # the ids (and the previous call's use of the descriptors) are ordered first
                with torch.cuda.stream(xs):
                    r, rows, w, off, stats = self._route_exchange(g, cap_l, cap_s, ids, batch, hots, fi["offsets"],
                                                                  fi["weights"], emit_w)
                    ev = torch.cuda.Event()
                    ev.record(xs)
                for t in (ids, fi["offsets"], fi["weights"]):
                    if t is not None:
                        t.record_stream(xs)
            else:
                r, rows, w, off, stats = self._route_exchange(g, cap_l, cap_s, ids, batch, hots, fi["offsets"],
                                                              fi["weights"], emit_w)
                ev = None
            self._prefetched[gi] = dict(ids=ids, cap=(cap_l, cap_s), emit_w=emit_w, r=r, rows=rows, w=w, off=off,
                                        stats=stats, event=ev)
        return inputs

    def _plan_ahead(self, gi, g, rows, off, saved) -> None:
        """Starts the backward plan of this lookup on the plan stream (see `plan_ahead`); the backward joins it."""
        k = self.kernels
        if not (self.plan_ahead and rows.is_cuda and getattr(self, "_backward_follows", False) and hasattr(k, "plan_segments")
                and rows.numel() and off.numel() > 1):
            return
        dev = rows.device
        shard, slot = getattr(self, g.pname).data, self._slot(g)
        k.prepare_plan(shard, slot, rows, off)                 # descriptor uploads on the main stream, before the fork
        main = torch.cuda.current_stream(dev)
        side = self._plan_stream
        if side is None:
            side = self._plan_stream = torch.cuda.Stream(device=dev)
        side.wait_stream(main)
        busy = self._plan_busy.get(gi, False)
        keep = None if busy else self._plan_ws.get(gi)
        with torch.cuda.stream(side):
            ws = k.plan_segments(shard, slot, rows, off, ws=keep)
            ev = torch.cuda.Event()
            ev.record(side)
        owns = [False]
        if not busy and ws is not None:
            self._plan_ws[gi], self._plan_busy[gi] = ws, True
            owns[0] = True
        for t in (ws, rows, off):
            if t is not None:
                t.record_stream(side)
        saved["plan"] = (ws, ev, owns)
        self.plans_ahead += 1
        if base.stream_capturing():
            from keras_rs_amd import graphs

            graphs.join_at_capture_end(side)     # (a captured step without its backward would leave the fork open)

    def _release_plan(self, gi, owns) -> None:
        if owns[0]:
            owns[0] = False
            self._plan_busy[gi] = False

    def _forward_static(self, gi, g, cap, key, ids, batch, hots, offsets, weights, lead, emit_w):
        k, n = self.kernels, self.world
        cap_l, cap_s = cap
        dev = ids.device
        n_feats = len(g.paths)
        shard = getattr(self, g.pname).data
        off_rank = (n - 1) / n if n > 1 else 1.0   # one-GPU dry run: what the stand-in copies move
        pf = self._prefetched.pop(gi, None)
        if pf is not None and (pf["ids"].data_ptr() != ids.data_ptr() or pf["ids"].numel() != ids.numel() or
                               pf["ids"]._version != ids._version or pf["cap"] != (cap_l, cap_s) or pf["emit_w"] != emit_w):
            pf = None      # another batch, or the capacity moved since: the prefetched blocks do not fit this call
        if pf is not None:
            # route / id all-to-all / unpack of this call ran ahead on the exchange stream (prefetch()): join it here
            r, rows, w, off, stats = pf["r"], pf["rows"], pf["w"], pf["off"], pf["stats"]
            if pf["event"] is not None:
                cur = torch.cuda.current_stream(dev)
                cur.wait_event(pf["event"])
                for t in (*r.values(), rows, w, off, stats):
                    if t is not None:
                        t.record_stream(cur)
            self.prefetch_hits += 1
        else:
            r, rows, w, off, stats = self._route_exchange(g, cap_l, cap_s, ids, batch, hots, offsets, weights, emit_w)
        words = r["packed"].shape[1]
        pdt = self._partial_dtype or self.compute_dtype
        if isinstance(pdt, str):
            pdt = {"float32": torch.float32, "bfloat16": torch.bfloat16}[pdt]
        with probe.span("pool"):
            partial = k.pool_segments(shard, rows, off, w, pdt)                   # [n * cap_s, dim]: one per segment slot
        with probe.span("a2a_partials", off_rank * partial.numel() * partial.element_size()):
            back = self._a2a(partial)
        with probe.span("combine"):
            slab = torch.empty((batch, lead + n_feats * g.dim), dtype=back.dtype, device=dev)
            k.combine(back, r["bag_seg"], batch, n_feats, g.dim, slab[:, lead:])
            if slab.dtype != self.compute_dtype:
                slab = slab.to(self.compute_dtype)
        es = back.element_size()
        self.last_exchange = dict(mode="static", capacity=(cap_l, cap_s),
                                  bytes=dict(ids_fwd=4 * n * words, partials_fwd=n * cap_s * g.dim * es,
                                             grads_bwd=n * cap_s * g.dim * es),
                                  bytes_per_step=int(off_rank * (4 * n * words + 2 * n * cap_s * g.dim * es)),
                                  overflow_steps=self.overflow_steps)
        self._note_stats(gi, stats, key)
        saved = dict(batch=batch, seg_grow=r["seg_grow"], send_segs=None, recv_segs=None,
                     rows=rows, off=off, w=w, pdt=pdt, out_meta=(slab.dtype, slab.device))
        self._plan_ahead(gi, g, rows, off, saved)
        return slab, saved

    def _forward_impl(self, gi, ids, batch, hots, offsets, weights, lead):
        k, n, g = self.kernels, self.world, self._sgroups[gi]
        dev = ids.device
        n_feats = len(g.paths)
        shard = getattr(self, g.pname).data
        emit_w = weights is not None or any(g.table_configs[t].combiner != "sum" for t in g.table_of_feature)
        if self.exchange == "static":
            cap, key = self._capacity(gi, g, batch, hots, ids.numel())
            if cap is not None:
                return self._forward_static(gi, g, cap, key, ids, batch, hots, offsets, weights, lead, emit_w)
        with probe.span("route"):
            r = k.route(self._route_desc(g, batch, hots), ids, offsets, weights, batch, n, emit_w, self._err_flag(dev))
        with probe.span("counts_host_wait"):
            mine, theirs = self._exchange_sizes(r["counts"])
        send_cnt, send_segs, send_words = mine
        recv_cnt, recv_segs, recv_words = theirs
        n_seg = sum(send_segs)
        es_p = 2 if (self._partial_dtype or self.compute_dtype) in (torch.bfloat16, "bfloat16") else 4
        own = self.rank if n > 1 else -1      # the block a rank sends itself crosses no link
        off_words = sum(wd for d, wd in enumerate(send_words) if d != own)
        off_segs = sum(sg for d, sg in enumerate(send_segs) if d != own)
        self.last_exchange = dict(mode="exact", send_lookups=send_cnt, recv_lookups=recv_cnt, send_segments=send_segs,
                                  recv_segments=recv_segs,
                                  bytes=dict(ids_fwd=4 * sum(send_words), partials_fwd=n_seg * g.dim * es_p,
                                             grads_bwd=n_seg * g.dim * es_p),
                                  bytes_per_step=int(4 * off_words + 2 * off_segs * g.dim * es_p) if n > 1 else
                                  int(4 * sum(send_words) + 2 * n_seg * g.dim * es_p))
        # to the owners: ONE packed buffer (rows | weights | segment lengths per owner)
        with probe.span("a2a_ids", 4 * (off_words if n > 1 else sum(send_words))):
            recv_packed = self._a2a(r["packed"][:sum(send_words)], send_words, recv_words)
        with probe.span("unpack"):
            rows, w, off = k.unpack(recv_packed, recv_cnt, recv_segs, emit_w)
        pdt = self._partial_dtype or self.compute_dtype
        if isinstance(pdt, str):
            pdt = {"float32": torch.float32, "bfloat16": torch.bfloat16}[pdt]
        with probe.span("pool"):
            partial = k.pool_segments(shard, rows, off, w, pdt)
        with probe.span("a2a_partials", (off_segs if n > 1 else n_seg) * g.dim * es_p):
            back = self._a2a(partial, recv_segs, send_segs)                       # home side, segment order
        with probe.span("combine"):
            slab = torch.empty((batch, lead + n_feats * g.dim), dtype=back.dtype, device=dev)
            k.combine(back, r["bag_seg"], batch, n_feats, g.dim, slab[:, lead:])
            if slab.dtype != self.compute_dtype:
                slab = slab.to(self.compute_dtype)
        saved = dict(batch=batch, seg_grow=r["seg_grow"][:n_seg], send_segs=send_segs, recv_segs=recv_segs,
                     rows=rows, off=off, w=w, pdt=pdt, out_meta=(slab.dtype, slab.device))
        self._plan_ahead(gi, g, rows, off, saved)
        return slab, saved

    def _backward_impl(self, gi, grad, s, slab_grad=None, lead_slots=0):
        k, g = self.kernels, self._sgroups[gi]
        n_feats, batch = len(g.paths), s["batch"]
        # d(partial of a segment) = the output gradient of its bag: rows of grad viewed as [batch * n_feats, dim]
        with probe.span("gather_grads"):
            if slab_grad is not None:
                self.slab_grad_gathers += 1
                # row of (sample b, feature f) in the slab gradient seen as [B * (n + ls), dim]: b * (n + ls) + ls + f
                # = seg_grow + (seg_grow // n + 1) * ls
                # (one gather through a cached table of the B * n remapped row numbers instead of three elementwise ops)
                key = (batch, n_feats, lead_slots, str(slab_grad.device))
                lut = self._grow_luts.get(key)
                if lut is None:
                    r = torch.arange(batch * n_feats, dtype=torch.int32, device=slab_grad.device)
                    lut = self._grow_luts[key] = r + (torch.div(r, n_feats, rounding_mode="floor") + 1) * lead_slots
                grow = torch.index_select(lut, 0, s["seg_grow"])
                dpart = k.gather_rows(slab_grad.view(batch * (n_feats + lead_slots), g.dim), grow)
            else:
                grad = grad.contiguous()
                if grad.dtype != s["pdt"]:
                    grad = grad.to(s["pdt"])
                dpart = k.gather_rows(grad.view(batch * n_feats, g.dim), s["seg_grow"])
        off_rank = (self.world - 1) / self.world if self.world > 1 else 1.0
        with probe.span("a2a_grads", off_rank * dpart.numel() * dpart.element_size()):
            dseg = self._a2a(dpart, s["send_segs"], s["recv_segs"])               # to the owners
        if (callable(g.fused.lr) or g.fused.kind == "adam") and getattr(k, "device_step_constants", False):
            # constants that depend on the update count live in device memory (embedding_ops.StepConstants): written here by an
            # eager step, before every replay by GraphedStep when this backward is being captured
            sc = self._step_constants.get(gi)
            if sc is None:
                from keras_rs_amd.embedding_ops import StepConstants

                sc = self._step_constants[gi] = StepConstants(
                    g, lambda g=g: k._bags_for(getattr(self, g.pname).data, self._slot(g), 0.0),
                    (lambda step, g=g: [g.fused.lr_at(step)]) if callable(g.fused.lr) else None,
                    g.fused.consts[:2] if g.fused.kind == "adam" else None)
            sc.on_backward()
            lr = None if callable(g.fused.lr) else g.fused.lr_at(g.step)
            hyper = g.fused.consts + (sc.bias_correction,) if g.fused.kind == "adam" else g.fused.hyper(g.step)
        else:
            from keras_rs_amd import graphs

            lr = g.fused.lr_at(g.step)
            if callable(g.fused.lr) or g.fused.kind == "adam":
                g.step += 1               # (test kernels without device-resident constants: by-value arguments, eager only)
            else:
                graphs.count_update(g)    # (per replay under GraphedStep)
            hyper = g.fused.hyper(max(g.step, 1))
        plan = s.get("plan")
        ws = None
        if plan is not None:
            ws, ev, owns = plan
            cur = torch.cuda.current_stream(s["rows"].device)
            cur.wait_event(ev)                      # the plan ran ahead on the plan stream: join it here
            if ws is not None:
                ws.record_stream(cur)
        with probe.span("k2"):
            if ws is not None:
                k.apply_segments(getattr(self, g.pname).data, self._slot(g), s["rows"], s["off"], s["w"], dseg, lr,
                                 g.fused.kind, hyper, 1.0 / self.world if self.grad_average else 1.0, ws=ws)
            else:     # (kernels without plan_segments -- the oracle-backed test kernels -- plan inside the call)
                k.apply_segments(getattr(self, g.pname).data, self._slot(g), s["rows"], s["off"], s["w"], dseg, lr,
                                 g.fused.kind, hyper, 1.0 / self.world if self.grad_average else 1.0)
        if plan is not None:
            self._release_plan(gi, plan[2])       # (the next forward's plan is ordered behind this apply: it may take the buffer)
