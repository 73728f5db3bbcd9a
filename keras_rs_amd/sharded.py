"""Row-sharded embedding tables across the GPUs of one node (config C4 / SURVEY.md section 8e).

Layout (the reference's own convention, sharding_strategy="MOD" at
keras_rs/src/layers/embedding/jax/embedding_utils.py:194, reassembly code at
tensorflow/distributed_embedding.py:316-328): global row r of every table lives on rank
r % N at local row r // N.  Each rank keeps ONE stacked buffer [sum_t ceil(V_t / N), D]; table t
starts at row off_t = sum_{s<t} ceil(V_s / N), so a lookup anywhere in the group is `off_t + r // N`.

Per step (one process per GPU, torch.distributed over RCCL/xGMI; xGMI is point-to-point, so the
exchange is an all-to-all whose pairs each use their own link).  What crosses the links is one
PARTIALLY POOLED vector per (bag, owner) pair that has lookups -- not one vector per lookup: with
the ml_perf bag lengths (214 lookups per sample) that is 41 vectors per sample at N = 2 and 79 at
N = 8.
  fwd  1. composite id c = off_t*N + r  (c % N = owner, c // N = stacked local row)
       2. K5 MOD-bucketise c (stable: inside a bucket the lookups stay in bag order); runs of equal
          bag inside a bucket are the SEGMENTS; combiner scale (mean / sqrtn) and user weights are
          folded into one weight per lookup; all-to-all of (lookup, segment) counts, all-to-all-v of
          local rows, segment lengths and weights
       3. owner: K1 in CSR form over the received segments -> one partial vector per segment
       4. all-to-all-v of the partial vectors back to the sample's home rank
       5. home: K1 again, with the partials as the "table": every bag sums its <= N partials
  bwd  mirror image: d(partial) of a segment is its bag's output gradient (a row gather),
       all-to-all-v to the owners, K2 fused SGD / Adagrad on the shard in CSR form.  Every row has
       exactly one owner, so table gradients need no cross-GPU reduction.

The exchange logic is independent of the compute kernels: `kernels` is an object with the four
methods of HipShardKernels.  The product default runs the HIP kernels; the CPU/gloo tests in
tests/test_sharded_gloo.py inject an oracle-backed implementation to check the permutation and
collective plumbing with world_size 2.
"""

from __future__ import annotations

import math
from typing import Any, Sequence

import numpy as np
import torch
import torch.distributed as dist

from keras_rs_amd.layers import base
from keras_rs_amd.layers.distributed_embedding import resolve_fused_optimizer
from keras_rs_amd.layers.distributed_embedding_config import FeatureConfig


class HipShardKernels:
    """The compute side of the sharded path on MI355X (K1 / K2 / K5 through the C ABI)."""

    def __init__(self):
        self._shard_bags: dict = {}

    def _bags_for(self, table, slot, lr):
        """One FusedBags (device descriptors) per shard storage, re-used across steps."""
        from keras_rs_amd.embedding_ops import FusedBags

        key = (table.data_ptr(), 0 if slot is None else slot.data_ptr(), float(lr))
        fb = self._shard_bags.get(key)
        if fb is None:
            fb = self._shard_bags[key] = FusedBags([table], [(0, "sum", 0)], slots=[slot], lrs=[lr])
        return fb

    def bucketize(self, ids: torch.Tensor, n_shards: int):
        from keras_rs_amd import dense_ops as D

        return D.mod_bucketize(ids, n_shards)

    def _transient_bags(self, table, feats):
        """FusedBags for a per-step `table` (returned vectors, an output gradient): one object per feature
        list is kept and re-pointed, so that only the 32-byte table descriptor is uploaded per call."""
        from keras_rs_amd.embedding_ops import FusedBags

        key = ("transient", tuple(feats), table.shape[1], table.dtype)
        fb = self._shard_bags.get(key)
        if fb is None:
            fb = self._shard_bags[key] = FusedBags([table], list(feats))
        fb.tables = [table]
        fb.row_bases[:] = (0, table.shape[0])
        fb.total_rows = int(table.shape[0])
        return fb

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

    def pool(self, vectors, ids, feats, batch, offsets, out_dtype, out=None):
        """Home side: bags (feature-major CSR over `ids`) summed out of `vectors` into [batch, n_feats*dim]
        (`out`: a row-major buffer of that shape, possibly a column window of a wider one)."""
        fb = self._transient_bags(vectors, feats)
        out, _ = fb.forward(ids, batch, offsets=offsets, out=out, out_dtype=out_dtype)
        fb.tables = []
        return out

    def apply_segments(self, table, slot, rows, offsets, weights, seg_grads, lr, kind, hyper=None):
        """Owner side: fused optimizer step; lookup i of segment s carries weights[i] * seg_grads[s]."""
        n_seg = offsets.numel() - 1
        if rows.numel() == 0 or n_seg == 0:
            return
        fb = self._bags_for(table, slot, 0.0)
        fb.lrs = [float(lr)]   # scheduled rates change per step: the descriptor is re-uploaded when it does
        ws = fb.plan_backward(rows, n_seg, offsets=offsets)
        fb.backward_fused(kind, ws, seg_grads, n_seg, rows.numel(), weights=weights, hyper=hyper)


class _ShardedLookupFn(torch.autograd.Function):
    """Outputs: the slab [B, lead + n*dim] (see DistributedEmbedding.slab_lead_cols) and its n feature views."""

    @staticmethod
    def forward(ctx, layer, ids, batch, hots, offsets, weights, anchor):
        from keras_rs_amd.autograd import _split_columns

        ctx.set_materialize_grads(False)  # unused outputs arrive as None, not as zero tensors
        slab, saved = layer._forward_impl(ids, batch, hots, offsets, weights)
        ctx.layer, ctx.saved = layer, saved
        return (slab,) + _split_columns(slab, len(layer._paths), layer.dim, layer.slab_lead_cols)

    @staticmethod
    def backward(ctx, g_slab, *gs):
        layer = ctx.layer
        from keras_rs_amd.autograd import _sum_slab_and_feature_grads

        out_dtype, device = ctx.saved["out_meta"]
        g = _sum_slab_and_feature_grads(g_slab, gs, layer.slab_lead_cols, ctx.saved["batch"], len(layer._paths),
                                        layer.dim, out_dtype, device)
        layer._backward_impl(g, ctx.saved)
        return (None, None, None, None, None, None, torch.zeros((), device=device))


class ShardedDistributedEmbedding(base.Layer):
    """DistributedEmbedding whose tables are MOD row-sharded over the ranks of `process_group`.

    feature_configs: flat dict {name: FeatureConfig}; all tables share embedding_dim; the
    per-table optimizer (SGD / Adagrad) is fused into the backward as on the single-GPU
    'sparsecore' placement.  call(inputs) takes raw {name: ids} or the result of preprocess()."""

    def __init__(self, feature_configs: dict[str, FeatureConfig], *, process_group=None, kernels=None,
                 slab_lead_cols: int = 0, **kwargs: Any):
        super().__init__(**kwargs)
        self.slab_lead_cols = int(slab_lead_cols)  # as DistributedEmbedding: room for layers.concat_features
        self._pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self.kernels = kernels or HipShardKernels()
        self._feature_configs = feature_configs
        self._paths = list(feature_configs.keys())
        tcs: list = []
        self._table_of_feature = []
        for p in self._paths:
            tc = feature_configs[p].table
            idx = next((i for i, t in enumerate(tcs) if t is tc), None)
            if idx is None:
                idx = len(tcs)
                tcs.append(tc)
            self._table_of_feature.append(idx)
        self._table_configs = tcs
        dims = {tc.embedding_dim for tc in tcs}
        if len(dims) != 1:
            raise NotImplementedError("ShardedDistributedEmbedding: tables must share embedding_dim")
        self.dim = dims.pop()
        kinds = {resolve_fused_optimizer(tc.optimizer) for tc in tcs}
        if None in kinds or len(kinds) != 1:
            raise NotImplementedError("ShardedDistributedEmbedding: one fusable optimizer setting (SGD / Adagrad / "
                                      "Adam / Ftrl, see resolve_fused_optimizer) for all tables")
        self._fused = next(iter(kinds))
        self._opt_kind = self._fused.kind
        self._step = 0
        # local rows of table t sit at [row_off[t], row_off[t] + ceil(V_t / N)) of this rank's stacked buffer
        self._local_rows = [math.ceil(tc.vocabulary_size / self.world) for tc in tcs]
        self._row_off = [0]
        for n_loc in self._local_rows:
            self._row_off.append(self._row_off[-1] + n_loc)
        self._combiners = [feature_configs[p].table.combiner for p in self._paths]
        self.register_parameter("shard", None)
        self._slot = None
        self._anchor = None
        self._offset_cache: dict = {}

    # ---------------------------------------------------------------- tables
    def build(self, *_):
        if self.shard is not None:
            self.built = True
            return
        rows = self._row_off[-1]
        shard = torch.zeros((rows, self.dim), dtype=self.variable_dtype, device=self._device)
        for t, tc in enumerate(self._table_configs):
            # rank r holds global rows r, r+N, r+2N, ...: initialise the full table deterministically
            # only when it is small; otherwise draw the local rows directly
            n_local = len(range(self.rank, tc.vocabulary_size, self.world))
            init = base.get_initializer(tc.initializer)
            if tc.vocabulary_size * self.dim <= (1 << 24):
                full = init((tc.vocabulary_size, self.dim), self.variable_dtype, self._device)
                shard[self._row_off[t]: self._row_off[t] + n_local] = full[self.rank::self.world]
            else:
                shard[self._row_off[t]: self._row_off[t] + n_local] = init((n_local, self.dim), self.variable_dtype,
                                                                     self._device)
        self.shard = torch.nn.Parameter(shard, requires_grad=False)
        self._weight_order.append(self.shard)
        self._slot = self._fused.new_slot((rows, self.dim), self._device)
        self._anchor = torch.zeros((), device=self._device, requires_grad=True)
        self.built = True

    def get_embedding_tables(self) -> dict[str, torch.Tensor]:
        """Unsharded [V, D] tables by name (all-gather + un-interleave), base:810-825 contract."""
        if not self.built:
            self.build()
        if self.world > 1:
            mine = self.shard.data.contiguous()
            staged = mine.is_cuda and dist.get_backend(self._pg) == "gloo"   # gloo gathers host tensors only
            src = mine.cpu() if staged else mine
            parts = [torch.empty_like(src) for _ in range(self.world)]
            dist.all_gather(parts, src, group=self._pg)
            if staged:
                parts = [q.to(mine.device) for q in parts]
        else:
            parts = [self.shard.data]
        out = {}
        for t, tc in enumerate(self._table_configs):
            full = torch.empty((tc.vocabulary_size, self.dim), dtype=self.shard.dtype, device=self.shard.device)
            for r in range(self.world):
                n_local = len(range(r, tc.vocabulary_size, self.world))
                full[r::self.world] = parts[r][self._row_off[t]: self._row_off[t] + n_local]
            out[tc.name] = full
        return out

    def set_embedding_tables(self, tables: dict) -> None:
        if not self.built:
            self.build()
        with torch.no_grad():
            for t, tc in enumerate(self._table_configs):
                if tc.name in tables:
                    full = torch.as_tensor(np.asarray(tables[tc.name])).to(self.shard.dtype).to(self.shard.device)
                    mine = full[self.rank::self.world]
                    self.shard[self._row_off[t]: self._row_off[t] + mine.shape[0]] = mine

    # ---------------------------------------------------------------- inputs
    def preprocess(self, inputs: dict, weights: dict | None = None, training: bool = False):
        """{feature: ids} -> one feature-major id buffer.  A feature is a dense [batch, hot] (or [batch])
        array, or ragged: an embed_reduce.Ragged (values + row offsets) or a numpy object array of rows;
        with any ragged feature the bags are described by CSR offsets instead of `hots`."""
        from keras_rs_amd.layers.distributed_embedding import _ragged_numpy_to_csr
        from keras_rs_amd.layers.embed_reduce import Ragged

        if not self.built:
            self.build()
        dev = self.shard.device
        parts, wparts, hots, lens, batch = [], [], [], [], None
        ragged = False
        for p in self._paths:
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
        ids = torch.cat(parts).to(dev, non_blocking=True)
        w = torch.cat(wparts).to(dev, non_blocking=True) if weights is not None else None
        offsets = None
        if ragged:
            offsets = torch.from_numpy(np.concatenate([[0], np.cumsum(np.concatenate(lens))]).astype(np.int64)).to(dev)
        return {"preprocessed_inputs_per_placement": {"sparsecore": {
            "inputs": {"ids": ids, "hots": None if ragged else tuple(hots), "batch": batch, "offsets": offsets},
            "weights": w}}}

    def _composite_offsets(self, batch, hots, dtype, device):
        key = (batch, hots, dtype, str(device))
        off = self._offset_cache.get(key)
        if off is None:
            per_feat = torch.tensor([self._row_off[self._table_of_feature[i]] * self.world for i in range(len(hots))],
                                    dtype=dtype)
            reps = torch.tensor([batch * h for h in hots])
            off = torch.repeat_interleave(per_feat, reps).to(device)
            self._offset_cache[key] = off
        return off

    # ---------------------------------------------------------------- step
    def call(self, inputs, weights=None, training: bool = False):
        if not (isinstance(inputs, dict) and "preprocessed_inputs_per_placement" in inputs):
            inputs = self.preprocess(inputs, weights, training)
        pre = inputs["preprocessed_inputs_per_placement"]["sparsecore"]
        fi = pre["inputs"]
        slab, *outs = _ShardedLookupFn.apply(self, fi["ids"], fi["batch"], fi["hots"], fi["offsets"],
                                             pre.get("weights"), self._anchor)
        for i, o in enumerate(outs):
            o._krs_slab = (slab, self.slab_lead_cols + i * self.dim, len(outs), self.slab_lead_cols)
        return {p: o for p, o in zip(self._paths, outs)}

    def _a2a(self, send: torch.Tensor, send_counts: list[int], recv_counts: list[int]) -> torch.Tensor:
        recv = torch.empty((sum(recv_counts),) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
        if self.world == 1:
            recv.copy_(send)
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

    def _bag_tables(self, batch, hots, device):
        """Per (batch, hots): bag of every lookup position (feature-major), combiner code of every bag."""
        key = ("bags", batch, hots, str(device))
        got = self._offset_cache.get(key)
        if got is None:
            bag = torch.cat([f * batch + torch.arange(batch, dtype=torch.int32).repeat_interleave(h)
                             for f, h in enumerate(hots)])
            comb = torch.tensor([{"sum": 0, "mean": 1, "sqrtn": 2}[c] for c in self._combiners],
                                dtype=torch.int32).repeat_interleave(batch)
            got = self._offset_cache[key] = (bag.to(device), comb.to(device))
        return got

    def _lookup_weights(self, weights, bag_of_pos, comb_of_bag, n_bags):
        """User weight x combiner scale per lookup (mean: 1 / sum w, sqrtn: 1 / sqrt(sum w^2), both
        divide_no_nan: embed_reduce.py:236-262), or None when every bag is a plain sum."""
        if weights is None and all(c == "sum" for c in self._combiners):
            return None
        w = torch.ones(bag_of_pos.numel(), dtype=torch.float32, device=bag_of_pos.device) if weights is None \
            else weights.float()
        idx = bag_of_pos.long()
        s1 = torch.zeros(n_bags, dtype=torch.float32, device=w.device).index_add_(0, idx, w)
        s2 = torch.zeros(n_bags, dtype=torch.float32, device=w.device).index_add_(0, idx, w * w)
        inv = lambda d: torch.where(d != 0, 1.0 / d, torch.zeros_like(d))  # noqa: E731
        scale = torch.where(comb_of_bag == 1, inv(s1), torch.where(comb_of_bag == 2, inv(s2.sqrt()),
                                                                   torch.ones_like(s1)))
        return w * scale[idx]

    def _forward_impl(self, ids, batch, hots, offsets, weights):
        k, n, dev = self.kernels, self.world, ids.device
        if ids.dtype == torch.int32 and self._row_off[-1] * self.world >= 2 ** 31:
            ids = ids.long()   # the composite id space (all tables, interleaved over the ranks) needs 64 bits
        n_feats = len(self._paths)
        nnz, n_bags = ids.numel(), batch * n_feats
        if offsets is None:
            comp = ids + self._composite_offsets(batch, hots, ids.dtype, dev)
            bag_of_pos, comb_of_bag = self._bag_tables(batch, hots, dev)
        else:  # ragged bags: CSR offsets over the feature-major bags
            lens = torch.diff(offsets)
            bag_of_pos = torch.repeat_interleave(torch.arange(n_bags, dtype=torch.int32, device=dev), lens)
            _, comb_of_bag = self._bag_tables(batch, (1,) * n_feats, dev)
            feat_off = torch.tensor([self._row_off[t] * self.world for t in self._table_of_feature], dtype=ids.dtype,
                                    device=dev)
            comp = ids + feat_off[(bag_of_pos // batch).long()]
        local_rows, perm, counts = k.bucketize(comp, n)
        w_eff = self._lookup_weights(weights, bag_of_pos, comb_of_bag, n_bags)
        # segments: runs of one bag inside a bucket (the bucketise is stable, so bags ascend in a bucket).
        # Everything up to the size exchange has a data-independent shape (length nnz, valid in the first
        # n_seg entries), so the host waits for the device exactly once per lookup: for the sizes.
        order = perm.long()
        bag_b = bag_of_pos[order]
        ends = torch.cumsum(counts, 0)
        starts = ends - counts
        head = torch.ones(nnz + 1, dtype=torch.bool, device=dev)
        if nnz > 1:
            head[1:nnz] = bag_b[1:] != bag_b[:-1]
        head[starts] = True          # an empty bucket's start is the next bucket's (or the spare slot nnz)
        heads_before = torch.zeros(nnz + 1, dtype=torch.int64, device=dev)
        heads_before[1:] = torch.cumsum(head[:nnz], 0)                               # segments in front of position p
        seg_counts = heads_before[ends] - heads_before[starts]                       # per destination bucket
        # slot s < n_seg: first lookup of segment s; slots n_seg..nnz keep the sentinel nnz; slot nnz + 1
        # swallows the writes of the non-head positions
        pos = torch.arange(nnz, dtype=torch.int64, device=dev)
        slot = torch.where(head[:nnz], heads_before[:nnz], torch.full_like(pos, nnz + 1))
        first = torch.full((nnz + 2,), nnz, dtype=torch.int64, device=dev)
        first[slot] = pos
        seg_len_all = (first[1:nnz + 1] - first[:nnz]).to(torch.int32)
        seg_bag_all = bag_b[first[:nnz].clamp_(max=max(nnz - 1, 0))]
        # sizes: every rank learns how many lookups / segments it receives (one tiny all-to-all + host sync)
        mine = torch.stack([counts.to(torch.int64), seg_counts], dim=1).contiguous()   # [n, 2]
        if n > 1:
            theirs = self._a2a(mine, [1] * n, [1] * n)
        else:
            theirs = mine.clone()
        sizes = torch.stack([mine, theirs]).cpu()   # ONE device-to-host copy / sync for all four lists
        send_counts, send_segs = sizes[0, :, 0].tolist(), sizes[0, :, 1].tolist()
        recv_counts, recv_segs = sizes[1, :, 0].tolist(), sizes[1, :, 1].tolist()
        n_seg = sum(send_segs)
        seg_bag, seg_len = seg_bag_all[:n_seg], seg_len_all[:n_seg]
        # to the owners: rows, segment lengths, weights (bucket order)
        recv_rows = self._a2a(local_rows, send_counts, recv_counts)
        recv_len = self._a2a(seg_len, send_segs, recv_segs)
        recv_w = None if w_eff is None else self._a2a(w_eff[order], send_counts, recv_counts)
        recv_off = torch.zeros(recv_len.numel() + 1, dtype=torch.int32, device=dev)
        recv_off[1:] = torch.cumsum(recv_len, 0)
        partial = k.pool_segments(self.shard.data, recv_rows, recv_off, recv_w, self.compute_dtype)
        back = self._a2a(partial, recv_segs, send_segs)                           # home side, segment order
        # every bag sums its partials: segments sorted by bag -> feature-major CSR over segment ids
        seg_sorted = torch.argsort(seg_bag, stable=True).to(torch.int32)
        bag_off = torch.zeros(n_bags + 1, dtype=torch.int32, device=dev)
        per_bag = torch.zeros(n_bags, dtype=torch.int32, device=dev).scatter_add_(
            0, seg_bag.long(), torch.ones_like(seg_bag, dtype=torch.int32))
        bag_off[1:] = torch.cumsum(per_bag, 0)
        feats = [(0, "sum", i * self.dim) for i in range(len(self._combiners))]
        lead = self.slab_lead_cols
        slab = torch.empty((batch, lead + len(feats) * self.dim), dtype=back.dtype, device=dev)
        k.pool(back, seg_sorted, feats, batch, bag_off, self.compute_dtype, out=slab[:, lead:])
        saved = dict(batch=batch, seg_bag=seg_bag, send_segs=send_segs, recv_segs=recv_segs, recv_rows=recv_rows,
                     recv_off=recv_off, recv_w=recv_w, out_meta=(slab.dtype, slab.device))
        return slab, saved

    def _backward_impl(self, g, s):
        k, n_feats, batch = self.kernels, len(self._combiners), s["batch"]
        # d(partial of a segment) = the output gradient of its bag: rows of g viewed as [batch * n_feats, dim]
        g = g.contiguous()
        seg_bag = s["seg_bag"]
        rows = ((seg_bag % batch) * n_feats + seg_bag // batch).to(torch.int32)
        dpart = k.gather_rows(g.view(batch * n_feats, self.dim), rows)
        dseg = self._a2a(dpart, s["send_segs"], s["recv_segs"])                   # to the owners
        lr = self._fused.lr_at(self._step)
        self._step += 1
        k.apply_segments(self.shard.data, self._slot, s["recv_rows"], s["recv_off"], s["recv_w"], dseg, lr,
                         self._opt_kind, self._fused.hyper(self._step))
