"""EmbedReduce on MI355X: drop-in for keras_rs.layers.EmbedReduce
(keras_rs/src/layers/embedding/embed_reduce.py:133-309): an embedding lookup
followed by a weighted sum / mean / sqrtn reduction over axis -2, in one HIP
kernel (K1), with the reference's dense-gradient autodiff (K2)."""

from __future__ import annotations

from typing import Any

import numpy as np
import torch

from keras_rs_amd.autograd import EmbedBagFn
from keras_rs_amd.embedding_ops import FusedBags
from keras_rs_amd.layers import base

SUPPORTED_COMBINERS = ("mean", "sum", "sqrtn")


class Ragged:
    """Rows of variable length as CSR: `values` [nnz], `row_offsets` [rows + 1].
    Stands in for tf.RaggedTensor / sparse inputs (embed_reduce_test.py:51-80)."""

    def __init__(self, values, row_offsets):
        self.values = values
        self.row_offsets = row_offsets

    @classmethod
    def from_rows(cls, rows, dtype=np.int32):
        """Rows (arrays or lists of varying length; the ragged input form of base_distributed_embedding.py:31-92)
        -> CSR without a Python-level loop body per row: the lengths come from one C-level map, the values from
        one concatenate (array rows) or one chained fromiter (list rows) -- 1.7 M rows per ml_perf batch."""
        import itertools

        n = len(rows)
        lens = np.fromiter(map(len, rows), dtype=np.int64, count=n)
        total = int(lens.sum())
        if n and all(isinstance(r, np.ndarray) for r in (rows[0], rows[-1])):
            vals = np.concatenate(rows).astype(dtype, copy=False) if total else np.zeros(0, dtype)
        else:
            vals = np.fromiter(itertools.chain.from_iterable(rows), dtype=dtype, count=total)
        off = np.zeros(n + 1, np.int64)
        np.cumsum(lens, out=off[1:])
        return cls(vals, off.astype(np.int32 if total < 2 ** 31 else np.int64))

    @property
    def shape(self):
        return (len(self.row_offsets) - 1, None)


def as_index_tensor(x, device) -> torch.Tensor:
    t = torch.as_tensor(np.asarray(x) if not isinstance(x, torch.Tensor) else x)
    if t.dtype not in (torch.int32, torch.int64):
        t = t.to(torch.int32)  # keras.layers.Embedding.call casts other dtypes to int32
    return t.to(device)


def check_shapes_compatible(a, b) -> bool:
    """keras_rs/src/utils/keras_utils.py:54-64: equal rank, every known pair equal."""
    if len(a) != len(b):
        return False
    return all(x is None or y is None or x == y for x, y in zip(a, b))


class EmbedReduce(base.Layer):
    """Args (embed_reduce.py:133-160): input_dim, output_dim, embeddings_initializer,
    embeddings_regularizer, embeddings_constraint, mask_zero, weights (initial table),
    combiner in {mean, sum, sqrtn}."""

    def __init__(self, input_dim: int, output_dim: int, embeddings_initializer="uniform",
                 embeddings_regularizer=None, embeddings_constraint=None, mask_zero: bool = False,
                 weights=None, combiner: str = "mean", **kwargs: Any):
        super().__init__(**kwargs)
        if combiner not in SUPPORTED_COMBINERS:  # embed_reduce.py:155-159
            raise ValueError(f"Invalid `combiner`: '{combiner}', use one of {', '.join(SUPPORTED_COMBINERS)}.")
        self.input_dim = int(input_dim)
        self.output_dim = int(output_dim)
        self.embeddings_initializer = base.get_initializer(embeddings_initializer)
        # keras.layers.Embedding under the reference's EmbedReduce (embed_reduce.py:138-150) hands both to its variable:
        # the regulariser's penalty appears in `layer.losses`; the constraint is a projection the optimizer applies after
        # its update -- keras_rs_amd.optim.Adagrad.step() does, other training loops call `layer.apply_constraints()`
        self.embeddings_regularizer = base.get_regularizer(embeddings_regularizer)
        self.embeddings_constraint = base.get_constraint(embeddings_constraint)
        self.mask_zero = mask_zero
        self.combiner = combiner
        self._initial_weights = weights
        self.register_parameter("embeddings", None)
        self._bags: dict = {}

    def build(self, *_) -> None:
        if self.embeddings is None:
            self.embeddings = self.add_weight((self.input_dim, self.output_dim), self.embeddings_initializer,
                                              "embeddings", regularizer=self.embeddings_regularizer,
                                              constraint=self.embeddings_constraint)
            if self._initial_weights is not None:
                with torch.no_grad():
                    self.embeddings.copy_(torch.as_tensor(np.asarray(self._initial_weights)))
        self.built = True

    def _input_shapes(self, args):
        return (getattr(args[0], "shape", None),)

    def _fused(self, combiner: str) -> FusedBags:
        fb = self._bags.get(combiner)
        if fb is None or fb.tables[0] is not self.embeddings:
            fb = FusedBags([self.embeddings], [(0, combiner, 0)])
            self._bags[combiner] = fb
        return fb

    def call(self, inputs, weights=None) -> torch.Tensor:
        dev = self.embeddings.device
        if isinstance(inputs, Ragged):
            ids = as_index_tensor(inputs.values, dev).reshape(-1)
            offsets = torch.as_tensor(np.asarray(inputs.row_offsets) if not isinstance(inputs.row_offsets, torch.Tensor)
                                      else inputs.row_offsets).to(torch.int32).to(dev)
            batch = offsets.numel() - 1
            w = None
            if weights is not None:
                wv = weights.values if isinstance(weights, Ragged) else weights
                w = torch.as_tensor(np.asarray(wv) if not isinstance(wv, torch.Tensor) else wv).to(dev)
                if w.numel() != ids.numel():
                    raise ValueError(f"The shape of `weights`: {tuple(w.shape)} is not compatible with the ragged "
                                     f"`inputs` ({ids.numel()} values).")
            out_dtype = self._out_dtype(w)
            return EmbedBagFn.apply(self._fused(self.combiner), ids, batch, None, offsets,
                                    None if w is None else w.float().reshape(-1), out_dtype, True,
                                    self.embeddings)[0]

        ids = as_index_tensor(inputs, dev)
        if ids.dim() not in (1, 2):
            raise ValueError(f"EmbedReduce inputs must be rank 1 or 2, got shape {tuple(ids.shape)}")
        unreduced = tuple(ids.shape) + (self.output_dim,)
        w = None
        if weights is not None:
            w = torch.as_tensor(np.asarray(weights) if not isinstance(weights, torch.Tensor) else weights).to(dev)
            if w.dim() > len(unreduced) or not check_shapes_compatible(unreduced[: w.dim()], tuple(w.shape)):
                raise ValueError(f"The shape of `weights`: {tuple(w.shape)} is not compatible with the shape of "
                                 f"`inputs` after embedding: {unreduced}.")  # embed_reduce.py:182-190
        out_dtype = self._out_dtype(w)
        combiner = self.combiner
        if ids.dim() == 1:
            # no reduction; weights only survive for "sum" (embed_reduce.py:224)
            if combiner != "sum":
                w = None
            combiner, hot = "sum", 1
        else:
            hot = ids.shape[1]
        batch = ids.shape[0]
        if w is not None and w.dim() > ids.dim():
            return self._call_per_dimension_weights(ids, w, combiner, out_dtype)
        if w is not None and w.dim() < ids.dim():
            # [B] weights with [B, L] ids.  The reference appends axes at the END (embed_reduce.py:244-248):
            # one weight per ROW, and the divisor sums the expanded [B, 1, 1] weights over an axis of length
            # one -- mean divides by w_b (not L * w_b), sqrtn by |w_b|, both divide_no_nan.  In per-lookup
            # weights of a plain sum: sum -> w_b, mean -> [w_b != 0], sqrtn -> sign(w_b).
            wb = w.float().reshape(-1, 1)
            wb = wb if combiner == "sum" else ((wb != 0).float() if combiner == "mean" else torch.sign(wb))
            w, combiner = wb.expand(ids.shape), "sum"
        if w is not None:
            w = w.float().contiguous().reshape(-1)
        return EmbedBagFn.apply(self._fused(combiner), ids.contiguous().reshape(-1), batch, (hot,), None, w,
                                out_dtype, True, self.embeddings)[0]

    def _call_per_dimension_weights(self, ids, w, combiner, out_dtype):
        """Weights of the rank of the embedded inputs ([B, D] with 1-D ids, [B, L, D] with 2-D ids: one weight
        per embedding COLUMN, embed_reduce.py:181-190 accepts them).  The bag kernel carries one weight per
        lookup, so this rare form gathers the rows with it (L = 1 bags) and applies embed_reduce.py:253-274 as
        device tensor ops on the gathered block."""
        n = ids.numel()
        rows = EmbedBagFn.apply(self._fused("sum"), ids.contiguous().reshape(-1), n, (1,), None, None,
                                torch.float32, True, self.embeddings)[0]
        x = rows.reshape(tuple(ids.shape) + (self.output_dim,))
        wf = w.float().expand(x.shape)
        if ids.dim() == 1:                       # no reduction; weights only survive for "sum"
            return (x * wf if self.combiner == "sum" else x).to(out_dtype)
        s = (x * wf).sum(dim=-2)
        if combiner == "mean":
            d = wf.sum(dim=-2)
        elif combiner == "sqrtn":
            d = wf.square().sum(dim=-2).sqrt()
        else:
            return s.to(out_dtype)
        return torch.where(d != 0, s / torch.where(d != 0, d, torch.ones_like(d)), torch.zeros_like(s)).to(out_dtype)

    def _out_dtype(self, w):
        cd = self.compute_dtype
        # keras.backend.result_type(x.dtype, weights.dtype): bf16 x fp32 weights -> fp32
        if w is not None and w.dtype == torch.float32 and cd == torch.bfloat16:
            return torch.float32
        return cd

    def compute_output_shape(self, input_shape, weights_shape=None):
        if len(input_shape) <= 1:
            return (*input_shape, self.output_dim)
        return (*input_shape[0:-1], self.output_dim)

    def get_config(self) -> dict:
        config = super().get_config()
        config.update({
            "input_dim": self.input_dim,
            "output_dim": self.output_dim,
            "embeddings_initializer": self.embeddings_initializer.serialize(),
            "embeddings_regularizer": base.serialize_regularizer(self.embeddings_regularizer),
            "embeddings_constraint": base.serialize_constraint(self.embeddings_constraint),
            "mask_zero": self.mask_zero,
            "combiner": self.combiner,
        })
        return config


class Embedding(EmbedReduce):
    """keras.layers.Embedding look-alike (plain row gather, no reduction): ids of any
    rank -> ids.shape + (output_dim,).  Used by the examples' small tables
    (examples/ml_perf/model.py:193-201, README.md:56-60); runs on the same K1 kernel
    with every id as its own one-element bag."""

    def __init__(self, input_dim: int, output_dim: int, embeddings_initializer="uniform", **kwargs):
        kwargs.pop("combiner", None)
        super().__init__(input_dim, output_dim, embeddings_initializer=embeddings_initializer, combiner="sum",
                         **kwargs)

    def call(self, inputs, weights=None) -> torch.Tensor:
        if weights is not None:
            raise ValueError("Embedding takes no weights; use EmbedReduce")
        ids = as_index_tensor(inputs, self.embeddings.device)
        flat = super().call(ids.reshape(-1))
        return flat.reshape(*ids.shape, self.output_dim)

    def compute_output_shape(self, input_shape, weights_shape=None):
        return (*input_shape, self.output_dim)
