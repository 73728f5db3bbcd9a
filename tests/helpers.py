"""Shared helpers for the parity tests (oracle side lives here, not in the product)."""

import numpy as np
import torch

from oracle import krs_oracle as ko


def to_np(t: torch.Tensor) -> np.ndarray:
    """torch -> numpy; bf16 travels as uint16 bit patterns (oracle convention)."""
    t = t.detach().cpu().contiguous()
    if t.dtype == torch.bfloat16:
        return t.view(torch.int16).numpy().view(np.uint16)
    return t.numpy()


def to_f32(a: np.ndarray) -> np.ndarray:
    return ko.bf16_bits_to_f32(a) if a.dtype == np.uint16 else a.astype(np.float32)


def make_bags(rng, n_feats, batch, vocabs, max_hot, csr, with_empty=True, id_dtype=np.int32):
    """Random feature-major bags.  Returns dict(ids, offsets|None, hots|None, weights, nnz)."""
    if csr:
        lens = rng.integers(0 if with_empty else 1, max_hot + 1, size=(n_feats, batch))
        offsets = np.concatenate([[0], np.cumsum(lens.reshape(-1))]).astype(np.int32)
        ids = np.concatenate([rng.integers(0, vocabs[f], size=int(lens[f].sum())) for f in range(n_feats)]
                             + [np.zeros(0, np.int64)]).astype(id_dtype)
        hots = None
    else:
        hots = [int(h) for h in rng.integers(1, max_hot + 1, size=n_feats)]
        ids = np.concatenate([rng.integers(0, vocabs[f], size=batch * hots[f]) for f in range(n_feats)]
                             ).astype(id_dtype)
        offsets = None
    w = rng.uniform(0.0, 1.0, size=ids.shape[0]).astype(np.float32)
    return dict(ids=ids, offsets=offsets, hots=hots, weights=w, nnz=int(ids.shape[0]))


def oracle_embed_fwd(tables_np, feat_specs, bags, batch, dim, out_cols, out_np_dtype, use_w=True):
    """feat_specs: [(table_idx, combiner, out_col)].  Returns (out, bag_scale, flags)."""
    tabs = ko.make_tables(tables_np)
    feats = ko.make_features([t for t, _, _ in feat_specs], [c for _, c, _ in feat_specs],
                             [col for _, _, col in feat_specs], hots=bags["hots"], batch=batch)
    out = np.zeros((batch, out_cols), dtype=out_np_dtype)
    scale = np.zeros(len(feat_specs) * batch, np.float32)
    flags = ko.embed_bag_fwd_raw(tabs, ko.fdtype(tables_np[0]), feats, bags["ids"], bags["offsets"],
                                 bags["weights"] if use_w else None, batch, dim, out, scale)
    return out, scale, flags
