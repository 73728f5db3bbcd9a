"""numpy front-end of the CPU oracle (oracle/krs_oracle.c).  TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module, and only as the checker / the timed CPU baseline.  The product
package (keras_rs_amd) never imports it.

Arrays are numpy; bf16 tensors are carried as ``np.uint16`` bit patterns (numpy
has no bfloat16), fp32 as ``np.float32``.  Struct layouts mirror include/krs.h.
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libkrs_oracle.so")

F32, BF16 = 0, 1
I32, I64 = 0, 1
SUM, MEAN, SQRTN = 0, 1, 2
COMBINERS = {"sum": SUM, "mean": MEAN, "sqrtn": SQRTN}
ACTS = {None: 0, "linear": 0, "relu": 1, "sigmoid": 2, "tanh": 3}

TABLE_DT = np.dtype(
    [("weights", "<u8"), ("slot", "<u8"), ("row_base", "<i8"), ("vocab", "<i4"), ("lr", "<f4")]
)
FEATURE_DT = np.dtype(
    [("ids_base", "<i8"), ("table", "<i4"), ("hot", "<i4"), ("combiner", "<i4"), ("out_col", "<i4")]
)
assert TABLE_DT.itemsize == 32 and FEATURE_DT.itemsize == 24


class GemmEpilogue(C.Structure):
    _fields_ = [
        ("bias", C.c_void_p),
        ("act", C.c_int32),
        ("diag_scale", C.c_float),
        ("x0", C.c_void_p),
        ("x", C.c_void_p),
        ("ldx", C.c_int64),
        ("u_out", C.c_void_p),
        ("ldu", C.c_int64),
        ("r", C.c_void_p),
        ("ldr", C.c_int64),
        ("beta", C.c_float),
    ]


def build(force: bool = False) -> str:
    """Compile the C restatement with gcc (a few hundred ms)."""
    src = os.path.join(_HERE, "krs_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
    return _lib


def _p(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


def fdtype(a: np.ndarray) -> int:
    if a.dtype == np.float32:
        return F32
    if a.dtype == np.uint16:
        return BF16
    raise TypeError(f"float tensors must be float32 or uint16(bf16 bits), got {a.dtype}")


def itype(a: np.ndarray) -> int:
    if a.dtype == np.int32:
        return I32
    if a.dtype == np.int64:
        return I64
    raise TypeError(f"index tensors must be int32/int64, got {a.dtype}")


def f32_to_bf16_bits(a: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even fp32 -> bf16 bit pattern (same rule as the C file)."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = (u + 0x7FFF + ((u >> 16) & 1)) >> 16
    return r.astype(np.uint16)


def bf16_bits_to_f32(a: np.ndarray) -> np.ndarray:
    return (a.astype(np.uint32) << 16).view(np.float32)


def make_tables(weights, slots=None, lrs=None, row_bases=None) -> np.ndarray:
    """krs_table array for host arrays `weights` (list of [V, D] arrays)."""
    t = np.zeros(len(weights), dtype=TABLE_DT)
    base = 0
    for i, w in enumerate(weights):
        assert w.flags["C_CONTIGUOUS"]
        t[i]["weights"] = w.ctypes.data
        t[i]["slot"] = 0 if slots is None or slots[i] is None else slots[i].ctypes.data
        t[i]["row_base"] = base if row_bases is None else row_bases[i]
        t[i]["vocab"] = w.shape[0]
        t[i]["lr"] = 0.0 if lrs is None else lrs[i]
        base += w.shape[0]
    return t


def make_features(table_idx, combiners, out_cols, hots=None, batch=None) -> np.ndarray:
    """krs_feature array.  `hots` (ids per bag) given -> dense mode ids_base."""
    f = np.zeros(len(table_idx), dtype=FEATURE_DT)
    base = 0
    for i in range(len(table_idx)):
        f[i]["table"] = table_idx[i]
        c = combiners[i]
        f[i]["combiner"] = COMBINERS[c] if isinstance(c, str) else c
        f[i]["out_col"] = out_cols[i]
        if hots is not None:
            f[i]["hot"] = hots[i]
            f[i]["ids_base"] = base
            base += batch * hots[i]
    return f


def embed_bag_fwd_raw(tables, table_dtype, feats, ids, offsets, weights, batch, dim, out,
                      bag_scale=None):
    flag = np.zeros(1, dtype=np.int32)
    rc = lib().krs_oracle_embed_bag_fwd(
        _p(tables), _p(feats), C.c_int(len(feats)),
        _p(ids), C.c_int(itype(ids)),
        _p(offsets), C.c_int(itype(offsets) if offsets is not None else I32),
        _p(weights), C.c_int(batch), C.c_int(dim), C.c_int(table_dtype),
        _p(out), C.c_int(fdtype(out)), C.c_int64(out.strides[0] // out.itemsize),
        _p(bag_scale), _p(flag), None,
    )
    assert rc == 0, rc
    return int(flag[0])


def embed_bag_bwd_dense(grad_tables, feats, ids, offsets, weights, bag_scale, grad, batch, dim):
    rc = lib().krs_oracle_embed_bag_bwd_dense(
        _p(grad_tables), C.c_int(len(grad_tables)), _p(feats), C.c_int(len(feats)),
        _p(ids), C.c_int(itype(ids)),
        _p(offsets), C.c_int(itype(offsets) if offsets is not None else I32),
        _p(weights), _p(bag_scale),
        _p(grad), C.c_int(fdtype(grad)), C.c_int64(grad.strides[0] // grad.itemsize),
        C.c_int(batch), C.c_int(dim),
    )
    assert rc == 0, rc


def apply_optimizer(table, acc, grad, touched, lr, kind, hyper=None):
    """kind: 'sgd' | 'adagrad' | 'adam' | 'ftrl' | 'adagrad_rowwise'.  In place on `table` (and `acc`: [V, D] for
    adagrad, [V] for adagrad_rowwise, [2, V, D] for adam (m, v) / ftrl (accumulator, linear)).  hyper: adam (beta_1,
    beta_2, epsilon, bias_correction); ftrl (learning_rate_power, l1, l2, beta); adagrad_rowwise (columns per
    16-byte piece of the gradient dtype: 8 for bf16, 4 for fp32 -- the order of its sum of squares)."""
    h = np.zeros(4, np.float32) if hyper is None else np.asarray(hyper, np.float32)
    rc = lib().krs_oracle_apply_optimizer2(
        _p(table), C.c_int(fdtype(table)), _p(acc), _p(grad), _p(touched),
        C.c_int64(table.shape[0]), C.c_int(table.shape[1]), C.c_float(lr),
        C.c_int({"sgd": 0, "adagrad": 1, "adam": 2, "ftrl": 3, "adagrad_rowwise": 4}[kind]), _p(h),
    )
    assert rc == 0, rc


def gemm(a, b, m, n, k, a_is_km=False, b_is_nk=False, out_dtype=None, bias=None, act=None,
         diag_scale=0.0, x0=None, x=None, want_u=False, r=None, beta=1.0):
    """C = epilogue(A @ B); see include/krs.h krs_gemm.  Returns (c, u_or_None)."""
    in_dt = fdtype(a)
    odt = in_dt if out_dtype is None else out_dtype
    npdt = np.float32 if odt == F32 else np.uint16
    c = np.zeros((m, n), dtype=npdt)
    u = np.zeros((m, n), dtype=npdt) if want_u else None
    ep = GemmEpilogue()
    ep.bias = None if bias is None else bias.ctypes.data
    ep.act = ACTS[act] if not isinstance(act, int) else act
    ep.diag_scale = diag_scale
    if x0 is not None:
        assert x is not None and x0.strides[0] == x.strides[0]
        ep.x0, ep.x, ep.ldx = x0.ctypes.data, x.ctypes.data, x.strides[0] // x.itemsize
    if u is not None:
        ep.u_out, ep.ldu = u.ctypes.data, n
    if r is not None:
        ep.r, ep.ldr, ep.beta = r.ctypes.data, r.strides[0] // r.itemsize, beta
    rc = lib().krs_oracle_gemm(
        _p(a), C.c_int64(a.strides[0] // a.itemsize), C.c_int(int(a_is_km)),
        _p(b), C.c_int64(b.strides[0] // b.itemsize), C.c_int(int(b_is_nk)),
        _p(c), C.c_int64(n), C.c_int64(m), C.c_int64(n), C.c_int64(k),
        C.c_int(in_dt), C.c_int(odt), C.byref(ep),
    )
    assert rc == 0, rc
    return c, u


def cross_epilogue_fwd(u, x0, x, diag_scale=0.0):
    y = np.zeros_like(x)
    m, n = x.shape
    rc = lib().krs_oracle_cross_epilogue_fwd(
        _p(u), _p(x0), _p(x), _p(y), C.c_int64(m), C.c_int64(n), C.c_int64(n),
        C.c_float(diag_scale), C.c_int(fdtype(x)))
    assert rc == 0, rc
    return y


def cross_epilogue_bwd(g, u, x0, x, diag_scale=0.0, dx0_init=None, act=None, fold_direct=False):
    m, n = x.shape
    du = np.zeros_like(x)
    dx0 = np.zeros_like(x) if dx0_init is None else dx0_init.copy()
    dxd = dx0 if fold_direct else np.zeros_like(x)
    dbias = np.zeros(n, dtype=np.float32)
    rc = lib().krs_oracle_cross_epilogue_bwd(
        _p(g), _p(u), _p(x0), _p(x), _p(du), _p(dx0), C.c_int(int(dx0_init is not None)),
        _p(dxd), _p(dbias), C.c_int64(m), C.c_int64(n), C.c_int64(n), C.c_float(diag_scale),
        C.c_int(ACTS[act] if not isinstance(act, int) else act), C.c_int(fdtype(x)))
    assert rc == 0, rc
    return du, dx0, dxd, dbias


def colsum(a):
    m, n = a.shape
    out = np.zeros(n, dtype=np.float32)
    rc = lib().krs_oracle_colsum(_p(a), C.c_int64(a.strides[0] // a.itemsize), C.c_int64(m),
                                 C.c_int64(n), C.c_int(fdtype(a)), _p(out))
    assert rc == 0, rc
    return out


def _ptr_array(arrs):
    return (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])


def _ld_array(arrs):
    return (C.c_int64 * len(arrs))(*[a.strides[0] // a.itemsize for a in arrs])


def dot_interaction_out_cols(n_feats, self_interaction, skip_gather):
    if skip_gather:
        return n_feats * n_feats
    return n_feats * (n_feats + 1) // 2 if self_interaction else n_feats * (n_feats - 1) // 2


def dot_interaction_fwd(feats, self_interaction=False, skip_gather=False):
    batch, dim = feats[0].shape
    cols = dot_interaction_out_cols(len(feats), self_interaction, skip_gather)
    out = np.zeros((batch, cols), dtype=feats[0].dtype)
    rc = lib().krs_oracle_dot_interaction_fwd(
        _ptr_array(feats), _ld_array(feats), C.c_int(len(feats)), C.c_int64(batch), C.c_int(dim),
        C.c_int(fdtype(feats[0])), C.c_int(int(self_interaction)), C.c_int(int(skip_gather)),
        _p(out), C.c_int64(cols))
    assert rc == 0, rc
    return out


def dot_interaction_bwd(feats, grad_out, self_interaction=False, skip_gather=False, existing=None, accumulate_mask=0):
    """existing / accumulate_mask: krs_dot_interaction_bwd_accumulate -- `existing[f]` (copied) is the stored
    gradient of feature f, added to when bit f of the mask is set and overwritten otherwise."""
    batch, dim = feats[0].shape
    grads = [np.zeros_like(f) for f in feats] if existing is None else [np.array(e, copy=True, order="C") for e in existing]
    rc = lib().krs_oracle_dot_interaction_bwd_accumulate(
        _ptr_array(feats), _ld_array(feats), C.c_int(len(feats)), C.c_int64(batch), C.c_int(dim),
        C.c_int(fdtype(feats[0])), C.c_int(int(self_interaction)), C.c_int(int(skip_gather)),
        _p(grad_out), C.c_int64(grad_out.strides[0] // grad_out.itemsize),
        _ptr_array(grads), _ld_array(grads), C.c_uint64(int(accumulate_mask)))
    assert rc == 0, rc
    return grads


def dense_act_bwd(g, y, act):
    """include/krs.h krs_dense_act_bwd (autodiff of keras.layers.Dense's activation + bias, derivative from the saved
    output): g, y fp32 arrays or bf16 bit patterns; returns (dz in the input format, dbias fp32 = column sums of the
    unrounded products accumulated in float64)."""
    bf = g.dtype == np.uint16
    gf = bf16_bits_to_f32(g) if bf else np.asarray(g, np.float32)
    if act in (None, "none", 0):
        d = np.ones_like(gf)
    else:
        yf = bf16_bits_to_f32(y) if bf else np.asarray(y, np.float32)
        d = {"relu": (yf > 0).astype(np.float32), "sigmoid": yf * (np.float32(1) - yf),
             "tanh": np.float32(1) - yf * yf}[act]
    dz = (gf * d).astype(np.float32)
    return (f32_to_bf16_bits(dz) if bf else dz), dz.astype(np.float64).sum(0).astype(np.float32)


def bce_fwd_bwd(pred, labels, epsilon=1e-7, grad_scale=1.0):
    """include/krs.h krs_bce_fwd_bwd: keras.losses.BinaryCrossentropy() as examples/ml_perf/main.py:201-210 compiles it
    (probabilities in, mean reduction): p = clip(pred, eps, 1 - eps) in fp32, loss = mean(-(y log p + (1-y) log(1-p)))
    summed in float64, dpred = grad_scale / n * ((1-y)/(1-p) - y/p) where the clip is inactive, else 0.
    pred: fp32 array or bf16 bit patterns; returns (loss float32, dpred in the input format)."""
    bf = pred.dtype == np.uint16
    x = (bf16_bits_to_f32(pred) if bf else np.asarray(pred, np.float32)).reshape(-1)
    y = np.asarray(labels, np.float32).reshape(-1)
    eps, hi = np.float32(epsilon), np.float32(1.0) - np.float32(epsilon)
    p = np.minimum(np.maximum(x, eps), hi)
    one = np.float32(1.0)
    li = -(y * np.log(p) + (one - y) * np.log(one - p))
    loss = np.float32(li.astype(np.float64).sum() / len(x))
    inside = (x >= eps) & (x <= hi)
    g = np.where(inside, np.float32(grad_scale) * (one / np.float32(len(x))) * ((one - y) / (one - p) - y / p),
                 np.float32(0)).astype(np.float32)
    return loss, (f32_to_bf16_bits(g) if bf else g)


def dense_adagrad(p, g, acc, lr, eps):
    """include/krs.h krs_dense_adagrad (keras / torch Adagrad on a dense fp32 weight, epsilon outside the root):
    returns (new p, new acc); fp32 arithmetic with the fused multiply-add of acc + g*g done in float64 and
    rounded once, as fmaf does."""
    p, g, acc = (np.asarray(t, np.float32) for t in (p, g, acc))
    acc2 = (acc.astype(np.float64) + g.astype(np.float64) * g.astype(np.float64)).astype(np.float32)
    upd = (np.float32(lr) * g) / (np.sqrt(acc2) + np.float32(eps))
    return (p - upd.astype(np.float32)).astype(np.float32), acc2


def cast_transpose(w, to_bf16):
    """include/krs.h krs_cast_transpose: (cast(w), cast(w)^T); w fp32 array or bf16 bit pattern (uint16); the
    cast is ops.cast under a mixed-precision policy (feature_cross.py:182-194): round-to-nearest-even."""
    wf = bf16_bits_to_f32(w) if w.dtype == np.uint16 else np.asarray(w, np.float32)
    out = f32_to_bf16_bits(wf) if to_bf16 else wf
    return np.ascontiguousarray(out), np.ascontiguousarray(out.T)


def mod_bucketize(ids, n_shards):
    nnz = ids.shape[0]
    local = np.zeros_like(ids)
    perm = np.zeros(nnz, dtype=np.int32)
    counts = np.zeros(n_shards, dtype=np.int64)
    rc = lib().krs_oracle_mod_bucketize(_p(ids), C.c_int(itype(ids)), C.c_int64(nnz),
                                        C.c_int(n_shards), _p(local), _p(perm), _p(counts))
    assert rc == 0, rc
    return local, perm, counts


# --------------------------------------------------------------------------- #
# Layer-level restatements (what one reference layer call returns), built from
# the C functions above.  These are what the parity tests compare against.
# --------------------------------------------------------------------------- #


def embed_reduce(table, ids, weights=None, combiner="mean", out_dtype=None):
    """EmbedReduce.call (embed_reduce.py:162-274) for dense ids of rank 1 or 2.

    Rank-1 ids: no reduction; given weights are honoured only for 'sum'
    (embed_reduce.py:224: discarded when unreduced_rank <= 2 and combiner != sum).
    """
    ids = np.ascontiguousarray(ids)
    if ids.ndim == 1:
        hot = 1
        if combiner != "sum":
            weights = None
        comb = "sum"
    elif ids.ndim == 2:
        hot = ids.shape[1]
        comb = combiner
    else:
        raise ValueError("EmbedReduce inputs must be rank 1 or 2")
    batch = ids.shape[0]
    dim = table.shape[1]
    tables = make_tables([table])
    feats = make_features([0], [comb], [0], hots=[hot], batch=batch)
    odt = fdtype(table) if out_dtype is None else out_dtype
    out = np.zeros((batch, dim), dtype=np.float32 if odt == F32 else np.uint16)
    w = None if weights is None else np.ascontiguousarray(weights, dtype=np.float32).reshape(-1)
    flags = embed_bag_fwd_raw(tables, fdtype(table), feats, ids.reshape(-1), None, w, batch, dim, out)
    if flags:
        raise IndexError("embedding id out of range")
    return out


def embed_reduce_csr(table, ids, offsets, weights=None, combiner="mean"):
    """EmbedReduce on ragged / sparse rows given as CSR (embed_reduce_test.py:51-80)."""
    batch = offsets.shape[0] - 1
    dim = table.shape[1]
    tables = make_tables([table])
    feats = make_features([0], [combiner], [0])
    out = np.zeros((batch, dim), dtype=table.dtype)
    w = None if weights is None else np.ascontiguousarray(weights, dtype=np.float32)
    flags = embed_bag_fwd_raw(tables, fdtype(table), feats, np.ascontiguousarray(ids),
                              np.ascontiguousarray(offsets), w, batch, dim, out)
    if flags:
        raise IndexError("embedding id out of range")
    return out


def feature_cross(x0, x, kernel, bias=None, down_kernel=None, diag_scale=0.0, act=None):
    """FeatureCross.call (feature_cross.py:155-194).  kernels in keras layout [in, out]."""
    if x is None:
        x = x0
    if x0.shape != x.shape:
        raise ValueError("`x0` and `x` should have the same shape")
    lead = x.shape[:-1]
    d = x.shape[-1]
    x2 = np.ascontiguousarray(x).reshape(-1, d)
    x02 = np.ascontiguousarray(x0).reshape(-1, d)
    m = x2.shape[0]
    h = x2
    if down_kernel is not None:
        p = down_kernel.shape[1]
        h, _ = gemm(x2, down_kernel, m, p, d)
    kk = h.shape[1]
    y, _ = gemm(h, kernel, m, d, kk, bias=bias, act=act, diag_scale=diag_scale or 0.0, x0=x02, x=x2)
    return y.reshape(*lead, d)


# --------------------------------------------------------------------------- #
# K6: the id side of the row-sharded lookup (include/krs.h: krs_shard_*)
# --------------------------------------------------------------------------- #
SHARD_FEATURE_DT = np.dtype(
    [("ids_base", "<i8"), ("comp_off", "<i8"), ("hot", "<i4"), ("combiner", "<i4"), ("vocab", "<i4"),
     ("reserved", "<i4")]
)
assert SHARD_FEATURE_DT.itemsize == 32


def shard_route(feats, ids, offsets, weights, batch, n_shards, emit_weights):
    """Returns dict(packed, seg_bag, seg_grow, bag_seg, counts [3, n_shards], flags); arrays sized as the device
    call sizes them (first n_seg / sum(counts[2]) entries valid)."""
    nnz = int(ids.shape[0])
    n_feats = len(feats)
    packed = np.zeros(max(nnz * (2 + int(bool(emit_weights))), 1), np.int32)
    seg_bag = np.zeros(max(nnz, 1), np.int32)
    seg_grow = np.zeros(max(nnz, 1), np.int32)
    bag_seg = np.zeros((max(batch * n_feats, 1), n_shards), np.int32)
    counts = np.zeros(3 * n_shards, np.int64)
    flag = np.zeros(1, np.int32)
    w = None if weights is None else np.ascontiguousarray(weights, np.float32)
    rc = lib().krs_oracle_shard_route(
        _p(feats), C.c_int(n_feats), _p(ids), C.c_int(itype(ids)),
        _p(offsets), C.c_int(itype(offsets) if offsets is not None else I32), _p(w), C.c_int64(nnz),
        C.c_int(batch), C.c_int(n_shards), C.c_int(int(bool(emit_weights))), _p(packed), _p(seg_bag), _p(seg_grow),
        _p(bag_seg), _p(counts), _p(flag))
    assert rc == 0, rc
    return dict(packed=packed, seg_bag=seg_bag, seg_grow=seg_grow, bag_seg=bag_seg,
                counts=counts.reshape(3, n_shards), flags=int(flag[0]))


def shard_unpack(packed, lookups, segments, weighted):
    lookups = np.ascontiguousarray(lookups, np.int64)
    segments = np.ascontiguousarray(segments, np.int64)
    n_cnt, n_seg = int(lookups.sum()), int(segments.sum())
    rows = np.zeros(max(n_cnt, 1), np.int32)
    w = np.zeros(max(n_cnt, 1), np.float32)
    off = np.zeros(n_seg + 1, np.int32)
    rc = lib().krs_oracle_shard_unpack(_p(np.ascontiguousarray(packed, np.int32)), C.c_int(len(lookups)), _p(lookups),
                                       _p(segments), C.c_int(int(bool(weighted))), _p(rows), _p(w), _p(off))
    assert rc == 0, rc
    return rows[:n_cnt], (w[:n_cnt] if weighted else None), off


STATIC_HEADER = 4
FLAG_CAPACITY_OVERFLOW = 4


def shard_static_block_words(cap_l, cap_s, emit_weights):
    return STATIC_HEADER + cap_l * (1 + int(bool(emit_weights))) + cap_s


def shard_route_static(feats, ids, offsets, weights, batch, n_shards, emit_weights, cap_l, cap_s):
    """Static-capacity form of shard_route (include/krs.h: krs_shard_route_static), restated on top of the exact
    oracle: the reference's static buffers + id dropping (distributed_embedding_config.py:54-61,
    jax/embedding_utils.py:187-197).  Owner d's block = [kept lookups, kept segments, need_l, need_s | rows[cap_l] |
    weights[cap_l] | segment lengths[cap_s]]; owner d keeps its first cap_s segments and of those the first cap_l
    lookups.  Returns dict(packed [n_shards, W], seg_grow [n_shards*cap_s], bag_seg (slots), counts, flags)."""
    r = shard_route(feats, ids, offsets, weights, batch, n_shards, emit_weights)
    ew = int(bool(emit_weights))
    n_feats = len(feats)
    W = shard_static_block_words(cap_l, cap_s, ew)
    packed = np.zeros((n_shards, W), np.int32)
    seg_grow = np.zeros(n_shards * cap_s, np.int32)
    bag_seg = np.full((max(batch * n_feats, 1), n_shards), -1, np.int32)
    cnt, segs, words = (r["counts"][k] for k in range(3))
    need_l, need_s = int(cnt.max(initial=0)), int(segs.max(initial=0))
    flags = r["flags"]
    base, seg0 = 0, 0
    for d in range(n_shards):
        c, g = int(cnt[d]), int(segs[d])
        blk = r["packed"][base: base + int(words[d])]
        rows, wts, lens = blk[:c], blk[c: c + c * ew], blk[c * (1 + ew):]
        keep_s = min(g, cap_s)
        keep_l = min(int(lens[:keep_s].sum()), cap_l)
        if keep_l < c or keep_s < g:
            flags |= FLAG_CAPACITY_OVERFLOW
        packed[d, :4] = (keep_l, keep_s, need_l, need_s)
        packed[d, 4: 4 + keep_l] = rows[:keep_l]
        if ew:
            packed[d, 4 + cap_l: 4 + cap_l + keep_l] = wts[:keep_l]
        ends = np.minimum(np.cumsum(lens[:keep_s]), keep_l)
        starts = np.minimum(np.concatenate([[0], np.cumsum(lens[:keep_s])[:-1]]), keep_l) if keep_s else np.zeros(0, np.int64)
        packed[d, 4 + cap_l * (1 + ew): 4 + cap_l * (1 + ew) + keep_s] = (ends - starts).astype(np.int32)
        for j in range(keep_s):
            bag = int(r["seg_bag"][seg0 + j])
            seg_grow[d * cap_s + j] = (bag % batch) * n_feats + bag // batch
            bag_seg[bag, d] = d * cap_s + j
        base += int(words[d])
        seg0 += g
    return dict(packed=packed, seg_grow=seg_grow, bag_seg=bag_seg, counts=r["counts"], flags=flags)


def shard_unpack_static(packed, cap_l, cap_s, weighted):
    """Owner side of the static form (krs_shard_unpack_static): packed [n_src, W] -> rows / w [n_src*cap_l] (compact,
    tail = -1 / 0), offsets [n_src*cap_s + 1] over the segment slots, stats [4]."""
    packed = np.ascontiguousarray(packed, np.int32)
    n_src = packed.shape[0]
    wt = int(bool(weighted))
    rows = np.full(n_src * cap_l, -1, np.int32)
    w = np.zeros(n_src * cap_l, np.float32)
    lens = np.zeros(n_src * cap_s + 1, np.int64)
    run = 0
    nl = ns = tot_s = 0
    for s in range(n_src):
        c = int(np.clip(packed[s, 0], 0, cap_l))
        g = int(np.clip(packed[s, 1], 0, cap_s))
        rows[run: run + c] = packed[s, 4: 4 + c]
        if wt:
            w[run: run + c] = packed[s, 4 + cap_l: 4 + cap_l + c].view(np.float32)
        lens[s * cap_s: s * cap_s + g] = packed[s, 4 + cap_l * (1 + wt): 4 + cap_l * (1 + wt) + g]
        run += c
        tot_s += g
        nl, ns = max(nl, int(packed[s, 2])), max(ns, int(packed[s, 3]))
    off = np.concatenate([[0], np.cumsum(lens[:-1])]).astype(np.int32)
    return rows, (w if wt else None), off, np.array([nl, ns, run, tot_s], np.int64)


def shard_combine(partials, bag_seg, batch, n_feats, dim, out=None):
    n_shards = bag_seg.shape[1]
    if out is None:
        out = np.zeros((batch, n_feats * dim), partials.dtype)
    part = np.ascontiguousarray(partials) if partials.shape[0] else np.zeros((1, dim), partials.dtype)
    rc = lib().krs_oracle_shard_combine(_p(part), _p(np.ascontiguousarray(bag_seg, np.int32)), C.c_int(batch),
                                        C.c_int(n_feats), C.c_int(n_shards), C.c_int(dim), C.c_int(fdtype(part)),
                                        _p(out), C.c_int64(out.strides[0] // out.itemsize))
    assert rc == 0, rc
    return out
