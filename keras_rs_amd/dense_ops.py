"""Host side of K3/K4/K5: thin torch wrappers over krs_gemm, the cross epilogue
kernels, krs_dot_interaction_{fwd,bwd} and krs_mod_bucketize.

Everything here marshals raw device pointers into the C ABI (include/krs.h);
torch only provides memory and the current stream.
"""

from __future__ import annotations

import ctypes as C
from typing import Sequence

import torch

from keras_rs_amd import _lib as L
from keras_rs_amd import probe

ACTS = {None: L.ACT_NONE, "linear": L.ACT_NONE, "relu": L.ACT_RELU, "sigmoid": L.ACT_SIGMOID,
        "tanh": L.ACT_TANH}


def _rowmajor(t: torch.Tensor, what: str) -> torch.Tensor:
    L.require_device(t, what)
    if t.dim() != 2:
        raise L.KrsError(f"{what}: expected a matrix, got shape {tuple(t.shape)}")
    # a row-broadcast view (strides (0, 1), e.g. the gradient of y.sum(0)) has no leading dimension a kernel can walk
    if t.stride(1) != 1 or (t.shape[0] > 1 and t.stride(0) < t.shape[1]):
        return t.contiguous()
    return t


def gemm(a: torch.Tensor, b: torch.Tensor, *, a_is_km: bool = False, b_is_nk: bool = False,
         out_dtype: torch.dtype | None = None, bias: torch.Tensor | None = None, act: int = L.ACT_NONE,
         diag_scale: float = 0.0, x0: torch.Tensor | None = None, x: torch.Tensor | None = None,
         want_u: bool = False, r: torch.Tensor | None = None, beta: float = 1.0,
         out: torch.Tensor | None = None):
    """C = epilogue(A @ B) (include/krs.h: krs_gemm).  Returns (C, u_or_None).

    a: [M,K] (or [K,M] when a_is_km);  b: [K,N] (or [N,K] when b_is_nk)."""
    a = _rowmajor(a, "gemm A")
    b = _rowmajor(b, "gemm B")
    if a.dtype != b.dtype:
        raise L.KrsError("gemm: A and B must share a dtype")
    m, k = (a.shape[1], a.shape[0]) if a_is_km else (a.shape[0], a.shape[1])
    kb, n = (b.shape[1], b.shape[0]) if b_is_nk else (b.shape[0], b.shape[1])
    if k != kb:
        raise L.KrsError(f"gemm: inner dimensions differ ({k} vs {kb})")
    odt = out_dtype or a.dtype
    c = out if out is not None else torch.empty((m, n), dtype=odt, device=a.device)
    ep = L.GemmEpilogue()
    keep = []
    if bias is not None:
        bias = bias.float().contiguous()
        keep.append(bias)
        ep.bias = bias.data_ptr()
    ep.act = act
    ep.diag_scale = float(diag_scale or 0.0)
    if x0 is not None:
        x0 = _rowmajor(x0, "gemm x0")
        x = _rowmajor(x, "gemm x")
        if x0.stride(0) != x.stride(0):
            x0, x = x0.contiguous(), x.contiguous()
        if x0.dtype != odt or x.dtype != odt:
            raise L.KrsError("gemm: x0/x must have the output dtype")
        keep += [x0, x]
        ep.x0, ep.x, ep.ldx = x0.data_ptr(), x.data_ptr(), x.stride(0)
    u = None
    if want_u:
        u = torch.empty((m, n), dtype=odt, device=a.device)
        ep.u_out, ep.ldu = u.data_ptr(), u.stride(0)
    if r is not None:
        r = _rowmajor(r, "gemm R")
        if r.dtype != odt:
            raise L.KrsError("gemm: R must have the output dtype")
        keep.append(r)
        ep.r, ep.ldr, ep.beta = r.data_ptr(), r.stride(0), float(beta)
    wsb = L.lib().krs_gemm_workspace_bytes(C.c_int64(m), C.c_int64(n), C.c_int64(k), C.c_int(int(a_is_km)))
    ws = torch.empty(int(wsb), dtype=torch.uint8, device=a.device) if wsb else None
    # (bench.py's per-product roofline: one span name per product FAMILY -- operand layout, epilogue form, dtype, shape)
    name = "gemm"
    if probe.ACTIVE is not None:
        name = "gemm[%s%s %s] %dx%dx%d" % ("tn" if a_is_km else ("nt" if b_is_nk else "nn"),
                                           ":cross" if x0 is not None else (":res" if r is not None else ""),
                                           "bf16" if a.dtype == torch.bfloat16 else "f32", m, n, k)
    with probe.span(name, 2.0 * m * n * k):
        rc = L.lib().krs_gemm(
            L.ptr(a), C.c_int64(a.stride(0)), C.c_int(int(a_is_km)),
            L.ptr(b), C.c_int64(b.stride(0)), C.c_int(int(b_is_nk)),
            L.ptr(c), C.c_int64(c.stride(0)), C.c_int64(m), C.c_int64(n), C.c_int64(k),
            C.c_int(L.fdtype(a)), C.c_int(L.fdtype(c)), C.byref(ep),
            L.ptr(ws), C.c_size_t(int(wsb)), L.stream_ptr())
    L.check(rc, "krs_gemm")
    return c, u


def cross_epilogue_fwd(u, x0, x, diag_scale=0.0):
    u, x0, x = (_rowmajor(t, "cross_epilogue_fwd").contiguous() for t in (u, x0, x))
    y = torch.empty_like(x)
    m, n = x.shape
    rc = L.lib().krs_cross_epilogue_fwd(L.ptr(u), L.ptr(x0), L.ptr(x), L.ptr(y), C.c_int64(m), C.c_int64(n),
                                        C.c_int64(n), C.c_float(diag_scale or 0.0), C.c_int(L.fdtype(x)),
                                        L.stream_ptr())
    L.check(rc, "krs_cross_epilogue_fwd")
    return y


def _colsum_ws(m: int, n: int, device):
    """(pointer, byte count) arguments of the two-stage column sums (run-to-run identical bias gradients) + the
    tensor that keeps the workspace alive until the call has been enqueued."""
    nbytes = int(L.lib().krs_colsum_workspace_bytes(C.c_int64(m), C.c_int64(n)))
    ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=device)
    return ws, L.ptr(ws), C.c_size_t(nbytes)


def cross_epilogue_bwd(g, u, x0, x, diag_scale=0.0, *, act: int = L.ACT_NONE,
                       dx0_into: torch.Tensor | None = None,
                       want_du=True, want_dxd=True, want_dbias=True, fold_direct=False, want_dx0=True):
    """Returns (du, dx0, dxd, dbias); dx0 accumulates into `dx0_into` when given.
    fold_direct (the case x is x0): the direct term g + diag*g*x0 is added into dx0 instead of
    being written to its own buffer (the C ABI's `dxd == dx0` aliasing rule); dxd is then dx0."""
    g, x0, x = (_rowmajor(t, "cross_epilogue_bwd").contiguous() for t in (g, x0, x))
    # (u None: allowed when neither dL/dx0 nor an activation derivative is wanted -- dz = g * x0 needs no third stream)
    u = None if u is None else _rowmajor(u, "cross_epilogue_bwd").contiguous()
    if u is None and (want_dx0 or act != L.ACT_NONE):
        raise L.KrsError("cross_epilogue_bwd: u is needed for dL/dx0 and for an activation's derivative")
    m, n = x.shape
    du = torch.empty_like(x) if want_du else None
    dx0 = None
    if want_dx0:
        dx0 = dx0_into if dx0_into is not None else torch.empty_like(x)
        if not dx0.is_contiguous():
            raise L.KrsError("cross_epilogue_bwd: dx0 buffer must be contiguous")
    dxd = dx0 if fold_direct else (torch.empty_like(x) if want_dxd else None)
    dbias = torch.empty(n, dtype=torch.float32, device=x.device) if want_dbias else None
    ws, ws_ptr, ws_bytes = _colsum_ws(m, n, x.device) if want_dbias else (None, None, C.c_size_t(0))
    rc = L.lib().krs_cross_epilogue_bwd(
        L.ptr(g), L.ptr(u), L.ptr(x0), L.ptr(x), L.ptr(du), L.ptr(dx0), C.c_int(int(dx0_into is not None)),
        L.ptr(dxd), L.ptr(dbias), C.c_int64(m), C.c_int64(n), C.c_int64(n), C.c_float(diag_scale or 0.0),
        C.c_int(act), C.c_int(L.fdtype(x)), ws_ptr, ws_bytes, L.stream_ptr())
    L.check(rc, "krs_cross_epilogue_bwd")
    return du, dx0, dxd, dbias


def gemm_cross_bwd(a: torch.Tensor, bt: torch.Tensor, r: torch.Tensor, x0: torch.Tensor, u: torch.Tensor, *,
                   act: int = L.ACT_NONE, dx0_into: torch.Tensor | None = None, want_dbias: bool = True,
                   fold_direct: bool = False, beta: float = 1.0, u_upper: torch.Tensor | None = None,
                   want_dx0: bool = True):
    """krs_gemm_cross_bwd: G = A @ Bt^T + beta * R (the data gradient of a cross layer = dL/dy of the layer below it)
    and, from G as stored, the elementwise backward of that layer below -- dz = G x0 act'(u), dx0 = [dx0_into +] G u,
    dbias = column sums of dz -- in ONE launch (fold_direct: the layer below is fed x0 itself, its direct term G joins
    dx0; u_upper: the saved activation output of the layer ABOVE, whose own term R * u_upper then starts dx0 here instead
    of in a matrix that layer would have written; want_dx0=False, without R only: no dL/dx0 from this launch -- `u` is
    handed to the next one as ITS u_upper).  Returns (G, dz, dx0, dbias).  a: [M, K], bt: [N, K] (K-contiguous
    weight), r / x0 / u: [M, N] row-major of a's dtype."""
    a, bt = _rowmajor(a, "gemm_cross_bwd A"), _rowmajor(bt, "gemm_cross_bwd Bt")
    x0, u = (_rowmajor(t, "gemm_cross_bwd operand").contiguous() for t in (x0, u))
    r = None if r is None else _rowmajor(r, "gemm_cross_bwd R").contiguous()      # (None: no residual term)
    m, k = a.shape
    n = bt.shape[0]
    if bt.shape[1] != k or (r is not None and tuple(r.shape) != (m, n)) or tuple(x0.shape) != (m, n) or tuple(u.shape) != (m, n):
        raise L.KrsError("gemm_cross_bwd: shapes do not fit")
    if not (a.dtype == bt.dtype == x0.dtype == u.dtype) or (r is not None and r.dtype != a.dtype):
        raise L.KrsError("gemm_cross_bwd: one dtype for every operand")
    g = torch.empty((m, n), dtype=a.dtype, device=a.device)
    dz = torch.empty_like(g)
    if not want_dx0 and (r is not None or dx0_into is not None or u_upper is not None or fold_direct):
        raise L.KrsError("gemm_cross_bwd: want_dx0=False is the form without R, dx0_into, u_upper and fold_direct")
    dx0 = None if not want_dx0 else (dx0_into if dx0_into is not None else torch.empty_like(g))
    if dx0 is not None and (not dx0.is_contiguous() or dx0.dtype != a.dtype or tuple(dx0.shape) != (m, n)):
        raise L.KrsError("gemm_cross_bwd: dx0 buffer must be a contiguous [M, N] matrix of the operands' dtype")
    if u_upper is not None:
        u_upper = _rowmajor(u_upper, "gemm_cross_bwd u_upper").contiguous()
        if dx0_into is not None or r is None or beta != 1.0 or u_upper.dtype != a.dtype or tuple(u_upper.shape) != (m, n):
            raise L.KrsError("gemm_cross_bwd: u_upper needs R (beta = 1), no dx0 to accumulate into, and the operands' shape / dtype")
    dbias = torch.empty(n, dtype=torch.float32, device=a.device) if want_dbias else None
    nbytes = int(L.lib().krs_gemm_cross_bwd_workspace_bytes(C.c_int64(m), C.c_int64(n))) if want_dbias else 0
    ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=a.device) if want_dbias else None
    # (its own span: the launch is a product AND five to seven [M, N] streams of elementwise work)
    with probe.span("gemm_cross_bwd", 2.0 * m * n * k):
        rc = L.lib().krs_gemm_cross_bwd(
            L.ptr(a), C.c_int64(a.stride(0)), L.ptr(bt), C.c_int64(bt.stride(0)), L.ptr(r),
            C.c_int64(r.stride(0) if r is not None else n),
            C.c_float(beta), L.ptr(g), C.c_int64(n), L.ptr(x0), L.ptr(u), L.ptr(dz), L.ptr(dx0), C.c_int64(n),
            C.c_int(int(dx0_into is not None)), L.ptr(u_upper), C.c_int(int(fold_direct)), L.ptr(dbias), C.c_int64(m),
            C.c_int64(n),
            C.c_int64(k), C.c_int(act),
            C.c_int(L.fdtype(a)), L.ptr(ws), C.c_size_t(nbytes), L.stream_ptr())
    L.check(rc, "krs_gemm_cross_bwd")
    return g, dz, dx0, dbias


def gemm_dense_bwd(a: torch.Tensor, bt: torch.Tensor, y: torch.Tensor, act: int, want_dbias: bool = True):
    """The dense form of krs_gemm_cross_bwd: dz = (A @ Bt^T) * act'(y) and dbias = column sums of dz in ONE launch -- the
    data-gradient product of a Dense layer running the activation backward of the Dense layer BELOW it (whose saved output
    is y) in its epilogue; the raw data gradient is never stored.  a: [M, K], bt: [N, K], y: [M, N].  Returns (dz, dbias)."""
    a, bt = _rowmajor(a, "gemm_dense_bwd A"), _rowmajor(bt, "gemm_dense_bwd Bt")
    y = _rowmajor(y, "gemm_dense_bwd y").contiguous()
    m, k = a.shape
    n = bt.shape[0]
    if bt.shape[1] != k or tuple(y.shape) != (m, n) or not (a.dtype == bt.dtype == y.dtype):
        raise L.KrsError("gemm_dense_bwd: shapes / dtypes do not fit")
    dz = torch.empty((m, n), dtype=a.dtype, device=a.device)
    dbias = torch.empty(n, dtype=torch.float32, device=a.device) if want_dbias else None
    nbytes = int(L.lib().krs_gemm_cross_bwd_workspace_bytes(C.c_int64(m), C.c_int64(n))) if want_dbias else 0
    ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=a.device) if want_dbias else None
    with probe.span("gemm_dense_bwd", 2.0 * m * n * k):
        rc = L.lib().krs_gemm_cross_bwd(
            L.ptr(a), C.c_int64(a.stride(0)), L.ptr(bt), C.c_int64(bt.stride(0)), None, C.c_int64(n), C.c_float(0.0),
            None, C.c_int64(n), None, L.ptr(y), L.ptr(dz), None, C.c_int64(n), C.c_int(0), None, C.c_int(0), L.ptr(dbias),
            C.c_int64(m), C.c_int64(n), C.c_int64(k), C.c_int(act), C.c_int(L.fdtype(a)), L.ptr(ws), C.c_size_t(nbytes),
            L.stream_ptr())
    L.check(rc, "krs_gemm_cross_bwd (dense form)")
    return dz, dbias


def colsum(a: torch.Tensor) -> torch.Tensor:
    a = _rowmajor(a, "colsum")
    out = torch.empty(a.shape[1], dtype=torch.float32, device=a.device)
    ws, ws_ptr, ws_bytes = _colsum_ws(a.shape[0], a.shape[1], a.device)
    rc = L.lib().krs_colsum(L.ptr(a), C.c_int64(a.stride(0)), C.c_int64(a.shape[0]), C.c_int64(a.shape[1]),
                            C.c_int(L.fdtype(a)), L.ptr(out), ws_ptr, ws_bytes, L.stream_ptr())
    L.check(rc, "krs_colsum")
    return out


def dense_act_bwd(g: torch.Tensor, y: torch.Tensor | None, act: int, want_dbias: bool = True):
    """(dz, dbias) of a Dense layer's bias + activation epilogue: dz = g * act'(y) from the saved output y
    (None without an activation), dbias = fp32 column sums of dz (None if not wanted); krs_dense_act_bwd."""
    g = _rowmajor(g, "dense_act_bwd g")
    m, n = g.shape
    if y is not None:
        y = _rowmajor(y, "dense_act_bwd y")
        if y.dtype != g.dtype or tuple(y.shape) != (m, n):
            raise L.KrsError("dense_act_bwd: y must have the shape and dtype of g")
    need_dz = act != L.ACT_NONE
    dz = torch.empty((m, n), dtype=g.dtype, device=g.device) if need_dz else None
    db = torch.empty(n, dtype=torch.float32, device=g.device) if want_dbias else None
    if need_dz or want_dbias:
        ws, ws_ptr, ws_bytes = _colsum_ws(m, n, g.device) if want_dbias else (None, None, C.c_size_t(0))
        rc = L.lib().krs_dense_act_bwd(L.ptr(g), C.c_int64(g.stride(0)), L.ptr(y if need_dz else None),
                                       C.c_int64(y.stride(0) if (y is not None and need_dz) else n),
                                       L.ptr(dz), C.c_int64(n), L.ptr(db), C.c_int64(m), C.c_int64(n), C.c_int(act),
                                       C.c_int(L.fdtype(g)), ws_ptr, ws_bytes, L.stream_ptr())
        L.check(rc, "krs_dense_act_bwd")
    return (dz if need_dz else g), db


def _cast_cache_key(w: torch.Tensor, dtype: torch.dtype):
    return (w.data_ptr(), w._version, tuple(w.shape), w.dtype, dtype)


def cast_transpose(w: torch.Tensor, dtype: torch.dtype, want_plain: bool = True, want_t: bool = True):
    """(w.to(dtype), w.to(dtype).t().contiguous()) of a 2-D weight in ONE launch (krs_cast_transpose); an output
    that is not wanted is None; the plain one is `w` itself when it already has the dtype and is row-major.

    Parameters remember that they were asked for (`_krs_cast_want`): `refresh_casts` -- called by
    keras_rs_amd.optim.Adagrad(prepare_casts=True) right behind its update -- then prepares the copies of ALL such
    weights for the next step in one launch and leaves them on the parameter (`_krs_cast`), where this function finds
    them ONCE (keyed by storage address, torch version, shape and dtypes)."""
    w = _rowmajor(w, "cast_transpose")
    if isinstance(w, torch.nn.Parameter) and w.dtype != dtype and want_plain and want_t and w.is_contiguous():
        hit = getattr(w, "_krs_cast", None)
        if hit is not None:
            w._krs_cast = None        # one use: the first forward behind the optimizer step that prepared it
            if hit[0] == _cast_cache_key(w, dtype):
                return hit[1], hit[2]
        w._krs_cast_want = dtype
    rows, cols = w.shape
    plain = None
    if want_plain:
        plain = w if w.dtype == dtype else torch.empty((rows, cols), dtype=dtype, device=w.device)
    wt = torch.empty((cols, rows), dtype=dtype, device=w.device) if want_t else None
    write_plain = plain is not None and plain is not w
    if write_plain or want_t:
        rc = L.lib().krs_cast_transpose(L.ptr(w), C.c_int64(rows), C.c_int64(cols), C.c_int64(w.stride(0)),
                                        C.c_int(L.fdtype(w)), L.ptr(plain) if write_plain else None,
                                        C.c_int64(cols), L.ptr(wt), C.c_int64(rows),
                                        C.c_int(L.fdtype(plain if plain is not None else wt)), L.stream_ptr())
        L.check(rc, "krs_cast_transpose")
    return plain, wt


def refresh_casts(params) -> int:
    """Cast + transposed copies of every parameter that a layer has asked `cast_transpose` for, in ONE launch
    (krs_cast_transpose_many); to be called when the weights have just changed (behind the optimizer step: the C-ABI
    update does not go through torch, so the copies are keyed by storage and version AFTER it).  Returns the count."""
    todo = [p for p in params if getattr(p, "_krs_cast_want", None) is not None and p.is_cuda and p.dim() == 2
            and p.is_contiguous()]
    by_dtype: dict = {}
    for p in todo:
        by_dtype.setdefault((p.dtype, p._krs_cast_want), []).append(p)
    for (sdt, ddt), ps in by_dtype.items():
        n = len(ps)
        # the copies live in buffers that belong to the parameter and are rewritten in place, step after step (stream
        # order puts the rewrite behind the last reader: the backward pass of the step whose update precedes it) --
        # fixed addresses, so a step captured in a graph (torch.cuda.graph) keeps reading what its replays write
        plains, trans = [], []
        for p in ps:
            buf = getattr(p, "_krs_cast_buf", None)
            if buf is None or buf[0].dtype != ddt or buf[0].device != p.device or tuple(buf[0].shape) != tuple(p.shape):
                buf = p._krs_cast_buf = (torch.empty(tuple(p.shape), dtype=ddt, device=p.device),
                                         torch.empty((p.shape[1], p.shape[0]), dtype=ddt, device=p.device))
            plains.append(buf[0])
            trans.append(buf[1])
        ptrs = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])  # noqa: E731
        rc = L.lib().krs_cast_transpose_many(
            C.c_int(n), ptrs(ps), (C.c_int64 * n)(*[p.shape[0] for p in ps]), (C.c_int64 * n)(*[p.shape[1] for p in ps]),
            C.c_int(L.fdtype(ps[0])), ptrs(plains), ptrs(trans), C.c_int(L.fdtype(plains[0])), L.stream_ptr())
        L.check(rc, "krs_cast_transpose_many")
        for p, a, b in zip(ps, plains, trans):
            # the C ABI rewrote these buffers behind torch's back: bump their version counters (host-side bookkeeping,
            # no kernel), so that a backward pass that still holds them from an EARLIER forward -- a retained graph,
            # two forwards before one backward across an optimizer step, a teacher / EMA forward -- fails autograd's
            # saved-tensor check loudly instead of differentiating against the new weights' copies
            torch.autograd.graph.increment_version(a)
            torch.autograd.graph.increment_version(b)
            p._krs_cast = (_cast_cache_key(p, ddt), a, b)
    return len(todo)


def _ptr_table(ts: Sequence[torch.Tensor]):
    return (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts]), (C.c_int64 * len(ts))(*[t.stride(0) for t in ts])


def dot_out_cols(n_feats: int, self_interaction: bool, skip_gather: bool) -> int:
    if skip_gather:
        return n_feats * n_feats
    return n_feats * (n_feats + 1) // 2 if self_interaction else n_feats * (n_feats - 1) // 2


def dot_interaction_fwd(feats: Sequence[torch.Tensor], self_interaction=False, skip_gather=False):
    feats = [_rowmajor(f, "dot_interaction feature") for f in feats]
    batch, dim = feats[0].shape
    out = torch.empty((batch, dot_out_cols(len(feats), self_interaction, skip_gather)), dtype=feats[0].dtype,
                      device=feats[0].device)
    ptrs, lds = _ptr_table(feats)
    rc = L.lib().krs_dot_interaction_fwd(ptrs, lds, C.c_int(len(feats)), C.c_int64(batch), C.c_int(dim),
                                         C.c_int(L.fdtype(feats[0])), C.c_int(int(self_interaction)),
                                         C.c_int(int(skip_gather)), L.ptr(out), C.c_int64(out.stride(0)),
                                         L.stream_ptr())
    L.check(rc, "krs_dot_interaction_fwd")
    return out


def dot_interaction_bwd(feats: Sequence[torch.Tensor], grad_out: torch.Tensor, self_interaction=False,
                        skip_gather=False, into: torch.Tensor | None = None, accumulate_mask: int = 0):
    """Gradients of the features, as column views of one [B, F*dim] buffer.

    into / accumulate_mask (krs_dot_interaction_bwd_accumulate): `into` is a [B, F*dim] row-major matrix that
    already holds gradients of the features; feature f with bit f of the mask set is ADDED there (its returned
    gradient is None: the caller hands `into` on), the others are written to a fresh buffer as usual."""
    feats = [_rowmajor(f, "dot_interaction feature") for f in feats]
    grad_out = _rowmajor(grad_out, "dot_interaction grad")
    batch, dim = feats[0].shape
    n = len(feats)
    # one buffer, per-feature column views: a consumer that wants the gradients side by side
    # (the embedding backward) can take the buffer as it is
    if into is None or not accumulate_mask:
        accumulate_mask = 0
        gbuf = torch.empty((batch, n * dim), dtype=feats[0].dtype, device=feats[0].device)
        grads = [gbuf[:, i * dim:(i + 1) * dim] for i in range(n)]
        outs = grads
    else:
        if tuple(into.shape) != (batch, n * dim) or into.dtype != feats[0].dtype or into.stride(1) != 1:
            raise L.KrsError("dot_interaction_bwd: `into` must be a row-major [batch, F*dim] matrix of the features' dtype")
        fresh = [i for i in range(n) if not (accumulate_mask >> i) & 1]
        gbuf = torch.empty((batch, max(len(fresh), 1) * dim), dtype=feats[0].dtype, device=feats[0].device)
        slot = {f: k for k, f in enumerate(fresh)}
        outs = [into[:, i * dim:(i + 1) * dim] if i not in slot else gbuf[:, slot[i] * dim:(slot[i] + 1) * dim]
                for i in range(n)]
        grads = [None if i not in slot else outs[i] for i in range(n)]
    ptrs, lds = _ptr_table(feats)
    gptrs, glds = _ptr_table(outs)
    rc = L.lib().krs_dot_interaction_bwd_accumulate(ptrs, lds, C.c_int(n), C.c_int64(batch), C.c_int(dim),
                                                    C.c_int(L.fdtype(feats[0])), C.c_int(int(self_interaction)),
                                                    C.c_int(int(skip_gather)), L.ptr(grad_out),
                                                    C.c_int64(grad_out.stride(0)), gptrs, glds,
                                                    C.c_uint64(int(accumulate_mask)), L.stream_ptr())
    L.check(rc, "krs_dot_interaction_bwd_accumulate")
    return grads


def mod_bucketize(ids: torch.Tensor, n_shards: int):
    """Stable grouping of ids by id % n_shards.  Returns (local_ids, perm, bucket_counts[int64])."""
    L.require_device(ids, "ids")
    ids = ids.contiguous().reshape(-1)
    nnz = ids.numel()
    local = torch.empty_like(ids)
    perm = torch.empty(nnz, dtype=torch.int32, device=ids.device)
    counts = torch.empty(n_shards, dtype=torch.int64, device=ids.device)
    wsb = L.lib().krs_mod_bucketize_workspace_bytes(C.c_int64(nnz), C.c_int(n_shards))
    ws = torch.empty(max(int(wsb), 1), dtype=torch.uint8, device=ids.device)
    rc = L.lib().krs_mod_bucketize(L.ptr(ids), C.c_int(L.itype(ids)), C.c_int64(nnz), C.c_int(n_shards),
                                   L.ptr(local), L.ptr(perm), L.ptr(counts), L.ptr(ws), C.c_size_t(ws.numel()),
                                   L.stream_ptr())
    L.check(rc, "krs_mod_bucketize")
    return local, perm, counts


def bce_fwd_bwd(pred: torch.Tensor, labels: torch.Tensor, epsilon: float = 1e-7, grad_scale: float = 1.0,
                want_grad: bool = True):
    """(loss [] fp32, dL/dpred | None): binary cross-entropy of probabilities, mean reduction, forward and backward in
    one pass (krs_bce_fwd_bwd; keras.losses.BinaryCrossentropy() of examples/ml_perf/main.py:201-210)."""
    L.require_device(pred, "bce pred")
    p = pred.reshape(-1)
    if not p.is_contiguous():
        p = p.contiguous()
    y = labels.reshape(-1).to(torch.float32)
    if not y.is_contiguous():
        y = y.contiguous()
    if y.numel() != p.numel():
        raise L.KrsError(f"bce: {p.numel()} predictions but {y.numel()} labels")
    loss = torch.empty((), dtype=torch.float32, device=pred.device)
    dp = torch.empty_like(p) if want_grad else None
    scratch = torch.empty(64, dtype=torch.float32, device=pred.device)     # KRS_BCE_MAX_BLOCKS partial sums
    rc = L.lib().krs_bce_fwd_bwd(L.ptr(p), C.c_int(L.fdtype(p)), L.ptr(y), C.c_int64(p.numel()), C.c_float(epsilon),
                                 C.c_float(grad_scale), L.ptr(loss), L.ptr(dp), L.ptr(scratch), L.stream_ptr())
    L.check(rc, "krs_bce_fwd_bwd")
    return loss, (None if dp is None else dp.reshape(pred.shape))
