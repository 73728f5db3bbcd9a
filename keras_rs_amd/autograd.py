"""torch.autograd.Function wrappers: one per op of the hot path, each calling the
HIP kernels through the C ABI (embedding_ops / dense_ops).  Gradients follow
SURVEY.md section 8 rows a4 (embedding), a9 (FeatureCross) and a11 (DotInteraction).
"""

from __future__ import annotations

import weakref

import torch

from keras_rs_amd import _lib as L
from keras_rs_amd import dense_ops as D


_SIDE_STREAMS: dict = {}


def _side_stream(device) -> "torch.cuda.Stream":
    key = str(device)
    st = _SIDE_STREAMS.get(key)
    if st is None:
        st = _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return st


# Weight-gradient GEMMs of the cross layers on a second stream (see CrossLayerFn.backward): they are off the critical
# path, and started behind a layer's dx product they run beside the HBM-bound dz / dx0 pass of the layer below (or the
# DotInteraction gradient and the table update behind the bottom layer) -- an elementwise pass and a ring GEMM do overlap
# a little on this part (scripts/exp/overlap_probe2.py: 563 + 271 us apart, 707 together), two GEMMs or a GEMM and the
# table update do not.  Measured in the step: 10.31-10.48 -> 10.20-10.29 ms.  Deferring all six to the table update was
# slower (10.49).
# OPT-IN since round 4 (`set_wgrad_side_stream(True)` or KRS_WGRAD_SIDE=1), and no longer used by bench.py / the example:
# with the elementwise backward inside the data-gradient products (krs_gemm_cross_bwd) the pass these GEMMs overlapped
# with is gone -- beside the next layer's ring GEMMs they measured 10.26 against 10.13-10.18 ms
# (profiles/archive/r4k_wgrad_side_ab.txt).  Also: a gradient produced on a private stream is only safe when NOTHING but the end-of-backward rejoin reads
# it, and a backward function cannot see every reader -- a second gradient contribution to the same weight (a manual
# L2 term, a weight shared by two layers) is summed by autograd's input buffer on the main stream with nothing ordering
# it behind this stream.  The owner of the training step can promise that; a library default cannot.  What the
# function can see it still checks (`_wgrad_side_ok`): an accumulated .grad, tensor hooks, foreign post-accumulate
# hooks, a regulariser on the weight (layers.base marks it: its penalty is a second contribution), the weight used by
# more than one pending CrossLayerFn.
WGRAD_SIDE_STREAM = bool(int(__import__("os").environ.get("KRS_WGRAD_SIDE", "0")))
WGRAD_SIDE_MIN_ROWS = 32768
_WGRAD_STREAMS: dict = {}
_WGRAD_SYNC_QUEUED: set = set()


def set_wgrad_side_stream(on: bool) -> bool:
    """Switch the second stream for the cross layers' weight gradients on or off (see above); returns the old value."""
    global WGRAD_SIDE_STREAM
    old, WGRAD_SIDE_STREAM = WGRAD_SIDE_STREAM, bool(on)
    return old


def _wgrad_stream(device) -> "torch.cuda.Stream":
    key = str(device)
    st = _WGRAD_STREAMS.get(key)
    if st is None:
        st = _WGRAD_STREAMS[key] = torch.cuda.Stream(device=device)
    return st


def _wgrad_side_ok(w) -> bool:
    """May this weight's gradient come off the second stream?  Only when nothing reads it before the end of the backward
    pass: no gradient to accumulate into yet (the first accumulation is an assignment, a later one is an add kernel on
    the main stream), no tensor hooks, no post-accumulate hooks other than ones that rejoin the stream themselves
    (dp.GradAllReduce marks its parameters), no second consumer this function knows of (a regulariser attached by
    `Layer.add_weight`, a second CrossLayerFn whose backward is still pending)."""
    if w is None:
        return True
    if w.grad is not None or w._backward_hooks:
        return False
    if getattr(w, "_krs_has_regularizer", False) or getattr(w, "_krs_pending_cross", 1) != 1:
        return False
    hooks = getattr(w, "_post_accumulate_grad_hooks", None)
    return not hooks or bool(getattr(w, "_krs_hooks_rejoin_wgrad_stream", False))


def _release_pending_cross(w_refs, counted) -> None:
    """Gives back one pending use of each weight of a CrossLayerFn call, once: from its backward, or from the finalizer
    of its graph node when the graph is dropped without a backward pass."""
    if not counted[0]:
        return
    counted[0] = False
    for r in w_refs:
        w = r()
        if w is not None:
            w._krs_pending_cross = max(0, getattr(w, "_krs_pending_cross", 1) - 1)


def _release_plan_ws(bags_ref, owns) -> None:
    """The plan workspace kept on a FusedBags is free again (EmbedBagFusedFn: after its backward, or when the graph
    that borrowed it is dropped without one -- otherwise every later step would allocate a fresh workspace)."""
    if not owns[0]:
        return
    owns[0] = False
    bags = bags_ref()
    if bags is not None:
        bags._plan_ws_busy = False


def wgrad_stream_sync() -> None:
    """The current stream waits for the weight-gradient stream(s): before anything reads those gradients (the end of
    the backward pass does it by itself; dp.GradAllReduce calls it before it reduces a gradient)."""
    for st in _WGRAD_STREAMS.values():
        torch.cuda.current_stream(st.device).wait_stream(st)


def _queue_wgrad_sync(task: int) -> None:
    """Once per backward pass: rejoin the main stream when the graph task ends."""
    if task in _WGRAD_SYNC_QUEUED:
        return
    if task == -1:          # not inside a backward pass (a direct call): rejoin at once
        wgrad_stream_sync()
        return
    _WGRAD_SYNC_QUEUED.add(task)

    def done():
        _WGRAD_SYNC_QUEUED.discard(task)
        wgrad_stream_sync()

    torch.autograd.Variable._execution_engine.queue_callback(done)


def _split_columns(out: torch.Tensor, n: int, dim: int, lead: int = 0):
    """Per-feature [B, dim] column views of the fused [B, lead + n*dim] lookup output."""
    return tuple(out[:, lead + i * dim:lead + (i + 1) * dim] for i in range(n))


def _sum_slab_and_feature_grads(g_slab, gs, lead, batch, n, dim, dtype, device):
    """Gradient of the fused lookup output [B, n*dim] from the gradient of the slab (its columns
    lead..) and of the per-feature views; either side may be absent."""
    gv = _gather_feature_grads(gs, batch, dim, dtype, device) if any(g is not None for g in gs) else None
    if g_slab is None:
        return gv if gv is not None else torch.zeros((batch, n * dim), dtype=dtype, device=device)
    g_s = g_slab[:, lead:lead + n * dim]
    return g_s if gv is None else g_s + gv


def _gather_feature_grads(grads, batch: int, dim: int, dtype, device):
    """The per-feature output gradients as ONE [B, n*dim] matrix for K2.  When autograd hands
    back consecutive column slices of a single buffer (the gradient of a concat), that buffer is
    used in place (zero copies); otherwise the pieces are concatenated once."""
    n = len(grads)
    if all(g is not None for g in grads):
        g0 = grads[0]
        es = g0.element_size()
        ok = g0.dim() == 2 and g0.stride(1) == 1
        for i, g in enumerate(grads):
            ok = ok and g.dtype == g0.dtype and g.stride() == g0.stride() and \
                g.untyped_storage().data_ptr() == g0.untyped_storage().data_ptr() and \
                g.data_ptr() - g0.data_ptr() == i * dim * es
        if ok and g0.stride(0) >= n * dim:
            return torch.as_strided(g0, (batch, n * dim), (g0.stride(0), 1), g0.storage_offset())
    parts = [g if g is not None else torch.zeros((batch, dim), dtype=dtype, device=device) for g in grads]
    return torch.cat(parts, dim=1)


class Dx0Relay:
    """Carries dL/dx0 down a stack of cross layers that share x0 (`xl = layer(x0, xl)` repeated,
    examples/ml_perf/model.py:332-336).  Every layer of the stack contributes a [B, d] term to dL/dx0;
    left to autograd these are summed by separate elementwise adds (1.4 GB of traffic each at C3).
    Instead the layer whose backward runs first writes its term into a buffer and leaves it here, on the
    relay it shares with the layer that produced its `x`; that layer's backward -- which autograd is bound
    to run next on this path, since it is handed dL/dx -- accumulates into the buffer inside its
    elementwise kernel (`dx0_accumulate` of krs_cross_epilogue_bwd) and passes it on, and the bottom layer
    of the stack returns the total.  `task` is the autograd graph-task id of the pass that filled `buf`:
    a buffer left over from another backward pass is ignored."""

    __slots__ = ("x0", "buf", "task", "lower", "fused")

    def __init__(self, x0: torch.Tensor):
        self.x0, self.buf, self.task = x0, None, -1
        # lower: what the consumer of y needs to run THIS layer's elementwise backward inside its own data-gradient
        #   product (krs_gemm_cross_bwd): (u, act, diag_scale, has_bias, x_is_x0, x0 in the compute dtype, (relay of the
        #   layer that produced this layer's x | None, dtype of that x)), left by this layer's forward;
        # fused: (G, version of G, dz, dbias, task, deferred) left by that consumer's backward when it did so -- `buf` then
        #   already holds this layer's term of dL/dx0 as well, unless `deferred`: then this layer's own data-gradient
        #   product computes it (u_upper) and no matrix exists yet.
        self.lower, self.fused = None, None

    def matches(self, x0: torch.Tensor) -> bool:
        a = self.x0
        return (a.data_ptr() == x0.data_ptr() and a.shape == x0.shape and a.stride() == x0.stride()
                and a.dtype == x0.dtype and a._version == x0._version)


class SlabGradRelay:
    """Lets the gradient of DotInteraction join the gradient of the concat of the same features inside the
    DotInteraction backward kernel instead of in a separate 1.4 GB add.  `layers.concat_features([dense,
    *embeddings])` returns the lookup slab itself, so the gradient of the concat IS the gradient of the slab: its
    backward (SlabFillFn) leaves that matrix here; when DotInteraction's backward runs later in the same pass --
    it does whenever the interaction was called before the concat, as in a DLRM forward -- and its trailing
    inputs are exactly the slab's feature views, it adds their gradients into the matrix
    (krs_dot_interaction_bwd_accumulate) and returns no gradient for them.  The leading inputs (the bottom-MLP
    output) keep their ordinary gradient tensors, so whatever else consumes them is unaffected.  Not used when
    the concat result is still referenced and retains its gradient or has hooks (its .grad would show the joined
    value), nor when one of the slab's feature views does (it would see no gradient from the interaction), nor once
    the slab has a second consumer (a second concat of the same features).  Known limit: a gradient of the concat
    result CAPTURED by torch.autograd.grad(..., inputs=[concat_out]) is the very tensor the interaction adds into,
    so it shows the joined value (the engine exposes no way to see a capture from inside a backward)."""

    __slots__ = ("buf", "task", "out_ref", "disabled")
    joined = 0   # times the in-kernel path was taken (read by the tests)

    def __init__(self):
        self.buf, self.task, self.out_ref = None, -1, None
        # set when the slab got a second consumer (a second concat of the same features): autograd then sums the two
        # gradients of the slab in a buffer of its own, and an in-place add into the first one could be lost
        self.disabled = False


# Round 4: dx = dh U^T + g of a layer IS dL/dy of the layer below it in the stack, whose elementwise backward starts
# by re-reading it.  krs_gemm_cross_bwd does that pass in the product's epilogue (one 453 MB stream less per layer at C3,
# and the other streams at the epilogue's rate): environment KRS_FUSE_CROSS_BWD=0 switches it off (A/B).
FUSE_CROSS_BWD = bool(int(__import__("os").environ.get("KRS_FUSE_CROSS_BWD", "1")))
# ... and the TOP layer's own term of dL/dx0 is computed in that epilogue too (it writes dz only): KRS_FUSE_TOP_DX0=0 for A/B
FUSE_TOP_DX0 = bool(int(__import__("os").environ.get("KRS_FUSE_TOP_DX0", "1")))


def _fusable_below(up, a, x_dtype, task) -> bool:
    """May the product `a @ Bt^T` run the elementwise backward of the cross layer behind relay `up` in its epilogue?"""
    low = up.lower if up is not None else None
    return not (low is None or task == -1 or not FUSE_CROSS_BWD or low[2] != 0.0 or a.dtype != torch.bfloat16
                or x_dtype != a.dtype       # (the product is handed to autograd as it is: a cast would be a different tensor)
                or low[5].dtype != a.dtype or (up.buf is not None and up.task == task))


def _dense_fusable_below(rel, dz, x_dtype, task) -> bool:
    """May the product `dz @ K^T` run the activation backward of the Dense layer behind relay `rel` in its epilogue?"""
    if rel is None or rel.lower is None or not FUSE_DENSE_BWD or task == -1 or x_dtype != dz.dtype:
        return False
    act_low, bias_low = rel.lower
    if act_low == L.ACT_NONE and not bias_low:
        return False            # (nothing to do below: a plain product)
    if rel.fused is not None and rel.fused[3] == task:
        return False            # (another consumer of that output already did it in this pass)
    out = rel.out_ref() if rel.out_ref is not None else None
    # (somebody watches the layer's output gradient: hand autograd the real dL/dy)
    return out is None or not (out.retains_grad or out._backward_hooks)


def _dx_product(ctx, dh, dc, direct, dx0, x0c, task, u_own=None):
    """dx = dh U^T + direct.  When x was produced by a cross layer on the same x0 (ctx.relay_up) whose elementwise
    backward can ride in this product's epilogue, it does: that layer's dz / dbias and its term of dL/dx0 (added into
    this layer's dx0 buffer) are left on its relay.  Returns (dx, dx0).
    dx0 None + u_own (the TOP layer of a stack, see CrossLayerFn.backward): this layer wrote no dL/dx0 of its own; its term
    direct * u_own starts the matrix inside the fused epilogue."""
    up = ctx.relay_up
    if dx0 is None:
        u_low, act_low, _, bias_low, same_low = up.lower[:5]
        dx, dz_low, dx0, db_low = D.gemm_cross_bwd(dh, dc, direct, x0c, u_low, act=act_low, want_dbias=bias_low,
                                                   fold_direct=same_low, u_upper=u_own)
        up.fused = (dx, dx._version, dz_low, db_low, task, False)
        return dx, dx0
    if not _fusable_below(up, dh, ctx.meta[6], task) or not dx0.is_contiguous() or dx0.dtype != dh.dtype:
        dx, _ = D.gemm(dh, dc, b_is_nk=True, r=direct, beta=1.0)
        return dx, dx0
    u_low, act_low, _, bias_low, same_low = up.lower[:5]
    dx, dz_low, dx0, db_low = D.gemm_cross_bwd(dh, dc, direct, x0c, u_low, act=act_low, dx0_into=dx0,
                                               want_dbias=bias_low, fold_direct=same_low)
    up.fused = (dx, dx._version, dz_low, db_low, task, False)
    return dx, dx0


class CrossLayerFn(torch.autograd.Function):
    """y = x0 * (act(h @ K + b) + diag * x) + x,  h = x (full rank) or x @ U (low rank).

    Reference: FeatureCross.call, feature_cross.py:182-194.  Forward = one MFMA GEMM
    per Dense with the cross epilogue fused; backward = one elementwise pass
    (dz, dx0, bias gradient) + the data/weight-gradient GEMMs.
    Weight layouts are the keras ones: U [d, p], K [p or d, d], b [d].
    relay_in: the Dx0Relay a consumer of y may fill; relay_up: the relay of the layer that produced x
    (None when x does not come from a cross layer on the same x0)."""

    @staticmethod
    def forward(ctx, x0, x, down, kernel, bias, diag_scale, act, compute_dtype, relay_in=None, relay_up=None):
        same = x is x0 or (x.data_ptr() == x0.data_ptr() and x.shape == x0.shape and x.stride() == x0.stride())
        cd = compute_dtype
        x0c = x0.to(cd).contiguous()
        xc = x0c if same else x.to(cd).contiguous()
        # The forward contracts over the kernels' leading axis: hand the (small) weights to the MFMA
        # kernel K-contiguous (one transposed copy per step, a few microseconds) so that both GEMM
        # operands stream into LDS with row-wise 16-byte stores.
        # (cast + transposed copy of a weight = one launch, krs_cast_transpose)
        kc, kct = D.cast_transpose(kernel, cd)
        h = xc
        dc = None
        if down is not None:
            dc, dct = D.cast_transpose(down, cd)
            h, _ = D.gemm(xc, dct, b_is_nk=True)
        y, u = D.gemm(h, kct, b_is_nk=True, bias=bias, act=act, diag_scale=diag_scale,
                      x0=x0c, x=xc, want_u=True)
        ctx.save_for_backward(x0c, xc, h if down is not None else None, u, dc, kc)
        ctx.meta = (diag_scale, act, same, down is not None, bias is not None,
                    x0.dtype, x.dtype, None if down is None else down.dtype, kernel.dtype)
        ctx.relay_in = relay_in
        both = ctx.needs_input_grad[0] and ctx.needs_input_grad[1]
        ctx.relay_up = relay_up if (relay_up is not None and not same and both and relay_up.matches(x0)) else None
        if relay_in is not None and FUSE_CROSS_BWD and down is not None:
            relay_in.lower = (u, act, float(diag_scale or 0.0), bias is not None, same, x0c, (ctx.relay_up, x.dtype))
        ctx.w_refs = tuple(weakref.ref(w) for w in (down, kernel) if w is not None)
        # pending uses of each weight (a weight shared by two layer calls gets two gradient contributions)
        # (the count is given back by the backward -- or, when no backward ever runs on this graph (a grad-enabled
        #  validation pass, an exception, a discarded output), by the finalizer of the graph node: a count left above one
        #  would keep the weight off the second stream for good, silently)
        ctx.counted = [any(ctx.needs_input_grad[i] for i in (2, 3))]
        if ctx.counted[0]:
            for w in (down, kernel):
                if w is not None:
                    w._krs_pending_cross = getattr(w, "_krs_pending_cross", 0) + 1
            weakref.finalize(ctx, _release_pending_cross, ctx.w_refs, ctx.counted)
        return y

    @staticmethod
    def backward(ctx, g):
        x0c, xc, h, u, dc, kc = ctx.saved_tensors
        diag, act, same, low_rank, has_bias, x0_dt, x_dt, down_dt, k_dt = ctx.meta
        g = g.to(x0c.dtype).contiguous()
        need_dxd = bool(diag) and not same
        task = torch._C._current_graph_task_id()
        # dL/dx0 of the layers above, left by the consumer of y during this backward pass
        incoming, extra = None, None
        rin = ctx.relay_in
        fused = None
        if rin is not None:
            fused, rin.fused = rin.fused, None
        if rin is not None and rin.buf is not None:
            if rin.task == task and task != -1:
                if rin.buf.dtype == x0c.dtype and rin.buf.shape == x0c.shape and rin.buf.is_contiguous():
                    incoming = rin.buf
                else:
                    extra = rin.buf
            rin.buf = None
        if fused is not None and fused[5]:
            # deferred: the consumer (a Dense layer) ran this layer's elementwise backward but left its term of dL/dx0 to
            # THIS layer's data-gradient product (u_upper).  Anything that is not the plain case -- another consumer's
            # buffer, a summed gradient, the layer below no longer fusable -- redoes this layer from g in the branches below.
            if not (incoming is None and extra is None and fused[4] == task and g.data_ptr() == fused[0].data_ptr()
                    and g.shape == fused[0].shape and g._version == fused[1]
                    and _fusable_below(ctx.relay_up, g, x_dt, task)):
                fused = None
        elif fused is not None and (incoming is None or fused[4] != task):
            fused = None
        defer_dx0 = False
        if fused is not None and fused[5]:
            dz, dbias = fused[2], fused[3]
            dx0, dxd, defer_dx0 = None, None, True
        elif fused is not None:
            # The consumer of y ran this layer's elementwise backward inside its data-gradient product
            # (krs_gemm_cross_bwd): `incoming` already holds this layer's term, dz and dbias are done.  That is only
            # right when its G is ALL of dL/dy -- the very tensor autograd hands over, untouched.  If y had another
            # consumer the engine summed their gradients (a new tensor, or in place: the version moves): the difference
            # goes through the elementwise kernel (linear in g), dz and dbias are redone.
            G, g_version, dz, dbias = fused[:4]
            if g.data_ptr() == G.data_ptr() and g.shape == G.shape and g._version == g_version:
                dx0, dxd = incoming, (incoming if same else None)
            else:
                delta = (g.float() - G.float()).to(g.dtype)
                _, dx0, _, _ = D.cross_epilogue_bwd(delta, u, x0c, xc, diag, act=act, want_du=False, want_dxd=False,
                                                    want_dbias=False, fold_direct=same, dx0_into=incoming)
                dz, _, _, dbias = D.cross_epilogue_bwd(g, u, x0c, xc, diag, act=act, want_dxd=False,
                                                       want_dbias=has_bias, want_dx0=False)
                dxd = dx0 if same else None
        elif (incoming is None and extra is None and not same and low_rank and not diag and FUSE_TOP_DX0
              and _fusable_below(ctx.relay_up, g, x_dt, task)):
            # The TOP layer of a stack (nothing above it left a dL/dx0) whose data-gradient product is going to run the
            # layer below's elementwise backward: its own term g * u is computed THERE (g is that product's residual, already
            # in registers), so this pass writes dz only -- three [B, d] streams instead of five, and no matrix for dL/dx0 is
            # written here and read back there.
            dz, _, dxd, dbias = D.cross_epilogue_bwd(g, u if act != L.ACT_NONE else None, x0c, xc, diag, act=act,
                                                     want_dxd=False, want_dbias=has_bias, want_dx0=False)
            dx0, defer_dx0 = None, True
        else:
            # x is x0 (the first layer of a stack): both halves of dL/dx0 go through one buffer, which
            # the data-gradient GEMM then takes as its residual -- no separate add.
            dz, dx0, dxd, dbias = D.cross_epilogue_bwd(g, u, x0c, xc, diag, act=act, want_dxd=need_dxd,
                                                       want_dbias=has_bias, fold_direct=same, dx0_into=incoming)
        if extra is not None:
            dx0 = dx0 + extra.to(dx0.dtype)
        u_own = u if defer_dx0 else None
        direct = dx0 if same else (dxd if need_dxd else g)  # dL/dx through "+ x" and "diag * x"
        # (worth it when the kernels are long against a launch: at a per-rank batch of 8192 the step is bound by the
        #  host's enqueue rate and the extra events cost more than the overlap returns: 2.42 -> 2.54 ms)
        # a weight with several pending uses gets several contributions in THIS pass (summed by autograd's input buffer
        # on the main stream): the first backward that runs sees the full count and marks the pass, the later ones --
        # which see a count of one again -- read the mark
        shared = False
        for r in ctx.w_refs:
            w = r()
            if w is None:
                continue
            if getattr(w, "_krs_pending_cross", 1) != 1:
                w._krs_shared_in_task = task
            shared = shared or (task != -1 and getattr(w, "_krs_shared_in_task", None) == task)
        side_ok = low_rank and WGRAD_SIDE_STREAM and dz.is_cuda and dz.shape[0] >= WGRAD_SIDE_MIN_ROWS and \
            not shared and all(_wgrad_side_ok(r()) for r in ctx.w_refs)
        _release_pending_cross(ctx.w_refs, ctx.counted)
        if side_ok:
            # Data-gradient path first; the two weight gradients (off the critical path: only the optimizer reads them)
            # go to a second stream that starts when dx is done -- i.e. beside the HBM-bound kernel that follows on the
            # main stream (the dz / dx0 pass of the layer below, or the DotInteraction gradient and the table update
            # behind the bottom layer).  The main stream rejoins at the end of the backward pass (wgrad_stream_sync).
            dh, _ = D.gemm(dz, kc, b_is_nk=True)                               # dh = dz K^T     [B, p]
            dx, dx0 = _dx_product(ctx, dh, dc, direct, dx0, x0c, task, u_own)  # dx = dh U^T + direct
            main = torch.cuda.current_stream()
            side = _wgrad_stream(dz.device)
            ev = torch.cuda.Event()
            ev.record(main)
            dk = torch.empty((h.shape[1], dz.shape[1]), dtype=torch.float32, device=dz.device)
            dd = torch.empty((xc.shape[1], dh.shape[1]), dtype=torch.float32, device=dz.device)

            side.wait_event(ev)
            with torch.cuda.stream(side):
                D.gemm(h, dz, a_is_km=True, out_dtype=torch.float32, out=dk)   # dK = h^T dz     [p, d]
                D.gemm(xc, dh, a_is_km=True, out_dtype=torch.float32, out=dd)  # dU = x^T dh     [d, p]
                # the casts to the weights' dtype (bf16 variables) belong to this stream too: on the main stream they
                # would read dk / dd with nothing ordering them behind the two products
                dk_out, dd_out = dk.to(k_dt), dd.to(down_dt)
            for t in (h, dz, xc, dh, dk, dd, dk_out, dd_out):
                t.record_stream(side)
            dk, dd = dk_out, dd_out
            _queue_wgrad_sync(task)
        elif low_rank:
            dk, _ = D.gemm(h, dz, a_is_km=True, out_dtype=torch.float32)      # dK = h^T dz     [p, d]
            dh, _ = D.gemm(dz, kc, b_is_nk=True)                               # dh = dz K^T     [B, p]
            dd, _ = D.gemm(xc, dh, a_is_km=True, out_dtype=torch.float32)      # dU = x^T dh     [d, p]
            dx, dx0 = _dx_product(ctx, dh, dc, direct, dx0, x0c, task, u_own)  # dx = dh U^T + direct
        else:
            dk, _ = D.gemm(xc, dz, a_is_km=True, out_dtype=torch.float32)      # dK = x^T dz     [d, d]
            dd = None
            dx, _ = D.gemm(dz, kc, b_is_nk=True, r=direct, beta=1.0)           # dx = dz K^T + direct
        if same:
            gx0, gx = dx.to(x0_dt), None
        else:
            gx, up = dx.to(x_dt), ctx.relay_up
            if up is not None and task != -1:
                # the producer of x takes it from here (its backward runs later in this pass: it is owed dL/dx)
                if up.buf is not None and up.task == task:
                    dx0 = dx0 + up.buf      # x has a second cross-layer consumer that already left its term
                up.buf, up.task = dx0, task
                gx0 = None
            else:
                gx0 = dx0.to(x0_dt)
        return (gx0, gx, None if dd is None else dd.to(down_dt), dk.to(k_dt),
                dbias if has_bias else None, None, None, None, None, None)


# development switch: 0 = every Dense layer runs its own activation backward (krs_dense_act_bwd), as before round 6
FUSE_DENSE_BWD = bool(int(__import__("os").environ.get("KRS_FUSE_DENSE_BWD", "1")))


class DenseActRelay:
    """Hand-off between two STACKED Dense layers (examples/ml_perf/model.py:214-262: `x = dense(x)` repeated): the upper
    layer's data-gradient product dx = dz K^T is the lower layer's dL/dy, and the lower layer's first backward step --
    dz_low = dL/dy * act'(y_low), dbias_low = column sums -- can ride in that product's epilogue (krs_gemm_cross_bwd, dense
    form): no [B, units] matrix is written for dL/dy and read back.
      lower  (act, has_bias) of the layer whose output carries this relay -- set by its forward (no tensor: the output
             holds the relay as an attribute, a reference back would be a cycle that only the cyclic collector frees; the
             upper layer has that output anyway -- it is its saved input);
      fused  (dz_low, its version, dbias_low, graph task) -- set by the upper layer's backward, taken by the lower one's.
    What autograd hands the lower layer is then dz_low itself.  If y had ANOTHER consumer the engine has summed that
    consumer's gradient into it: the lower backward recognises the tensor it was promised (pointer, shape, version) and
    otherwise sends the difference through the derivative (linear in the gradient)."""

    __slots__ = ("lower", "fused", "out_ref")

    def __init__(self):
        self.lower, self.fused, self.out_ref = None, None, None


class DenseFn(torch.autograd.Function):
    """y = act(x @ K + b): the Dense layers of the DLRM bottom / top MLPs
    (examples/ml_perf/model.py:214-262) on krs_gemm with the bias + activation epilogue fused.
    Backward: dz = g * act'(y) written through the output, dK = x^T dz (transposing-read GEMM),
    db = column sum, dx = dz K^T."""

    @staticmethod
    def forward(ctx, x, kernel, bias, act, compute_dtype, relay_up=None, dense_up=None, relay_out=None):
        # relay_up: the Dx0Relay of the cross layer that produced x (layers.Dense reads it off its input): this layer's
        # data-gradient product then runs that layer's elementwise backward in its epilogue (krs_gemm_cross_bwd)
        # dense_up / relay_out: the DenseActRelay of the Dense layer that produced x / of this layer's own output
        xc = x.to(compute_dtype).contiguous()
        kc, kct = D.cast_transpose(kernel, compute_dtype)
        y, _ = D.gemm(xc, kct, b_is_nk=True, bias=bias, act=act)
        ctx.save_for_backward(xc, kc, y if act != L.ACT_NONE else None)
        ctx.meta = (act, bias is not None, x.dtype, kernel.dtype)
        ctx.relay_up = relay_up if ctx.needs_input_grad[0] else None
        # (the lower layer's saved output is this layer's saved input when no cast / copy came between)
        ctx.dense_up = dense_up if (ctx.needs_input_grad[0] and xc.data_ptr() == x.data_ptr() and xc.dtype == x.dtype) else None
        ctx.relay_out = relay_out
        if relay_out is not None:
            relay_out.lower = (act, bias is not None)
        return y

    @staticmethod
    def backward(ctx, g):
        xc, kc, y = ctx.saved_tensors
        act, has_bias, x_dt, k_dt = ctx.meta
        task = torch._C._current_graph_task_id()
        rel, fused = ctx.relay_out, None
        if rel is not None:
            fused, rel.fused = rel.fused, None
            if fused is not None and fused[3] != task:
                fused = None
        if fused is not None:
            # the consumer of y ran this layer's activation backward inside its data-gradient product
            dz_f, version, db = fused[:3]
            if g.data_ptr() == dz_f.data_ptr() and g.shape == dz_f.shape and g._version == version and g.dtype == dz_f.dtype:
                dz = dz_f
            else:
                # y had another consumer: its share of dL/dy arrived summed into dz_f and still has to meet act'(y)
                delta = (g.float() - dz_f.float()).to(dz_f.dtype)
                rest, _ = D.dense_act_bwd(delta, y, act, want_dbias=False)
                dz = (dz_f.float() + (delta if rest is None else rest).float()).to(dz_f.dtype)
                db = D.colsum(dz) if has_bias else None
        else:
            g = g.to(xc.dtype)
            # dz = g * act'(y) and the bias gradient in one pass (krs_dense_act_bwd)
            dz, db = D.dense_act_bwd(g, y, act, want_dbias=has_bias)
        dz = dz.contiguous()
        dk, _ = D.gemm(xc, dz, a_is_km=True, out_dtype=torch.float32)            # [in, units]
        dx = None
        if ctx.needs_input_grad[0]:
            up, task = ctx.relay_up, torch._C._current_graph_task_id()
            if _fusable_below(up, dz, x_dt, task):
                # x is the output of a cross layer: dx = dz K^T is its dL/dy, and its elementwise backward (dz, its
                # term of dL/dx0, bias gradient) rides in the epilogue of this product
                u_low, act_low, _, bias_low, same_low, x0c, (below, x_dt_low) = up.lower
                # (that layer's own term of dL/dx0 waits for ITS data-gradient product when that one is going to be a fused
                #  launch as well: u_upper there, one [B, d] matrix less written here and read there)
                defer = FUSE_TOP_DX0 and not same_low and _fusable_below(below, dz, x_dt_low, task)
                dx, dz_low, dx0, db_low = D.gemm_cross_bwd(dz, kc, None, x0c, u_low, act=act_low, want_dbias=bias_low,
                                                           fold_direct=same_low, want_dx0=not defer)
                if not defer:
                    up.buf, up.task = dx0, task
                up.fused = (dx, dx._version, dz_low, db_low, task, defer)
            elif _dense_fusable_below(ctx.dense_up, dz, x_dt, task):
                # x is the output of a Dense layer: its activation backward and bias gradient ride in this product's epilogue;
                # what goes back to autograd is that layer's dz (its backward recognises it)
                below = ctx.dense_up
                act_low, bias_low = below.lower
                dx, db_low = D.gemm_dense_bwd(dz, kc, xc, act_low, want_dbias=bias_low)     # (xc IS the lower layer's output y)
                below.fused = (dx, dx._version, db_low, task)
            else:
                dx, _ = D.gemm(dz, kc, b_is_nk=True)                               # [B, in]
                dx = dx.to(x_dt)
        return dx, dk.to(k_dt), db, None, None, None, None, None


class BinaryCrossentropyFn(torch.autograd.Function):
    """loss = mean(-(y log p + (1 - y) log(1 - p))), p = clip(pred, eps, 1 - eps): keras.losses.BinaryCrossentropy()
    as examples/ml_perf/main.py:201-210 compiles it.  The gradient is produced by the forward's single pass over the
    predictions (krs_bce_fwd_bwd) and scaled by the incoming scalar in the backward."""

    @staticmethod
    def forward(ctx, pred, labels, epsilon):
        loss, dp = D.bce_fwd_bwd(pred, labels, epsilon, want_grad=ctx.needs_input_grad[0])
        ctx.save_for_backward(dp)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dp,) = ctx.saved_tensors
        return (dp if dp is None else dp * g.to(dp.dtype)), None, None


class CrossEpilogueFn(torch.autograd.Function):
    """y = x0 * (u + diag*x) + x for a host-composed u (arbitrary pre_activation callables)."""

    @staticmethod
    def forward(ctx, u, x0, x, diag_scale):
        u, x0, x = u.contiguous(), x0.contiguous(), x.contiguous()
        ctx.save_for_backward(u, x0, x)
        ctx.diag = diag_scale
        return D.cross_epilogue_fwd(u, x0, x, diag_scale)

    @staticmethod
    def backward(ctx, g):
        u, x0, x = ctx.saved_tensors
        du, dx0, dxd, _ = D.cross_epilogue_bwd(g.contiguous(), u, x0, x, ctx.diag, want_dbias=False)
        return du, dx0, dxd, None


class DotInteractionFn(torch.autograd.Function):
    """DotInteraction.call (dot_interaction.py:170-203) and its gradient.
    relay / n_heads: see SlabGradRelay -- feats[n_heads:] are the feature views of the relay's slab, in order."""

    @staticmethod
    def forward(ctx, self_interaction, skip_gather, relay, n_heads, *feats):
        ctx.save_for_backward(*feats)
        ctx.flags = (self_interaction, skip_gather)
        ctx.relay, ctx.n_heads = relay, n_heads
        # the joined path hands the slab views no gradient of their own: not when somebody watches one of them
        ctx.view_refs = [weakref.ref(t) for t in feats[n_heads:]] if relay is not None else []
        return D.dot_interaction_fwd(feats, self_interaction, skip_gather)

    @staticmethod
    def backward(ctx, g):
        feats = ctx.saved_tensors
        si, sg = ctx.flags
        relay = ctx.relay
        watched = any((t := r()) is not None and (t.retains_grad or t._backward_hooks) for r in ctx.view_refs)
        if relay is not None and relay.buf is not None and not watched:
            buf, task = relay.buf, torch._C._current_graph_task_id()
            n, (batch, dim) = len(feats), feats[0].shape
            if (relay.task == task and task != -1 and tuple(buf.shape) == (batch, n * dim) and buf.is_contiguous()
                    and buf.dtype == feats[0].dtype and g.dtype == buf.dtype):
                relay.buf = None
                SlabGradRelay.joined += 1
                mask = ((1 << n) - 1) & ~((1 << ctx.n_heads) - 1)
                grads = D.dot_interaction_bwd(feats, g, si, sg, into=buf, accumulate_mask=mask)
                return (None, None, None, None, *grads)
        grads = D.dot_interaction_bwd(feats, g, si, sg)
        return (None, None, None, None, *grads)


class EmbedBagFn(torch.autograd.Function):
    """Fused gather+pool over a FusedBags group with the reference's autodiff semantics:
    dense [V, D] table gradients, computed by the sort-based K2.
    Returns one [B, D] tensor per feature (column views of one fused buffer)."""

    @staticmethod
    def forward(ctx, bags, ids, batch, hots, offsets, weights, out_dtype, check_ids, *tables):
        # `tables` are passed explicitly so autograd tracks them; `bags.tables` hold the same storage.
        # check_ids: True = read the kernel's error word now (one host sync; EmbedReduce does that);
        # a device int32[1] tensor = let the kernel OR its bits into it and return (the caller reads it
        # later: DistributedEmbedding.check_ids); False / None = no report.
        lazy = isinstance(check_ids, torch.Tensor)
        err = check_ids if lazy else (torch.zeros(1, dtype=torch.int32, device=ids.device) if check_ids else None)
        out, scale = bags.forward(ids, batch, hots=hots, offsets=offsets, weights=weights, out_dtype=out_dtype,
                                  want_scale=any(comb != L.SUM for _, comb, _ in bags.features), err_flag=err)
        if check_ids is True and int(err.item()) & L.FLAG_ID_OUT_OF_RANGE:
            raise IndexError("embedding id out of range for its table (ids are never clamped)")
        ctx.bags, ctx.batch, ctx.hots = bags, batch, hots
        ctx.save_for_backward(ids, offsets, weights, scale)
        ctx.table_dtypes = [t.dtype for t in tables]
        ctx.out_meta = (out.dtype, out.device)
        return _split_columns(out, len(bags.features), bags.dim)

    @staticmethod
    def backward(ctx, *gs):
        ids, offsets, weights, scale = ctx.saved_tensors
        bags = ctx.bags
        g = _gather_feature_grads(gs, ctx.batch, bags.dim, *ctx.out_meta)
        ws = bags.plan_backward(ids, ctx.batch, hots=ctx.hots, offsets=offsets, global_order=False)
        grads = bags.backward_dense(ws, g, ctx.batch, ids.numel(), hots=ctx.hots, weights=weights,
                                    bag_scale=scale)
        grads = [gr.to(dt) for gr, dt in zip(grads, ctx.table_dtypes)]
        return (None, None, None, None, None, None, None, None, *grads)


# development switch (A/B of the two orders, see EmbedBagFusedFn.forward)
_PLAN_AFTER_GATHER = bool(int(__import__("os").environ.get("KRS_PLAN_AFTER_GATHER", "0")))


class EmbedBagFusedFn(torch.autograd.Function):
    """Fused gather+pool whose backward applies the per-table optimizer to the touched rows
    in place (the SparseCore-style path: jax/embedding_lookup.py:174-273 returns updated
    tables instead of gradients).  `anchor` is a dummy scalar that requires grad so that the
    backward runs; it receives a zero gradient."""

    @staticmethod
    def forward(ctx, bags, ids, batch, hots, offsets, weights, out_dtype, optimizer, anchor, lead=0, err_flag=None):
        # outputs: the whole slab [B, lead + n*dim] (columns 0..lead are left for the caller, see
        # layers.concat_features) followed by the per-feature column views
        ctx.set_materialize_grads(False)  # unused outputs (slab or views) arrive as None, not as zero tensors
        n = len(bags.features)
        slab = torch.empty((batch, lead + n * bags.dim), dtype=out_dtype or bags.dtype, device=ids.device)
        # The backward's plan (sort of the lookups by row) depends only on the ids: start it now on a side
        # stream, BEFORE the gather is enqueued, so that the two run side by side -- the gather is bound by
        # HBM, the sort's scatter passes by LDS ranking work -- and the plan is done when the dense part starts
        # (queued behind the gather it ran under the first FeatureCross GEMMs and slowed them by 0.4 ms).
        ctx.plan = None
        ctx.owns_plan_ws = False
        plan_first = not _PLAN_AFTER_GATHER
        # the per-bag combiner scale (1 / sum w, 1 / sqrt(sum w^2)) is what the backward multiplies by: all ones for
        # "sum" bags, which then neither write it here nor gather it per lookup there
        want_scale = any(comb != L.SUM for _, comb, _ in bags.features)
        if not plan_first:
            out, scale = bags.forward(ids, batch, hots=hots, offsets=offsets, weights=weights,
                                      out=slab[:, lead:], want_scale=want_scale, err_flag=err_flag)
        if ctx.needs_input_grad[8]:  # a backward will follow (not under torch.no_grad())
            main = torch.cuda.current_stream()
            side = _side_stream(ids.device)
            # The descriptor blocks are uploaded (and cached) by whoever asks first: do that HERE, on the main
            # stream, so that the copy is ordered before both users -- the plan (side waits for main below) and
            # the gather (main).  Uploaded on the side stream by a cold cache, the gather on main would have read
            # them with nothing ordering it behind the copy (first step, first-seen batch size, after .to()).
            bags.table_desc()
            bags.feature_desc(batch, hots, ids.device)
            side.wait_stream(main)
            # The plan's workspace (0.36 GB at C3) lives on the bags across steps.  Allocated per step inside the side
            # stream's context it could only be re-used once the DEVICE had passed the step that used it (record_stream
            # below), and a host that enqueues a step in 2 ms against 10 ms of device time gets ever further ahead: a
            # fresh hipMalloc every other step (1 - 11 ms of host time each, box-dependent: one bench leg in three ran
            # host-bound at 21 ms per step).  Safe: the side stream has just waited for everything the main stream holds,
            # including the previous step's apply; a second forward before that step's backward gets its own tensor.
            keep = None if getattr(bags, "_plan_ws_busy", False) else getattr(bags, "_plan_ws", None)
            with torch.cuda.stream(side):
                ws = bags.plan_backward(ids, batch, hots=hots, offsets=offsets, global_order=False, ws=keep)
                done = torch.cuda.Event()
                done.record(side)
            if not getattr(bags, "_plan_ws_busy", False):
                bags._plan_ws, bags._plan_ws_busy = ws, True
                ctx.owns_plan_ws = [True]
                weakref.finalize(ctx, _release_plan_ws, weakref.ref(bags), ctx.owns_plan_ws)
            for t in (ws, ids, offsets):
                if t is not None:
                    t.record_stream(side)
            ctx.plan = (ws, done)
        # err_flag: device int32[1] the kernel ORs KRS_FLAG_* into (out-of-range ids contribute nothing and are
        # never clamped); read later by the layer, so the step keeps running without a host sync
        if plan_first:
            out, scale = bags.forward(ids, batch, hots=hots, offsets=offsets, weights=weights,
                                      out=slab[:, lead:], want_scale=want_scale, err_flag=err_flag)
        ctx.bags, ctx.batch, ctx.hots, ctx.optimizer, ctx.lead = bags, batch, hots, optimizer, lead
        ctx.save_for_backward(ids, offsets, weights, scale)
        ctx.out_meta = (out.dtype, out.device)
        return (slab,) + _split_columns(slab, n, bags.dim, lead)

    @staticmethod
    def backward(ctx, g_slab, *gs):
        ids, offsets, weights, scale = ctx.saved_tensors
        bags = ctx.bags
        g = _sum_slab_and_feature_grads(g_slab, gs, ctx.lead, ctx.batch, len(bags.features), bags.dim,
                                        *ctx.out_meta)
        if ctx.plan is not None:
            ws, done = ctx.plan
            torch.cuda.current_stream().wait_event(done)
            ws.record_stream(torch.cuda.current_stream())
        else:
            ws = bags.plan_backward(ids, ctx.batch, hots=ctx.hots, offsets=offsets, global_order=False)
        opt = ctx.optimizer  # a kind string, or an object with .fused_kind / .next_hyper() (the layer's group)
        kind = opt if isinstance(opt, str) else opt.fused_kind
        hyper = None if isinstance(opt, str) else opt.next_hyper()   # also refreshes scheduled learning rates
        bags.backward_fused(kind, ws, g, ctx.batch, ids.numel(), hots=ctx.hots, weights=weights,
                            bag_scale=scale, hyper=hyper)
        if ctx.owns_plan_ws:
            # (the next forward's plan is ordered behind this apply: it may take the buffer)
            _release_plan_ws(weakref.ref(bags), ctx.owns_plan_ws)
        # (no gradient for the anchor -- it exists only to make autograd call this function; a zero tensor here cost a fill and an
        #  accumulate launch per step)
        return (None,) * 11


class SlabFillFn(torch.autograd.Function):
    """concat([*heads, slab[:, lead:]]) without the copy of the slab: the heads (lead columns in
    total) are written into the slab's reserved leading columns and the slab itself is the result.
    The write goes through `.data`, so tensors that saved views of the slab keep their version."""

    @staticmethod
    def forward(ctx, slab, relay, *heads):
        col = 0
        ctx.cols = []
        ctx.relay = relay
        with torch.no_grad():
            for h in heads:
                w = h.shape[1]
                slab.data[:, col:col + w].copy_(h)
                ctx.cols.append((col, col + w))
                col += w
        return slab.data.as_strided(slab.shape, slab.stride(), slab.storage_offset())

    @staticmethod
    def backward(ctx, g):
        relay = ctx.relay
        if relay is not None:
            # (a concat result nobody holds any more cannot show its .grad to anyone)
            out = relay.out_ref() if relay.out_ref is not None else None
            plain = (out is None or (not out.retains_grad and not out._backward_hooks)) and not relay.disabled
            task = torch._C._current_graph_task_id()
            relay.buf, relay.task = (g, task) if (plain and task != -1 and g.is_contiguous()) else (None, -1)
        return (g, None) + tuple(g[:, a:b] for a, b in ctx.cols)
