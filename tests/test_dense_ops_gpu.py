"""K3 / K4 / K5 parity: krs_gemm (+ fused cross epilogue), cross elementwise kernels,
DotInteraction fwd/bwd and MOD bucketise vs the CPU oracle, through the C ABI."""

import json
import os

import numpy as np
import pytest
import torch

from oracle import krs_oracle as ko
from tests.helpers import to_f32, to_np

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat.json")))


def _t(a, dt=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dt).to(DEV)


def _tol(dt, k):
    # bf16: inputs exact, fp32 accumulate (different order than the oracle), one rounding of the result
    if dt == torch.bfloat16:
        return dict(rtol=2 ** -6, atol=2e-2 * max(1.0, k ** 0.5 / 8))
    return dict(rtol=2e-5, atol=2e-5 * max(1.0, k ** 0.5 / 4))  # north-star: 1e-5 class for fp32


@pytest.mark.parametrize("m,n,k", [(1, 3, 3), (5, 7, 9), (128, 128, 64), (130, 200, 72), (256, 512, 512),
                                   (300, 136, 1000), (64, 3456, 512), (130, 200, 1088), (257, 512, 3456),
                                   (700, 300, 1024), (512, 768, 2048),  # the last three: 256x256-tile kernel
                                   (13, 520, 4100), (264, 1, 4100), (16, 16, 9000),  # tn: thin weight-gradient kernel
                                   (3, 700, 1030), (8, 300, 2048), (520, 5, 1500),      # ... its 4- / 8-wide builds
                                   (2048, 1, 256), (1500, 3, 77), (1100, 8, 40),        # nn / nt: gemm_rowdot_kernel
                                   (2048, 256, 1), (300, 512, 5), (5000, 16, 16)])      # nn / nt: gemm_smallk_kernel
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("layout", ["nn", "nt", "tn"])
def test_gemm_layouts(m, n, k, dt, layout):
    from keras_rs_amd import dense_ops as D

    rng = np.random.default_rng(m * 31 + n * 7 + k)
    a = _t(rng.uniform(-1, 1, (m, k)), dt)
    b = _t(rng.uniform(-1, 1, (k, n)), dt)
    a_in = a.t().contiguous() if layout == "tn" else a
    b_in = b.t().contiguous() if layout == "nt" else b
    c, _ = D.gemm(a_in, b_in, a_is_km=layout == "tn", b_is_nk=layout == "nt")
    exp, _ = ko.gemm(to_np(a), to_np(b), m, n, k)
    np.testing.assert_allclose(to_f32(to_np(c)), to_f32(exp), **_tol(dt, k))


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("act", [None, "relu", "sigmoid", "tanh"])
def test_gemm_tiny_dimension_epilogues(dt, act):
    # the last Dense of the DLRM top MLP (256 -> 1 unit): y = sigmoid(x k + b) on gemm_rowdot_kernel, and the data
    # gradient dz k^T (K = 1) on gemm_smallk_kernel
    from keras_rs_amd import dense_ops as D

    rng = np.random.default_rng(3)
    x = _t(rng.uniform(-1, 1, (3000, 256)), dt)
    kt = _t(rng.uniform(-0.2, 0.2, (1, 256)), dt)          # [units, in]: given K-contiguous
    bias = _t(rng.uniform(-0.5, 0.5, (1,)))
    y, _ = D.gemm(x, kt, b_is_nk=True, bias=bias, act=D.ACTS[act])
    ey, _ = ko.gemm(to_np(x), to_np(kt), 3000, 1, 256, b_is_nk=True, bias=to_np(bias), act=act)
    np.testing.assert_allclose(to_f32(to_np(y)), to_f32(ey), **_tol(dt, 256))
    dz = _t(rng.uniform(-1, 1, (3000, 1)), dt)
    k = _t(rng.uniform(-0.2, 0.2, (256, 1)), dt)           # [in, units] = [N, K]
    dx, _ = D.gemm(dz, k, b_is_nk=True)
    edx, _ = ko.gemm(to_np(dz), to_np(k), 3000, 256, 1, b_is_nk=True)
    np.testing.assert_allclose(to_f32(to_np(dx)), to_f32(edx), **_tol(dt, 1))


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("act", [None, "relu", "sigmoid", "tanh"])
@pytest.mark.parametrize("m,d,p", [(64, 128, 128), (200, 256, 64), (33, 24, 24), (200, 256, 1024), (300, 520, 1024)])
def test_gemm_cross_epilogue(dt, act, m, d, p):
    from keras_rs_amd import dense_ops as D

    rng = np.random.default_rng(7)
    h = _t(rng.uniform(-1, 1, (m, p)), dt)
    kern = _t(rng.uniform(-0.2, 0.2, (p, d)), dt)
    bias = _t(rng.uniform(-0.5, 0.5, (d,)))
    x0 = _t(rng.uniform(-1, 1, (m, d)), dt)
    x = _t(rng.uniform(-1, 1, (m, d)), dt)
    y, u = D.gemm(h, kern, bias=bias, act=D.ACTS[act], diag_scale=0.3, x0=x0, x=x, want_u=True)
    ey, eu = ko.gemm(to_np(h), to_np(kern), m, d, p, bias=to_np(bias), act=act, diag_scale=0.3,
                     x0=to_np(x0), x=to_np(x), want_u=True)
    np.testing.assert_allclose(to_f32(to_np(y)), to_f32(ey), **_tol(dt, p))
    np.testing.assert_allclose(to_f32(to_np(u)), to_f32(eu), **_tol(dt, p))


def test_gemm_residual_and_fp32_out_from_bf16_and_splitk():
    from keras_rs_amd import dense_ops as D

    rng = np.random.default_rng(11)
    # weight-gradient shape: contraction over a long batch axis -> split-K slabs
    B, p, d = 8192, 128, 256
    h = _t(rng.uniform(-1, 1, (B, p)), torch.bfloat16)
    dz = _t(rng.uniform(-1, 1, (B, d)), torch.bfloat16)
    dw, _ = D.gemm(h, dz, a_is_km=True, out_dtype=torch.float32)
    exp = to_f32(to_np(h)).astype(np.float64).T @ to_f32(to_np(dz)).astype(np.float64)
    np.testing.assert_allclose(dw.cpu().numpy(), exp, rtol=1e-4, atol=5e-3)
    dw2, _ = D.gemm(h, dz, a_is_km=True, out_dtype=torch.float32)
    assert torch.equal(dw, dw2)  # slab reduction is deterministic
    # residual epilogue: C = A@B + beta*R
    a = _t(rng.uniform(-1, 1, (96, 64)))
    b = _t(rng.uniform(-1, 1, (64, 40)))
    r = _t(rng.uniform(-1, 1, (96, 40)))
    c, _ = D.gemm(a, b, r=r, beta=0.5)
    np.testing.assert_allclose(c.cpu().numpy(), (a @ b + 0.5 * r).cpu().numpy(), rtol=2e-5, atol=2e-5)
    # the same through the 256x256-tile kernel (bf16, K-contiguous operands, vector residual epilogue)
    a = _t(rng.uniform(-1, 1, (520, 1024)), torch.bfloat16)
    bt = _t(rng.uniform(-1, 1, (264, 1024)), torch.bfloat16)
    r = _t(rng.uniform(-1, 1, (520, 264)), torch.bfloat16)
    c, _ = D.gemm(a, bt, b_is_nk=True, r=r, beta=1.0)
    exp = a.float() @ bt.float().t() + r.float()
    np.testing.assert_allclose(c.float().cpu().numpy(), exp.cpu().numpy(), rtol=2 ** -6, atol=0.1)


@pytest.mark.parametrize("case", KAT["feature_cross"]["cases"], ids=lambda c: c["name"])
def test_feature_cross_kat_through_hip(case):
    from keras_rs_amd import dense_ops as D

    fc = KAT["feature_cross"]
    x0 = _t(fc["x0"])
    x = x0 if case["one_input"] else _t(fc["x"])
    d, p = 3, case["projection_dim"]
    h = x
    if p is not None:
        h, _ = D.gemm(x, torch.ones(d, p, device=DEV))
    y, _ = D.gemm(h, torch.ones(h.shape[1], d, device=DEV), bias=torch.zeros(d, device=DEV),
                  diag_scale=case["diag_scale"], x0=x0, x=x)
    np.testing.assert_allclose(y.cpu().numpy(), np.array(case["expected"], np.float32), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("m,n", [(37, 24), (64, 128), (5, 3)])
@pytest.mark.parametrize("act", [None, "relu", "sigmoid", "tanh"])
def test_cross_elementwise_fwd_bwd(dt, m, n, act):
    from keras_rs_amd import dense_ops as D

    rng = np.random.default_rng(3)
    g, u, x0, x = (_t(rng.uniform(-1, 1, (m, n)), dt) for _ in range(4))
    y = D.cross_epilogue_fwd(u, x0, x, 0.25)
    # elementwise fp32: the GPU contracts a*b+c into fma, the oracle does not -> 1 ulp
    tol = dict(rtol=2 ** -7, atol=1e-6) if dt == torch.bfloat16 else dict(rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(to_f32(to_np(y)), to_f32(ko.cross_epilogue_fwd(to_np(u), to_np(x0), to_np(x), 0.25)), **tol)
    acc = _t(rng.uniform(-1, 1, (m, n)), dt)
    acc0 = acc.clone()
    du, dx0, dxd, dbias = D.cross_epilogue_bwd(g, u, x0, x, 0.25, act=D.ACTS[act], dx0_into=acc)
    edu, edx0, edxd, edb = ko.cross_epilogue_bwd(to_np(g), to_np(u), to_np(x0), to_np(x), 0.25,
                                                 dx0_init=to_np(acc0), act=act)
    np.testing.assert_allclose(to_f32(to_np(du)), to_f32(edu), **tol)
    np.testing.assert_allclose(to_f32(to_np(dx0)), to_f32(edx0), **tol)
    np.testing.assert_allclose(to_f32(to_np(dxd)), to_f32(edxd), **tol)
    np.testing.assert_allclose(dbias.cpu().numpy(), edb, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(D.colsum(g).cpu().numpy(), ko.colsum(to_np(g)), rtol=1e-5, atol=1e-5)
    # x is x0: dxd aliases dx0 and the buffer receives the sum of both terms
    _, fx0, fxd, _ = D.cross_epilogue_bwd(g, u, x0, x0, 0.25, act=D.ACTS[act], fold_direct=True, want_dbias=False)
    assert fxd is fx0
    _, efx0, _, _ = ko.cross_epilogue_bwd(to_np(g), to_np(u), to_np(x0), to_np(x0), 0.25, act=act, fold_direct=True)
    np.testing.assert_allclose(to_f32(to_np(fx0)), to_f32(efx0), **tol)
    _, sx0, sxd, _ = ko.cross_epilogue_bwd(to_np(g), to_np(u), to_np(x0), to_np(x0), 0.25, act=act)
    ftol = dict(rtol=2 ** -6, atol=2e-2) if dt == torch.bfloat16 else dict(rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(to_f32(efx0), to_f32(sx0) + to_f32(sxd), **ftol)


@pytest.mark.parametrize("case", KAT["dot_interaction"]["cases"],
                         ids=lambda c: f"self{int(c['self_interaction'])}_skip{int(c['skip_gather'])}")
def test_dot_interaction_kat_through_hip(case):
    from keras_rs_amd import dense_ops as D

    feats = [_t(f) for f in KAT["dot_interaction"]["inputs"]]
    out = D.dot_interaction_fwd(feats, case["self_interaction"], case["skip_gather"])
    np.testing.assert_allclose(out.cpu().numpy(), np.array(case["expected"], np.float32), rtol=1e-6, atol=1e-5)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("F,D_,B", [(3, 5, 4), (27, 128, 70), (8, 64, 33), (32, 16, 9), (40, 8, 5), (4, 256, 6), (32, 128, 7),
                                    (2, 32, 300), (17, 96, 2)])
@pytest.mark.parametrize("si,sg", [(False, False), (True, False), (False, True), (True, True)])
def test_dot_interaction_fwd_bwd(dt, F, D_, B, si, sg):
    from keras_rs_amd import dense_ops as D

    rng = np.random.default_rng(F * 100 + D_)
    # features as column views of one concat buffer (the ml_perf layout)
    buf = _t(rng.uniform(-1, 1, (B, F * D_)), dt)
    feats = [buf[:, f * D_:(f + 1) * D_] for f in range(F)]
    out = D.dot_interaction_fwd(feats, si, sg)
    fn = [np.ascontiguousarray(to_np(buf)[:, f * D_:(f + 1) * D_]) for f in range(F)]
    exp = ko.dot_interaction_fwd(fn, si, sg)
    tol = dict(rtol=2 ** -6, atol=3e-2) if dt == torch.bfloat16 else dict(rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(to_f32(to_np(out)), to_f32(exp), **tol)
    g = _t(rng.uniform(-1, 1, tuple(out.shape)), dt)
    grads = D.dot_interaction_bwd(feats, g, si, sg)
    eg = ko.dot_interaction_bwd(fn, to_np(g), si, sg)
    tol = dict(rtol=2 ** -6, atol=6e-2) if dt == torch.bfloat16 else dict(rtol=1e-5, atol=2e-5)
    for a, e in zip(grads, eg):
        np.testing.assert_allclose(to_f32(to_np(a)), to_f32(e), **tol)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("F,D_,B,mask", [(27, 128, 70, 0x7FFFFFE), (27, 128, 5, 0x7FFFFFF), (8, 64, 33, 0b10110100),
                                         (5, 6, 9, 0b11110), (40, 8, 5, (1 << 40) - 2), (3, 256, 4, 0b100)])
def test_dot_interaction_bwd_accumulate(dt, F, D_, B, mask):
    # krs_dot_interaction_bwd_accumulate: masked features are added to what `into` holds (fp32 sum, one
    # rounding), the others come back as fresh tensors; fast (F <= 32, even D) and generic kernels
    from keras_rs_amd import dense_ops as D

    rng = np.random.default_rng(F * 7 + D_)
    buf = _t(rng.uniform(-1, 1, (B, F * D_)), dt)
    feats = [buf[:, f * D_:(f + 1) * D_] for f in range(F)]
    g = _t(rng.uniform(-1, 1, (B, F * (F - 1) // 2)), dt)
    into = _t(rng.uniform(-2, 2, (B, F * D_)), dt)
    before = to_np(into).copy()
    fn = [np.ascontiguousarray(to_np(buf)[:, f * D_:(f + 1) * D_]) for f in range(F)]
    existing = [np.ascontiguousarray(before[:, f * D_:(f + 1) * D_]) for f in range(F)]
    exp = ko.dot_interaction_bwd(fn, to_np(g), False, False, existing=existing, accumulate_mask=mask)
    grads = D.dot_interaction_bwd(feats, g, False, False, into=into, accumulate_mask=mask)
    after = to_np(into)
    tol = dict(rtol=2 ** -6, atol=6e-2) if dt == torch.bfloat16 else dict(rtol=1e-5, atol=2e-5)
    for f in range(F):
        sl = slice(f * D_, (f + 1) * D_)
        if (mask >> f) & 1:
            assert grads[f] is None
            np.testing.assert_allclose(to_f32(after[:, sl]), to_f32(exp[f]), **tol)
        else:
            np.testing.assert_allclose(to_f32(to_np(grads[f])), to_f32(exp[f]), **tol)
            assert np.array_equal(after[:, sl], before[:, sl])      # untouched


@pytest.mark.parametrize("src,dst", [(torch.float32, torch.bfloat16), (torch.float32, torch.float32),
                                     (torch.bfloat16, torch.bfloat16), (torch.bfloat16, torch.float32)])
@pytest.mark.parametrize("rows,cols", [(3456, 512), (512, 3456), (13, 7), (1, 129), (64, 64), (200, 1)])
def test_cast_transpose_is_bit_exact(src, dst, rows, cols):
    from keras_rs_amd import dense_ops as D

    rng = np.random.default_rng(rows * 31 + cols)
    w = _t(rng.uniform(-3, 3, (rows, cols)), src)
    plain, wt = D.cast_transpose(w, dst)
    e_plain, e_t = ko.cast_transpose(to_np(w), dst == torch.bfloat16)
    assert np.array_equal(to_np(plain), e_plain) and np.array_equal(to_np(wt), e_t)
    if src == dst:
        assert plain is w                      # nothing copied
    # a strided source (a column window of a wider matrix) and a single output
    wide = _t(rng.uniform(-3, 3, (rows, cols + 5)), src)
    _, wt2 = D.cast_transpose(wide[:, 2:2 + cols], dst, want_plain=False)
    assert np.array_equal(to_np(wt2), ko.cast_transpose(np.ascontiguousarray(to_np(wide)[:, 2:2 + cols]), dst == torch.bfloat16)[1])


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("m,n", [(300, 1024), (65, 7), (1, 8), (4099, 256), (33, 1)])
@pytest.mark.parametrize("act", ["relu", "sigmoid", "tanh", "none"])
def test_dense_act_bwd(dt, m, n, act):
    from keras_rs_amd import _lib as KL
    from keras_rs_amd import dense_ops as D

    rng = np.random.default_rng(m * 13 + n)
    g = _t(rng.uniform(-1, 1, (m, n)), dt)
    y = _t(rng.uniform(-1, 1, (m, n)) if act != "sigmoid" else rng.uniform(0, 1, (m, n)), dt)
    code = {"relu": KL.ACT_RELU, "sigmoid": KL.ACT_SIGMOID, "tanh": KL.ACT_TANH, "none": KL.ACT_NONE}[act]
    dz, db = D.dense_act_bwd(g, None if act == "none" else y, code)
    edz, edb = ko.dense_act_bwd(to_np(g), to_np(y), act)
    if act in ("relu", "none"):
        assert np.array_equal(to_np(dz), edz)                               # one multiply, one rounding: bit-exact
    else:   # 1 - y*y / y*(1 - y) may be contracted into a fused multiply-add on the device
        np.testing.assert_allclose(to_f32(to_np(dz)), to_f32(edz), rtol=2 ** -7 if dt == torch.bfloat16 else 1e-6, atol=1e-7)
    np.testing.assert_allclose(db.cpu().numpy(), edb, rtol=1e-5, atol=1e-4 * max(1.0, m / 100))
    if act == "none":
        assert dz.data_ptr() == g.data_ptr()                                # nothing copied


def test_dense_adagrad_one_launch_matches_torch_and_the_oracle():
    # keras_rs_amd.optim.Adagrad (krs_dense_adagrad): a list of weights of odd sizes (vector path, scalar tails, an
    # unaligned view, > 32 tensors = two launches) against torch.optim.Adagrad and the numpy restatement, 3 steps
    from keras_rs_amd.optim import Adagrad

    rng = np.random.default_rng(3)
    shapes = [(3456, 512), (512, 3456), (3456,), (7, 5), (1,), (4099,)] + [(17, 3)] * 30
    base = [torch.tensor(rng.uniform(-1, 1, s).astype(np.float32), device=DEV) for s in shapes]
    big = torch.zeros(1000 + 3, device=DEV)
    mine = [b.clone().requires_grad_(True) for b in base] + [big[3:].detach().requires_grad_(True)]   # 12-byte offset
    ref = [b.clone().requires_grad_(True) for b in base] + [torch.zeros(1000, device=DEV, requires_grad=True)]
    o1 = Adagrad(mine, lr=0.0034, initial_accumulator_value=0.1)
    o2 = torch.optim.Adagrad(ref, lr=0.0034, initial_accumulator_value=0.1, foreach=True)
    pn = [to_np(b) for b in base] + [np.zeros(1000, np.float32)]
    an = [np.full_like(x, 0.1) for x in pn]
    for step in range(3):
        gs = [torch.tensor(rng.uniform(-1, 1, tuple(p.shape)).astype(np.float32) * 10.0 ** -step, device=DEV) for p in mine]
        for p, q, g in zip(mine, ref, gs):
            p.grad, q.grad = g.clone(), g.clone()
        o1.step()
        o2.step()
        for i, g in enumerate(gs):
            pn[i], an[i] = ko.dense_adagrad(pn[i], to_np(g), an[i], 0.0034, 1e-10)
    for p, q, e, ea in zip(mine, ref, pn, an):
        torch.testing.assert_close(p.detach(), q.detach(), rtol=2e-6, atol=1e-7)
        np.testing.assert_allclose(to_np(p.detach()), e, rtol=2e-6, atol=1e-7)
        np.testing.assert_allclose(to_np(o1.state[p]["sum"]), ea, rtol=1e-6, atol=0)
    with pytest.raises(Exception):
        bad = torch.zeros(4, device=DEV, dtype=torch.bfloat16, requires_grad=True)
        bad.grad = torch.zeros_like(bad)
        Adagrad([bad]).step()


@pytest.mark.parametrize("n_shards", [1, 2, 4, 8, 5])
@pytest.mark.parametrize("idt", [np.int32, np.int64])
@pytest.mark.parametrize("nnz", [0, 1, 255, 2048, 100_003])
def test_mod_bucketize_bit_exact(n_shards, idt, nnz):
    from keras_rs_amd import dense_ops as D

    rng = np.random.default_rng(nnz + n_shards)
    ids = rng.integers(0, 1_000_000, size=nnz).astype(idt)
    local, perm, counts = D.mod_bucketize(torch.from_numpy(ids).to(DEV), n_shards)
    el, ep, ec = ko.mod_bucketize(ids, n_shards)
    np.testing.assert_array_equal(counts.cpu().numpy(), ec)
    np.testing.assert_array_equal(perm.cpu().numpy(), ep)
    np.testing.assert_array_equal(local.cpu().numpy(), el)


@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
@pytest.mark.parametrize("n", [1, 1000, 65536])
def test_binary_crossentropy_fwd_bwd_matches_oracle(dtype, n):
    """krs_bce_fwd_bwd (keras.losses.BinaryCrossentropy() of examples/ml_perf/main.py:201-210) against the oracle:
    loss to 2e-6 (fp32 tree sum vs float64 sum), gradient element-wise, predictions beyond the clip (gradient 0),
    exact 0 / 1 predictions (finite loss), through autograd with an upstream scale."""
    import keras_rs_amd.layers as kl
    from keras_rs_amd import dense_ops as D

    rng = np.random.default_rng(n)
    pred = rng.uniform(0, 1, n).astype(np.float32)
    pred[:: 97] = 0.0
    pred[1:: 89] = 1.0
    pred[2:: 83] = 5e-8
    y = (rng.uniform(0, 1, n) < 0.3).astype(np.float32)
    t = torch.from_numpy(pred).to(DEV).to(getattr(torch, dtype))
    labels = torch.from_numpy(y).to(DEV)
    loss, dp = D.bce_fwd_bwd(t, labels, grad_scale=1.0)
    e_loss, e_dp = ko.bce_fwd_bwd(to_np(t), y)
    assert np.isfinite(float(loss))
    np.testing.assert_allclose(float(loss), float(e_loss), rtol=2e-6)
    if dtype == "float32":
        np.testing.assert_allclose(to_np(dp), e_dp, rtol=1e-6, atol=1e-12)
    else:
        np.testing.assert_allclose(ko.bf16_bits_to_f32(to_np(dp)), ko.bf16_bits_to_f32(e_dp), rtol=2 ** -7, atol=1e-12)
    # through autograd: d(3 * loss)/dpred = 3 * dpred
    tp = t.clone().reshape(n, 1).requires_grad_()
    (3.0 * kl.BinaryCrossentropy()(labels.reshape(n, 1), tp)).backward()
    np.testing.assert_allclose(tp.grad.float().cpu().numpy().reshape(-1), 3.0 * dp.float().cpu().numpy(), rtol=2 ** -6, atol=1e-12)


@pytest.mark.parametrize("n_feats,dim", [(65, 16), (100, 8), (130, 4)])
@pytest.mark.parametrize("self_interaction,skip_gather", [(False, False), (True, False), (False, True), (True, True)])
@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
def test_dot_interaction_beyond_64_features(n_feats, dim, self_interaction, skip_gather, dtype):
    """The reference puts no limit on the number of features (dot_interaction.py:134-205): F = 65, 100 (one chunk of j)
    and 130 (two chunks) through the block launches, forward and backward against the oracle in the four modes."""
    from keras_rs_amd import dense_ops as D

    rng = np.random.default_rng(n_feats)
    B = 37
    tdt = getattr(torch, dtype)
    feats = [_t(rng.uniform(-1, 1, (B, dim)), tdt) for _ in range(n_feats)]
    fnp = [to_np(f) for f in feats]
    out = D.dot_interaction_fwd(feats, self_interaction, skip_gather)
    exp = ko.dot_interaction_fwd(fnp, self_interaction, skip_gather)
    tol = _tol(tdt, dim)
    np.testing.assert_allclose(to_f32(to_np(out)), to_f32(exp), **tol)
    g = _t(rng.uniform(-1, 1, tuple(out.shape)), tdt)
    grads = D.dot_interaction_bwd(feats, g, self_interaction, skip_gather)
    e_grads = ko.dot_interaction_bwd(fnp, to_np(g), self_interaction, skip_gather)
    btol = _tol(tdt, 2 * n_feats)
    if dtype == "bfloat16" and n_feats > 128:
        btol = dict(rtol=2 ** -5, atol=btol["atol"] * 2)      # one more rounding per chunk of 128 features
    for a, b in zip(grads, e_grads):
        np.testing.assert_allclose(to_f32(to_np(a)), to_f32(b), **btol)


def test_dot_interaction_layer_with_100_features_trains():
    import keras_rs_amd.layers as kl

    rng = np.random.default_rng(0)
    feats = [_t(rng.uniform(-1, 1, (8, 4))).requires_grad_() for _ in range(100)]
    out = kl.DotInteraction()(feats)
    assert tuple(out.shape) == (8, 100 * 99 // 2)
    out.sum().backward()
    x = torch.stack([f.detach() for f in feats], 1)                      # dX_i = sum_{j != i} X_j for an all-ones gradient
    exp = x.sum(1, keepdim=True) - x
    for i, f in enumerate(feats):
        np.testing.assert_allclose(f.grad.cpu().numpy(), exp[:, i].cpu().numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("m,n", [(20000, 3456), (8192, 13), (4097, 250), (3, 8)])
def test_bias_gradients_are_two_stage_sums_identical_from_run_to_run(dt, m, n):
    """krs_colsum_workspace_bytes: with the workspace the three column sums of the path (cross bias, Dense bias, colsum)
    are formed per row group and then over the groups IN ORDER -- the same bits on every run -- and agree with the
    workspace-free form of the C ABI (fp32 atomics over the same groups) to rounding; a short workspace is refused."""
    import ctypes as C

    from keras_rs_amd import _lib as L
    from keras_rs_amd import dense_ops as D

    rng = np.random.default_rng(5)
    g, u, x0, x = (_t(rng.uniform(-1, 1, (m, n)), dt) for _ in range(4))
    runs = [(D.cross_epilogue_bwd(g, u, x0, x, 0.0, act=D.ACTS["relu"])[3], D.dense_act_bwd(g, u, D.ACTS["tanh"])[1],
             D.colsum(g)) for _ in range(3)]
    for later in runs[1:]:
        for a, b in zip(runs[0], later):
            assert torch.equal(a, b)
    exp = (to_f32(to_np(g)).astype(np.float64) * to_f32(to_np(x0)) * (to_f32(to_np(u)) > 0)).sum(0)
    scale = np.abs(exp).max() + 1.0
    np.testing.assert_allclose(runs[0][0].cpu().numpy(), exp, rtol=0, atol=2e-6 * scale * np.sqrt(m))
    np.testing.assert_allclose(runs[0][2].cpu().numpy(), to_f32(to_np(g)).astype(np.float64).sum(0), rtol=0,
                               atol=2e-6 * scale * np.sqrt(m))
    # workspace-free form of the ABI (atomics): same groups, other order
    nbytes = L.lib().krs_colsum_workspace_bytes(C.c_int64(m), C.c_int64(n))
    assert nbytes >= 4 * n
    out = torch.empty(n, dtype=torch.float32, device=DEV)
    rc = L.lib().krs_colsum(L.ptr(g), C.c_int64(n), C.c_int64(m), C.c_int64(n), C.c_int(L.fdtype(g)), L.ptr(out), None,
                            C.c_size_t(0), L.stream_ptr())
    assert rc == 0
    np.testing.assert_allclose(out.cpu().numpy(), runs[0][2].cpu().numpy(), rtol=0, atol=1e-6 * scale * np.sqrt(m))
    ws = torch.empty(max(nbytes // 2, 4), dtype=torch.uint8, device=DEV)
    rc = L.lib().krs_colsum(L.ptr(g), C.c_int64(n), C.c_int64(m), C.c_int64(n), C.c_int(L.fdtype(g)), L.ptr(out), L.ptr(ws),
                            C.c_size_t(nbytes // 2), L.stream_ptr())
    assert rc != 0 and b"workspace too small" in L.lib().krs_last_error()


@pytest.mark.parametrize("acc,fold,act", [(False, False, "linear"), (True, False, "relu"), (True, True, "linear")])
def test_fused_data_gradient_and_cross_backward_equals_the_two_calls(acc, fold, act):
    """krs_gemm_cross_bwd (round 4): G = A Bt^T + R, then from G as stored dz = G x0 act'(u), dx0 = [dx0 +] G u [+ G],
    dbias = colsum(dz) -- against krs_gemm(residual) followed by krs_cross_epilogue_bwd: G, dz and dx0 BIT-identical
    (bf16 shape the fused ring kernel takes: 192 tiles of 256 x 256), dbias to fp32 summation order; both against the
    oracle composition on a slice; and a small shape, which takes the two-call form inside the entry."""
    from keras_rs_amd import _lib as L
    from keras_rs_amd import dense_ops as D
    from oracle import krs_oracle as ko
    from tests.helpers import to_f32, to_np

    dev = "cuda:0"
    gen = torch.Generator(device=dev).manual_seed(11)
    a_id = {"linear": L.ACT_NONE, "relu": L.ACT_RELU}[act]
    for (m, n, k) in ((16384, 768, 256), (320, 264, 64)):
        rnd = lambda *sh: ((torch.rand(*sh, device=dev, generator=gen) - 0.5)).to(torch.bfloat16)  # noqa: E731
        A, Bt, R, x0, u, told = rnd(m, k), rnd(n, k) * 0.2, rnd(m, n), rnd(m, n), rnd(m, n), rnd(m, n)
        buf = told.clone() if acc else None
        G, dz, dx0, db = D.gemm_cross_bwd(A, Bt, R, x0, u, act=a_id, dx0_into=buf, fold_direct=fold)
        G2, _ = D.gemm(A, Bt, b_is_nk=True, r=R, beta=1.0)
        buf2 = told.clone() if acc else None
        dz2, dx02, _, db2 = D.cross_epilogue_bwd(G2, u, x0, x0, 0.0, act=a_id, dx0_into=buf2, want_dxd=False,
                                                  fold_direct=fold)
        assert torch.equal(G, G2) and torch.equal(dz, dz2) and torch.equal(dx0, dx02)
        torch.testing.assert_close(db, db2, rtol=1e-5, atol=1e-4)
        # oracle composition on the first 64 rows
        s = 64
        Gn, _ = ko.gemm(to_np(A[:s]), to_np(Bt), s, n, k, b_is_nk=True, r=to_np(R[:s]), beta=1.0)
        np.testing.assert_allclose(to_f32(to_np(G[:s])), to_f32(Gn), rtol=2.0 ** -7, atol=1e-6)
        dzn, dx0n, _, _ = ko.cross_epilogue_bwd(to_np(G[:s]), to_np(u[:s]), to_np(x0[:s]), to_np(x0[:s]), 0.0,
                                                dx0_init=to_np(told[:s]) if acc else None, act=act if act != "linear" else None,
                                                fold_direct=fold)
        np.testing.assert_allclose(to_f32(to_np(dz[:s])), to_f32(dzn), rtol=2.0 ** -7, atol=1e-6)
        np.testing.assert_allclose(to_f32(to_np(dx0[:s])), to_f32(dx0n), rtol=2.0 ** -7, atol=1e-6)
        if not acc:
            # u_upper: the term of the layer above, R * u_upper, starts dL/dx0 inside the same epilogue (R is its dL/dy):
            # G and dz as before, dL/dx0 = R u_upper + G u [+ G] rounded once (the two-pass composition rounds twice)
            u_up = rnd(m, n)
            G3, dz3, dx03, db3 = D.gemm_cross_bwd(A, Bt, R, x0, u, act=a_id, fold_direct=fold, u_upper=u_up)
            assert torch.equal(G3, G) and torch.equal(dz3, dz)
            torch.testing.assert_close(db3, db, rtol=1e-6, atol=1e-6)
            ref = R.float() * u_up.float() + G.float() * u.float() + (G.float() if fold else 0.0)
            # (one bf16 ulp of the LARGER of the two terms: they may cancel; the small shape runs the two-pass form
            #  inside the entry, whose intermediate rounding doubles that)
            scale = float((R.float() * u_up.float()).abs().max() + (G.float() * u.float()).abs().max())
            torch.testing.assert_close(dx03.float(), ref, rtol=2.0 ** -7, atol=2.0 ** -7 * scale)
            if not fold:
                # the form without a residual (a Dense layer above the stack) and without dL/dx0 (left to the next launch's
                # u_upper): G and dz are those of the same call with dx0
                G4, dz4, dx04, db4 = D.gemm_cross_bwd(A, Bt, None, x0, u, act=a_id)
                G5, dz5, none5, db5 = D.gemm_cross_bwd(A, Bt, None, x0, u, act=a_id, want_dx0=False)
                assert none5 is None and torch.equal(G5, G4) and torch.equal(dz5, dz4) and torch.equal(db5, db4)
                with pytest.raises(L.KrsError):
                    D.gemm_cross_bwd(A, Bt, R, x0, u, act=a_id, want_dx0=False)


# (k = 256, 320: whole 64-k blocks -> gemm_pp64_kernel, the 64-k ring of round 6; k = 288: not -> gemm_pp256_kernel)
@pytest.mark.parametrize("m,n,k", [(16384 + 72, 768 + 40, 256), (24576, 512, 320), (16384 + 72, 768 + 40, 288)])
def test_fused_cross_backward_every_epilogue_form_against_the_two_call_form_at_ragged_shapes(m, n, k):
    """krs_gemm_cross_bwd's fused ring kernel (EPI 3 .. 8) against the documented two-call form, which
    krs_gemm_set_option(KRS_GEMM_OPT_PIPELINE, 0) forces inside the same entry: m and n that are NOT multiples of the
    256-wide tile (partial tiles on both edges), with / without the residual, accumulating dx0, fold_direct, u_upper
    (EPI 7) and dx0 = NULL (EPI 8), with and without an activation.  G, dz and dx0 bit for bit (EPI 7's dx0: the fused form
    rounds R*u_upper + G*u once, the two calls twice -- one ulp of the larger term); dbias to fp32 summation order."""
    import ctypes as C

    from keras_rs_amd import _lib as L
    from keras_rs_amd import dense_ops as D

    dev = "cuda:0"
    gen = torch.Generator(device=dev).manual_seed(23)
    rnd = lambda *sh: ((torch.rand(*sh, device=dev, generator=gen) - 0.5)).to(torch.bfloat16)  # noqa: E731
    A, Bt, R, x0, u, told, u_up = rnd(m, k), rnd(n, k) * 0.2, rnd(m, n), rnd(m, n), rnd(m, n), rnd(m, n), rnd(m, n)
    forms = [dict(r=R), dict(r=R, acc=True), dict(r=R, fold=True), dict(r=R, acc=True, fold=True), dict(r=None),
             dict(r=None, acc=True), dict(r=R, u_upper=u_up), dict(r=R, u_upper=u_up, fold=True), dict(r=None, want_dx0=False)]

    def run(form, act):
        buf = told.clone() if form.get("acc") else None
        return D.gemm_cross_bwd(A, Bt, form["r"], x0, u, act=act, dx0_into=buf, fold_direct=form.get("fold", False),
                                u_upper=form.get("u_upper"), want_dx0=form.get("want_dx0", True))

    def set_pipe(v):
        L.check(L.lib().krs_gemm_set_option(C.c_int(0), C.c_int(v)), "krs_gemm_set_option")

    try:
        for act in (L.ACT_NONE, L.ACT_RELU):
            for form in forms:
                set_pipe(4)
                G, dz, dx0, db = run(form, act)
                set_pipe(0)
                G2, dz2, dx02, db2 = run(form, act)
                tag = (act, {k_: (v is not None if k_ in ("r", "u_upper") else v) for k_, v in form.items()})
                assert torch.equal(G, G2), tag
                assert torch.equal(dz, dz2), tag
                if dx0 is None:
                    assert dx02 is None
                elif form.get("u_upper") is not None:
                    scale = float((R.float() * u_up.float()).abs().max() + (G.float() * u.float()).abs().max()) + \
                        (float(G.float().abs().max()) if form.get("fold") else 0.0)
                    torch.testing.assert_close(dx0.float(), dx02.float(), rtol=2.0 ** -7, atol=2.0 ** -7 * scale)
                else:
                    assert torch.equal(dx0, dx02), tag
                torch.testing.assert_close(db, db2, rtol=1e-5, atol=2e-4)
    finally:
        set_pipe(4)


def test_ring_gemm_equals_the_two_stage_kernels_bit_for_bit():
    """krs_gemm's ring kernel (256 x 256 tiles, pipeline 4) against the 128 x 128 two-stage kernels (pipeline 0) on the
    three operand layouts of a cross layer at shapes with partial tiles: same fragment layout, same k order per
    accumulator -> identical bits (plain, cross-epilogue and residual forms; the weight gradient through split-K)."""
    import ctypes as C

    from keras_rs_amd import _lib as L
    from keras_rs_amd import dense_ops as D

    dev = "cuda:0"
    gen = torch.Generator(device=dev).manual_seed(29)
    rnd = lambda *sh: ((torch.rand(*sh, device=dev, generator=gen) - 0.5)).to(torch.bfloat16)  # noqa: E731
    m, d, p = 24576 + 64, 1024 + 64, 512      # (>= 192 tiles of 256 x 256 for every product; partial tiles on both edges)
    x, x0, g = rnd(m, d), rnd(m, d), rnd(m, d)
    Ut, Vt = rnd(p, d) * 0.1, rnd(d, p) * 0.1      # K-contiguous weights

    def products():
        h, _ = D.gemm(x, Ut, b_is_nk=True)                                   # [m, p], K = d
        y, uo = D.gemm(h, Vt, b_is_nk=True, x0=x0, x=x, want_u=True)         # cross epilogue, K = p
        dx, _ = D.gemm(h, Ut.t().contiguous(), b_is_nk=True, r=g, beta=1.0)  # residual form
        dk, _ = D.gemm(h, g, a_is_km=True, out_dtype=torch.float32)          # weight gradient [p, d], K = m
        return h, y, uo, dx, dk

    try:
        L.check(L.lib().krs_gemm_set_option(C.c_int(0), C.c_int(4)), "krs_gemm_set_option")
        ring = products()         # default: the 64-k ring (gemm_pp64_kernel) for the K-contiguous products, K = 1088 and 512
        L.check(L.lib().krs_gemm_set_option(C.c_int(0), C.c_int(5)), "krs_gemm_set_option")
        ring32 = products()       # the 32-k ring (gemm_pp256_kernel) for all of them
        L.check(L.lib().krs_gemm_set_option(C.c_int(0), C.c_int(0)), "krs_gemm_set_option")
        ref = products()
    finally:
        L.lib().krs_gemm_set_option(C.c_int(0), C.c_int(4))
    for got in (ring, ring32):
        for name, a, b in zip(("h", "y", "u", "dx", "dK"), got, ref):
            if name == "dK":     # split-K slabs: the split count is a function of the shape alone, but the tile kernels differ
                torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-4)
            else:
                assert torch.equal(a, b), name
    assert torch.equal(ring[4], ring32[4])     # (the weight gradient takes the same kernel under both)


@pytest.mark.parametrize("m,n,k,ep", [(8192, 512, 3456, None), (8192, 1024, 3456, "bias_relu"), (4096 + 64, 512 + 40, 2048 + 96, None)])
def test_split_k_on_the_ring_kernel_for_outputs_too_small_to_fill_the_chip(m, n, k, ep):
    """Round 5: K-contiguous bf16 products whose 256 x 256 tiles would not fill the chip (the per-rank `h = x U` / `dh = dz K^T`
    of a strongly-scaled job: M = 8192 against N = 512 is 64 tiles) run the ring kernel with the K range dealt to several
    workgroups per tile (fp32 slabs, fixed-order reduce with the vector epilogue) instead of one round of 128 x 128 tiles.
    Against float64: the bf16 output is the fp32 sum rounded once (<= 1 ulp of the exact value); partial tiles, a K whose last
    split is shorter, and the bias + ReLU epilogue of a Dense layer at that batch.  Run-to-run identical (no atomics)."""
    import ctypes as C

    from keras_rs_amd import _lib as L
    from keras_rs_amd import dense_ops as D

    dev = "cuda:0"
    gen = torch.Generator(device=dev).manual_seed(31)
    a = ((torch.rand(m, k, device=dev, generator=gen) - 0.5)).to(torch.bfloat16)
    bt = ((torch.rand(n, k, device=dev, generator=gen) - 0.5) * 0.2).to(torch.bfloat16)
    wsb = int(L.lib().krs_gemm_workspace_bytes(C.c_int64(m), C.c_int64(n), C.c_int64(k), C.c_int(0)))
    assert wsb >= 2 * m * n * 4, "the shape is expected to take split-K"
    bias = (torch.rand(n, device=dev, generator=gen) - 0.5) if ep else None
    got, _ = D.gemm(a, bt, b_is_nk=True, bias=bias, act=L.ACT_RELU if ep else L.ACT_NONE)
    again, _ = D.gemm(a, bt, b_is_nk=True, bias=bias, act=L.ACT_RELU if ep else L.ACT_NONE)
    assert torch.equal(got, again)
    rows = torch.cat([torch.arange(0, 256, device=dev), torch.arange(m - 130, m, device=dev)])
    ref = a[rows].double() @ bt.double().t()
    if ep:
        ref = torch.relu(ref + bias.double())
    # one bf16 rounding of an fp32 sum of k products of magnitude <= 0.05: relative 2^-8, absolute a few fp32 ulps of the sum's scale
    torch.testing.assert_close(got[rows].double(), ref, rtol=2.0 ** -8 * 1.01, atol=1e-4)
    # the reference schedule (pipeline 0: no ring) on the same split: same slabs order, other tile kernels -> same value to fp32 noise
    try:
        L.check(L.lib().krs_gemm_set_option(C.c_int(0), C.c_int(0)), "krs_gemm_set_option")
        ref0, _ = D.gemm(a, bt, b_is_nk=True, bias=bias, act=L.ACT_RELU if ep else L.ACT_NONE)
    finally:
        L.lib().krs_gemm_set_option(C.c_int(0), C.c_int(4))
    assert (got == ref0).float().mean() > 0.99
    torch.testing.assert_close(got.float(), ref0.float(), rtol=2.0 ** -7, atol=1e-4)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
def test_cross_backward_with_a_long_k_on_a_small_batch_needs_no_split_k_workspace(dt):
    """krs_gemm_cross_bwd's two-call form (fewer than 192 tiles of 256 x 256: a per-rank batch) with
    K >= 2048 -- a Dense layer of >= 2048 units on a cross output, or projection_dim >= 2048 -- used to hand krs_gemm a NULL
    workspace for a shape pick_splits deals over several workgroups per tile (KRS_ERR_WORKSPACE).  The inner product now runs in
    one pass over K; G against float64, dz / dx0 from G as stored.  fp32 / row-major products of that shape no longer take the
    ring's split either: krs_gemm with a NULL workspace succeeds for them."""
    import ctypes as C

    from keras_rs_amd import _lib as L
    from keras_rs_amd import dense_ops as D

    dev = "cuda:0"
    m, n, k = 2048, 3456, 2048
    assert int(L.lib().krs_gemm_workspace_bytes(C.c_int64(m), C.c_int64(n), C.c_int64(k), C.c_int(0))) > 0   # (a split shape)
    gen = torch.Generator(device=dev).manual_seed(37)
    rnd = lambda *sh: ((torch.rand(*sh, device=dev, generator=gen) - 0.5)).to(dt)  # noqa: E731
    A, Bt, R, x0, u = rnd(m, k), rnd(n, k) * 0.1, rnd(m, n), rnd(m, n), rnd(m, n)
    G, dz, dx0, db = D.gemm_cross_bwd(A, Bt, R, x0, u, act=L.ACT_NONE)
    ref = A.double() @ Bt.double().t() + R.double()
    tol = 2.0 ** -8 * 1.01 if dt == torch.bfloat16 else 1e-5
    torch.testing.assert_close(G.double(), ref, rtol=tol, atol=1e-4)
    torch.testing.assert_close(dz.double(), (G.double() * x0.double()), rtol=tol, atol=1e-6)
    torch.testing.assert_close(dx0.double(), (G.double() * u.double()), rtol=tol, atol=1e-6)
    # (the bias gradient sums the fp32 products G x0 BEFORE dz is rounded to the storage dtype)
    torch.testing.assert_close(db.double(), (G.double() * x0.double()).sum(0), rtol=1e-4, atol=2e-3)
    if dt == torch.float32:
        # krs_gemm itself, fp32 operands, NULL workspace: one pass (the ring's split is for bf16 [N, K] operands only)
        out = torch.empty(m, n, device=dev, dtype=dt)
        rc = L.lib().krs_gemm(C.c_void_p(A.data_ptr()), C.c_int64(k), C.c_int(0), C.c_void_p(Bt.data_ptr()), C.c_int64(k), C.c_int(1),
                              C.c_void_p(out.data_ptr()), C.c_int64(n), C.c_int64(m), C.c_int64(n), C.c_int64(k), C.c_int(L.F32),
                              C.c_int(L.F32), None, None, C.c_size_t(0), C.c_void_p(torch.cuda.current_stream().cuda_stream))
        L.check(rc, "krs_gemm")
        torch.testing.assert_close(out.double(), A.double() @ Bt.double().t(), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("m,n,k", [(24576 + 13, 515, 192), (24576, 776, 448), (65536, 512, 128 + 64)])
def test_the_64k_ring_at_its_edges(m, n, k):
    """gemm_pp64_kernel (round 6) where its bookkeeping is thinnest: K = 192 = three 64-k blocks (prologue, one tail block with the
    last piece, one with nothing in flight: the steady loop does not run), K = 448 (odd block count through five slots), an N that
    is not a multiple of 8 (scalar epilogue stores, clamped DMA rows in the last N tile) and an M with a partial last tile --
    against float64 on sampled rows and bit for bit against the reference schedule (pipeline 0) and the 32-k ring (pipeline 5),
    plain and with the bias + ReLU epilogue; run-to-run identical."""
    import ctypes as C

    from keras_rs_amd import _lib as L
    from keras_rs_amd import dense_ops as D

    dev = "cuda:0"
    gen = torch.Generator(device=dev).manual_seed(m + n + k)
    a = (torch.rand(m, k, device=dev, generator=gen) - 0.5).to(torch.bfloat16)
    bt = ((torch.rand(n, k, device=dev, generator=gen) - 0.5) * 0.2).to(torch.bfloat16)
    bias = torch.rand(n, device=dev, generator=gen) - 0.5
    rows = torch.cat([torch.arange(0, 300, device=dev), torch.arange(m - 300, m, device=dev)])
    ref = a[rows].double() @ bt.double().t()

    def both():
        plain, _ = D.gemm(a, bt, b_is_nk=True)
        act, _ = D.gemm(a, bt, b_is_nk=True, bias=bias, act=L.ACT_RELU)
        return plain, act

    got = both()
    again = both()
    assert torch.equal(got[0], again[0]) and torch.equal(got[1], again[1])
    torch.testing.assert_close(got[0][rows].double(), ref, rtol=2.0 ** -8 * 1.01, atol=1e-4)
    torch.testing.assert_close(got[1][rows].double(), torch.relu(ref + bias.double()), rtol=2.0 ** -8 * 1.01, atol=1e-4)
    try:
        for pipe in (0, 5):
            L.check(L.lib().krs_gemm_set_option(C.c_int(0), C.c_int(pipe)), "krs_gemm_set_option")
            other = both()
            assert torch.equal(got[0], other[0]) and torch.equal(got[1], other[1]), pipe
    finally:
        L.lib().krs_gemm_set_option(C.c_int(0), C.c_int(4))


def test_the_64k_ring_is_race_free_over_many_launches():
    """A staged LDS buffer read before its DMA has landed, or restaged before its last read, shows up as RARE wrong tiles that come
    and go with memory load (cdna_hip_programming.md, 8-phase template notes) -- so the C3 long-K product (54 blocks through the
    five slots, 512 workgroups in two rounds) and the K = 512 cross form run 150 times each, while a second stream keeps HBM busy,
    and every output must equal the first launch's bits (which equal the reference schedule's)."""
    import ctypes as C

    from keras_rs_amd import _lib as L
    from keras_rs_amd import dense_ops as D

    dev = "cuda:0"
    gen = torch.Generator(device=dev).manual_seed(77)
    m, d, p = 65536, 3456, 512
    x = (torch.rand(m, d, device=dev, generator=gen) - 0.5).to(torch.bfloat16)
    x0 = (torch.rand(m, d, device=dev, generator=gen) - 0.5).to(torch.bfloat16)
    ut = ((torch.rand(p, d, device=dev, generator=gen) - 0.5) * 0.1).to(torch.bfloat16)
    vt = ((torch.rand(d, p, device=dev, generator=gen) - 0.5) * 0.1).to(torch.bfloat16)
    try:
        L.check(L.lib().krs_gemm_set_option(C.c_int(0), C.c_int(0)), "krs_gemm_set_option")
        h_ref, _ = D.gemm(x, ut, b_is_nk=True)
        y_ref, u_ref = D.gemm(h_ref, vt, b_is_nk=True, x0=x0, x=x, want_u=True)
    finally:
        L.lib().krs_gemm_set_option(C.c_int(0), C.c_int(4))
    noise_src = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    noise_dst = torch.empty_like(noise_src)
    side = torch.cuda.Stream()
    bad = torch.zeros((), dtype=torch.int64, device=dev)
    for it in range(150):
        if it % 3 == 0:
            with torch.cuda.stream(side):
                noise_dst.copy_(noise_src)          # HBM traffic beside the product
        h, _ = D.gemm(x, ut, b_is_nk=True)
        y, u = D.gemm(h, vt, b_is_nk=True, x0=x0, x=x, want_u=True)
        bad += (h != h_ref).sum() + (y != y_ref).sum() + (u != u_ref).sum()
    torch.cuda.synchronize()
    assert int(bad) == 0
