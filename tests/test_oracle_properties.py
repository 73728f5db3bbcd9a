"""The oracle (oracle/krs_oracle.c) against INDEPENDENT formulations on random inputs: the golden
vectors (tests/test_oracle_golden.py) pin it at the reference's own small cases; here every function is
checked against a float64 numpy / torch-autograd composition of the reference formula it restates, on
random shapes with the awkward cases mixed in (empty bags, zero weights, shared tables, every combiner,
every activation, self-interaction / skip-gather).  CPU only; nothing here touches the HIP library."""

import numpy as np
import pytest
import torch

from oracle import krs_oracle as ko

SEEDS = range(4)


def _ragged(rng, batch, vocab, max_len):
    lens = rng.integers(0, max_len + 1, batch)
    lens[rng.integers(0, batch)] = 0                      # at least one empty bag
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    ids = rng.integers(0, vocab, int(offsets[-1])).astype(np.int32)
    w = rng.uniform(0.0, 2.0, ids.size).astype(np.float32)
    w[rng.random(ids.size) < 0.1] = 0.0                   # zero weights: divide_no_nan territory
    return ids, offsets, w


def _bag_reference(table, ids, offsets, w, combiner):
    """embedding/test_utils.py:245-267 with the divide_no_nan of embed_reduce.py:262-274, in float64."""
    out = np.zeros((offsets.size - 1, table.shape[1]))
    for b in range(offsets.size - 1):
        s, e = offsets[b], offsets[b + 1]
        ww = np.ones(e - s) if w is None else w[s:e].astype(np.float64)
        acc = (table[ids[s:e]].astype(np.float64) * ww[:, None]).sum(0)
        den = {"sum": 1.0, "mean": ww.sum(), "sqrtn": np.sqrt((ww * ww).sum())}[combiner]
        out[b] = 0.0 if den == 0 else acc / den
    return out


@pytest.mark.parametrize("seed", SEEDS)
@pytest.mark.parametrize("combiner", ["sum", "mean", "sqrtn"])
@pytest.mark.parametrize("weighted", [False, True])
def test_ragged_bags_match_the_reference_formula(seed, combiner, weighted):
    rng = np.random.default_rng(seed)
    vocab, dim, batch = int(rng.integers(3, 60)), int(rng.integers(1, 20)), int(rng.integers(2, 40))
    table = rng.normal(size=(vocab, dim)).astype(np.float32)
    ids, offsets, w = _ragged(rng, batch, vocab, 9)
    got = ko.embed_reduce_csr(table, ids, offsets, w if weighted else None, combiner)
    ref = _bag_reference(table, ids, offsets, w if weighted else None, combiner)
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("seed", SEEDS)
def test_lookup_gradient_matches_autograd_with_shared_tables(seed):
    """dE of two tables looked up by three features (two share a table), mixed combiners, dense bags:
    the oracle's scatter-add (jax/test_utils.py:395-417, 450-468) against torch autograd in float64."""
    rng = np.random.default_rng(100 + seed)
    dim, batch = int(rng.integers(1, 12)), int(rng.integers(2, 24))
    vocabs = [int(rng.integers(2, 30)), int(rng.integers(2, 30))]
    table_of = [0, 1, 0]
    combs = ["mean", "sum", "sqrtn"]
    hots = [int(rng.integers(1, 6)) for _ in table_of]
    tabs = [rng.normal(size=(v, dim)).astype(np.float32) for v in vocabs]
    ids = [rng.integers(0, vocabs[t], (batch, h)).astype(np.int32) for t, h in zip(table_of, hots)]
    ws = [rng.uniform(0.1, 2.0, (batch, h)).astype(np.float32) for h in hots]
    grad = rng.normal(size=(batch, 3 * dim)).astype(np.float32)
    # oracle: forward (for the per-bag scale) then the dense scatter-add
    tables = ko.make_tables(tabs)
    feats = ko.make_features(table_of, combs, [f * dim for f in range(3)], hots=hots, batch=batch)
    flat_ids = np.concatenate([i.reshape(-1) for i in ids])
    flat_w = np.concatenate([w.reshape(-1) for w in ws])
    out = np.zeros((batch, 3 * dim), np.float32)
    scale = np.zeros(3 * batch, np.float32)
    assert ko.embed_bag_fwd_raw(tables, ko.F32, feats, flat_ids, None, flat_w, batch, dim, out, bag_scale=scale) == 0
    dtabs = [np.zeros_like(t) for t in tabs]
    ko.embed_bag_bwd_dense(ko.make_tables(dtabs), feats, flat_ids, None, flat_w, scale, grad, batch, dim)
    # torch, float64
    tt = [torch.tensor(t, dtype=torch.float64, requires_grad=True) for t in tabs]
    cols = []
    for f, (t, c) in enumerate(zip(table_of, combs)):
        w = torch.tensor(ws[f], dtype=torch.float64)
        s = (tt[t][torch.tensor(ids[f], dtype=torch.long)] * w[..., None]).sum(1)
        den = {"sum": torch.ones(batch, dtype=torch.float64), "mean": w.sum(1), "sqrtn": (w * w).sum(1).sqrt()}[c]
        cols.append(s / den[:, None])
    ref_out = torch.cat(cols, 1)
    np.testing.assert_allclose(out, ref_out.detach().numpy(), rtol=1e-5, atol=1e-5)
    ref_out.backward(torch.tensor(grad, dtype=torch.float64))
    for d, t in zip(dtabs, tt):
        np.testing.assert_allclose(d, t.grad.numpy(), rtol=1e-5, atol=1e-5)


_ACT = {None: lambda z: z, "relu": torch.relu, "sigmoid": torch.sigmoid, "tanh": torch.tanh}


@pytest.mark.parametrize("seed", SEEDS)
@pytest.mark.parametrize("act", [None, "relu", "sigmoid", "tanh"])
@pytest.mark.parametrize("low_rank", [False, True])
def test_feature_cross_forward_and_backward_pieces_match_autograd(seed, act, low_rank):
    """feature_cross.py:182-194 and its autodiff: y, dz, the explicit-x0 half of dL/dx0, the
    direct half of dL/dx, the bias gradient and the four GEMM-shaped gradients."""
    rng = np.random.default_rng(200 + seed)
    m, d = int(rng.integers(1, 20)), int(rng.integers(2, 24))
    p = int(rng.integers(1, d)) if low_rank else None
    diag = float(rng.uniform(0.0, 0.5))
    f = lambda *s: rng.normal(size=s).astype(np.float32)   # noqa: E731
    x0, x, g = f(m, d), f(m, d), f(m, d)
    down = f(d, p) * 0.3 if low_rank else None
    kern = f(p if low_rank else d, d) * 0.3
    bias = f(d)
    y = ko.feature_cross(x0, x, kern, bias, down, diag, act)
    T = lambda a: torch.tensor(a, dtype=torch.float64, requires_grad=True)   # noqa: E731
    t0, tx, tk, tb = T(x0), T(x), T(kern), T(bias)
    td = T(down) if low_rank else None
    h = tx @ td if low_rank else tx
    z = h @ tk + tb
    z.retain_grad()
    u = _ACT[act](z)
    ty = t0 * (u + diag * tx) + tx
    np.testing.assert_allclose(y, ty.detach().numpy(), rtol=2e-5, atol=2e-5)
    ty.backward(torch.tensor(g, dtype=torch.float64))
    # oracle backward, composed the way keras_rs_amd/autograd.py composes the kernels
    un = u.detach().numpy().astype(np.float32)
    dz, dx0, dxd, dbias = ko.cross_epilogue_bwd(g, un, x0, x, diag, act=act)
    np.testing.assert_allclose(dz, z.grad.numpy(), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(dbias, tb.grad.numpy(), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(dx0, t0.grad.numpy(), rtol=2e-5, atol=2e-5)
    hn = h.detach().numpy().astype(np.float32)
    kk = hn.shape[1]
    dk, _ = ko.gemm(hn, dz, kk, d, m, a_is_km=True)                                 # dK = h^T dz
    np.testing.assert_allclose(dk, tk.grad.numpy(), rtol=5e-5, atol=5e-5)
    dh, _ = ko.gemm(dz, kern, m, kk, d, b_is_nk=True)                               # dh = dz K^T
    if low_rank:
        dd, _ = ko.gemm(x, dh, d, p, m, a_is_km=True)                               # dU = x^T dh
        np.testing.assert_allclose(dd, td.grad.numpy(), rtol=5e-5, atol=5e-5)
        dx, _ = ko.gemm(dh, down, m, d, p, b_is_nk=True, r=dxd, beta=1.0)           # dx = dh U^T + direct
    else:
        dx = dh + dxd
    np.testing.assert_allclose(dx, tx.grad.numpy(), rtol=5e-5, atol=5e-5)


@pytest.mark.parametrize("seed", SEEDS)
@pytest.mark.parametrize("self_interaction", [False, True])
@pytest.mark.parametrize("skip_gather", [False, True])
def test_dot_interaction_matches_autograd(seed, self_interaction, skip_gather):
    rng = np.random.default_rng(300 + seed)
    b, n, d = int(rng.integers(1, 9)), int(rng.integers(2, 9)), int(rng.integers(1, 12))
    feats = [rng.normal(size=(b, d)).astype(np.float32) for _ in range(n)]
    out = ko.dot_interaction_fwd(feats, self_interaction, skip_gather)
    tf = [torch.tensor(f, dtype=torch.float64, requires_grad=True) for f in feats]
    X = torch.stack(tf, 1)                                             # dot_interaction.py:170-203
    P = X @ X.transpose(1, 2)
    if skip_gather:
        mask = torch.tril(torch.ones(n, n, dtype=torch.float64), 0 if self_interaction else -1)
        ref = (P * mask).reshape(b, n * n)
    else:
        idx = [i * n + j for i in range(n) for j in range(i + 1 if self_interaction else i)]
        ref = P.reshape(b, n * n)[:, idx]
    np.testing.assert_allclose(out, ref.detach().numpy(), rtol=1e-5, atol=1e-5)
    go = rng.normal(size=out.shape).astype(np.float32)
    ref.backward(torch.tensor(go, dtype=torch.float64))
    for got, t in zip(ko.dot_interaction_bwd(feats, go, self_interaction, skip_gather), tf):
        np.testing.assert_allclose(got, t.grad.numpy(), rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("seed", SEEDS)
@pytest.mark.parametrize("dtype", [np.int32, np.int64])
def test_mod_bucketize_is_the_stable_sort_by_owner(seed, dtype):
    rng = np.random.default_rng(400 + seed)
    n, nnz = int(rng.integers(1, 9)), int(rng.integers(0, 300))
    ids = rng.integers(0, 10_000, nnz).astype(dtype)
    local, perm, counts = ko.mod_bucketize(ids, n)
    order = np.argsort(ids % n, kind="stable")
    np.testing.assert_array_equal(perm, order.astype(np.int32))
    np.testing.assert_array_equal(local, (ids // n)[order])
    np.testing.assert_array_equal(counts, np.bincount(ids % n, minlength=n))


@pytest.mark.parametrize("seed", SEEDS)
def test_sgd_and_adagrad_touch_only_the_flagged_rows(seed):
    """jax/test_utils.py:474-497: t -= lr g; acc += g^2, t -= lr g / sqrt(acc) (no epsilon), rows that
    were not looked up keep table and accumulator bit for bit."""
    rng = np.random.default_rng(500 + seed)
    v, d = int(rng.integers(2, 40)), int(rng.integers(1, 10))
    table = rng.normal(size=(v, d)).astype(np.float32)
    grad = rng.normal(size=(v, d)).astype(np.float32)
    touched = (rng.random(v) < 0.5).astype(np.uint8)
    t1 = table.copy()
    ko.apply_optimizer(t1, None, grad, touched, 0.1, "sgd")
    ref = np.where(touched[:, None] != 0, table.astype(np.float64) - 0.1 * grad, table)
    np.testing.assert_allclose(t1, ref, rtol=1e-6, atol=1e-7)
    np.testing.assert_array_equal(t1[touched == 0], table[touched == 0])
    t2, acc = table.copy(), np.full((v, d), 0.1, np.float32)
    ko.apply_optimizer(t2, acc, grad, touched, 0.01, "adagrad")
    a_ref = 0.1 + grad.astype(np.float64) ** 2
    ref = np.where(touched[:, None] != 0, table - 0.01 * grad / np.sqrt(a_ref), table)
    np.testing.assert_allclose(t2, ref, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(acc[touched != 0], a_ref[touched != 0], rtol=1e-6)
    np.testing.assert_array_equal(acc[touched == 0], np.full((int((touched == 0).sum()), d), 0.1, np.float32))
    np.testing.assert_array_equal(t2[touched == 0], table[touched == 0])


def test_bce_oracle_matches_a_float64_composition_with_autograd():
    """oracle bce_fwd_bwd against an independent float64 torch composition of keras.losses.BinaryCrossentropy()
    (clip to [eps, 1 - eps], mean of -(y log p + (1 - y) log(1 - p))) and its autograd gradient."""
    import torch

    rng = np.random.default_rng(4)
    n = 4097
    pred = rng.uniform(0, 1, n).astype(np.float32)
    pred[::50] = 0.0
    pred[1::50] = 1.0
    y = (rng.uniform(0, 1, n) < 0.4).astype(np.float32)
    loss, dp = ko.bce_fwd_bwd(pred, y, grad_scale=2.0)
    p64 = torch.from_numpy(pred.astype(np.float64)).requires_grad_()
    y64 = torch.from_numpy(y.astype(np.float64))
    eps = float(np.float32(1e-7))
    pc = p64.clamp(eps, float(np.float32(1.0) - np.float32(1e-7)))
    ref = -(y64 * torch.log(pc) + (1 - y64) * torch.log(1 - pc)).mean()
    (2.0 * ref).backward()
    np.testing.assert_allclose(float(loss), float(ref), rtol=1e-6)
    np.testing.assert_allclose(dp, p64.grad.numpy(), rtol=2e-5, atol=1e-12)
