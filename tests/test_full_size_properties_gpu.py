"""Parity at BASELINE.json's full sizes (C3: 26 tables x 1M rows x 128, batch 65,536) through
size-independent properties -- the oracle would need minutes there, these need seconds:
exact gathers, linearity in the weights, gradient checksums, optimizer round trips, permutation
checks of the MOD bucketise, algebraic identities of DotInteraction / FeatureCross."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
T, V, D, B = 26, 1_000_000, 128, 65536
HOTS = [3, 2, 1, 2, 6, 1, 1, 1, 1, 7, 3, 8, 1, 6, 9, 5, 1, 1, 1, 12, 100, 27, 10, 3, 1, 1]


@pytest.fixture(scope="module")
def c3():
    from keras_rs_amd.embedding_ops import FusedBags

    g = torch.Generator(device=DEV).manual_seed(1337)
    tables = [(torch.rand(V, D, device=DEV, generator=g) * 0.1 - 0.05).to(torch.bfloat16) for _ in range(T)]
    fb = FusedBags(tables, [(t, "sum", t * D) for t in range(T)])
    return tables, fb, g


def test_one_hot_lookup_is_an_exact_gather_at_full_size(c3):
    tables, fb, g = c3
    ids = torch.randint(0, V, (T, B), device=DEV, generator=g, dtype=torch.int32)
    out, _ = fb.forward(ids.reshape(-1), B, hots=[1] * T)
    for t in (0, 7, 25):
        assert torch.equal(out[:, t * D:(t + 1) * D], tables[t][ids[t].long()])
    # checksum of checksums over all tables: bit patterns of the gathered rows
    ref = torch.stack([tables[t][ids[t].long()].view(torch.int16).sum(dtype=torch.int64) for t in range(T)])
    got = torch.stack([out[:, t * D:(t + 1) * D].contiguous().view(torch.int16).sum(dtype=torch.int64) for t in range(T)])
    assert torch.equal(ref, got)


def test_multi_hot_pooling_is_linear_in_the_weights_and_matches_torch_on_a_slice(c3):
    tables, fb, g = c3
    ids = torch.cat([torch.randint(0, V, (B * h,), device=DEV, generator=g, dtype=torch.int32) for h in HOTS])
    nnz = ids.numel()
    w1 = torch.rand(nnz, device=DEV, generator=g)
    w2 = torch.rand(nnz, device=DEV, generator=g)
    o1, _ = fb.forward(ids, B, hots=HOTS, weights=w1, out_dtype=torch.float32)
    o2, _ = fb.forward(ids, B, hots=HOTS, weights=w2, out_dtype=torch.float32)
    o12, _ = fb.forward(ids, B, hots=HOTS, weights=w1 + w2, out_dtype=torch.float32)
    torch.testing.assert_close(o1 + o2, o12, rtol=1e-5, atol=1e-5)
    # the heaviest feature (hot = 100) against a plain torch composition on the first 512 samples
    f = 20
    base = B * sum(HOTS[:f])
    sl = ids[base: base + 512 * 100].reshape(512, 100).long()
    ws = w1[base: base + 512 * 100].reshape(512, 100)
    ref = (tables[f][sl].float() * ws[..., None]).sum(1)
    torch.testing.assert_close(o1[:512, f * D:(f + 1) * D], ref, rtol=1e-5, atol=1e-5)


def test_gradient_checksum_and_sgd_round_trip_at_full_size(c3):
    tables, fb, g = c3
    hots = [1] * T
    ids = torch.randint(0, V, (T * B,), device=DEV, generator=g, dtype=torch.int32)
    grad = (torch.rand(B, T * D, device=DEV, generator=g) - 0.5).to(torch.bfloat16)
    ws = fb.plan_backward(ids, B, hots=hots, global_order=True)   # the compact form is indexed by segment
    # sparse form: every lookup's gradient lands in exactly one unique row; column sums are preserved
    rows, vals = fb.backward_sparse(ws, grad, B, ids.numel(), hots=hots)
    assert torch.all(rows[1:] > rows[:-1])
    uniq = torch.unique(ids.reshape(T, B).long() + torch.arange(T, device=DEV)[:, None] * V)
    assert torch.equal(rows, uniq)
    per_table = vals.sum(0, dtype=torch.float64)  # [D]: sum over tables of the column sums
    ref = grad.float().reshape(B, T, D).sum((0, 1), dtype=torch.float64)
    torch.testing.assert_close(per_table, ref, rtol=1e-6, atol=1e-3)
    # fused SGD with lr then -lr restores the tables up to bf16 rounding of the intermediate
    fb.lrs = [0.5] * T
    before = [t.clone() for t in tables[:2]]
    fb.backward_fused("sgd", ws, grad, B, ids.numel(), hots=hots)
    assert not torch.equal(tables[0], before[0])
    fb.lrs = [-0.5] * T
    fb.backward_fused("sgd", ws, grad, B, ids.numel(), hots=hots)
    for t, b in zip(tables[:2], before):
        torch.testing.assert_close(t.float(), b.float(), rtol=0, atol=2 ** -8)
    fb.lrs = [0.0] * T


def test_mod_bucketize_is_a_stable_partition_at_full_size():
    from keras_rs_amd import dense_ops as Dn

    g = torch.Generator(device=DEV).manual_seed(3)
    ids = torch.randint(0, V, (B * sum(HOTS),), device=DEV, generator=g, dtype=torch.int32)
    for n in (2, 8):
        local, perm, counts = Dn.mod_bucketize(ids, n)
        assert int(counts.sum()) == ids.numel()
        src = ids[perm.long()]
        shard = torch.repeat_interleave(torch.arange(n, device=DEV), counts)
        assert torch.equal(src % n, shard.to(src.dtype))           # grouped by owner
        assert torch.equal(local * n + shard.to(local.dtype), src)  # local row + shard reconstruct the id
        assert torch.equal(torch.sort(perm.long()).values, torch.arange(ids.numel(), device=DEV))
        for s in range(n):  # stable: source positions ascend inside every bucket
            lo = int(counts[:s].sum())
            seg = perm[lo: lo + int(counts[s])]
            assert torch.all(seg[1:] > seg[:-1])


def test_dot_interaction_and_feature_cross_identities_at_full_size():
    import keras_rs_amd.layers as kl

    g = torch.Generator(device=DEV).manual_seed(5)
    F = 27
    x0 = (torch.rand(B, F * D, device=DEV, generator=g) - 0.5).to(torch.bfloat16)
    feats = [x0[:, f * D:(f + 1) * D] for f in range(F)]
    full = kl.DotInteraction(self_interaction=True, skip_gather=True, dtype="bfloat16")(feats).reshape(B, F, F)
    packed = kl.DotInteraction(dtype="bfloat16")(feats)
    idx = [(i, j) for i in range(F) for j in range(i)]
    ii, jj = torch.tensor([i for i, _ in idx], device=DEV), torch.tensor([j for _, j in idx], device=DEV)
    assert torch.equal(packed, full[:, ii, jj])                       # same dots, two packings
    assert torch.all(full[:, jj, ii] == 0)                            # strictly-upper part is masked
    diag = torch.stack([(f.float() ** 2).sum(1) for f in feats], 1)   # P[i][i] = |x_i|^2
    torch.testing.assert_close(full[:, range(F), range(F)].float(), diag, rtol=2 ** -7, atol=1e-2)
    # FeatureCross with a zero kernel: y = x0 * bias + x exactly (one fp32 fma, one rounding)
    layer = kl.FeatureCross(projection_dim=512, kernel_initializer="zeros", bias_initializer="ones",
                            dtype="mixed_bfloat16")
    x = (torch.rand(B, F * D, device=DEV, generator=g) - 0.5).to(torch.bfloat16)
    y = layer(x0, x)
    assert torch.equal(y, (x0.float() + x.float()).to(torch.bfloat16))


def test_backward_identities_at_full_size():
    """C3-size backward through the 256x256-tile GEMMs (LDS-DMA, transposing-read weight gradient) and
    the DotInteraction backward, against closed forms that need no oracle run:
    FeatureCross with kernel V = 0 and bias 1: u = 1, so dL/dx = g and dL/dx0 = g exactly, dU = 0 exactly,
    dV = (x U)^T (g * x0) and dbias = colsum(g * x0) (checked against fp32 torch on a row/column slice);
    DotInteraction: dX = (G + G^T) X on a slice of the batch."""
    import keras_rs_amd.layers as kl
    from keras_rs_amd.layers import base

    gen = torch.Generator(device=DEV).manual_seed(11)
    F, d, p = 27, 27 * D, 512
    x0 = ((torch.rand(B, d, device=DEV, generator=gen) - 0.5)).to(torch.bfloat16).requires_grad_(True)
    x = ((torch.rand(B, d, device=DEV, generator=gen) - 0.5)).to(torch.bfloat16).requires_grad_(True)
    layer = kl.FeatureCross(projection_dim=p, kernel_initializer=base.GlorotUniform(seed=2), bias_initializer="ones",
                            dtype="mixed_bfloat16")
    layer.build(x0.shape)
    with torch.no_grad():
        layer.kernel.zero_()                                           # V = 0, U stays random
    g = ((torch.rand(B, d, device=DEV, generator=gen) - 0.5)).to(torch.bfloat16)
    y = layer(x0, x)
    assert torch.equal(y, (x0.float() + x.float()).to(torch.bfloat16))
    y.backward(g)
    assert torch.equal(x.grad, g) and torch.equal(x0.grad, g)          # u = 1: both are g, bit for bit
    assert torch.count_nonzero(layer.down_kernel.grad) == 0            # dh = dz V^T = 0
    dz = (g.float() * x0.detach().float()).to(torch.bfloat16)          # what the elementwise pass hands the GEMMs
    h = (x.detach().float() @ layer.down_kernel.detach().to(torch.bfloat16).float()).to(torch.bfloat16)
    cols = slice(1000, 1256)
    exp = (h.double().t() @ dz[:, cols].double()).float()             # the same bf16 products, summed in fp64
    # fp32 accumulation of 65,536 exact products (|sum| ~ 5): split-K partial sums + fixed-order reduce.  `h` above is
    # torch's rounding of x U; the layer's own h differs from it by one bf16 ulp in a few elements (different
    # summation order), which moves a sum by up to ~ 5e-3 -- test_c3_gemm_slices_against_fp64 below checks the same
    # product on the kernel's own operands to 2e-3
    torch.testing.assert_close(layer.kernel.grad[:, cols], exp, rtol=2e-4, atol=8e-3)
    # the bias gradient sums the unrounded fp32 products (the bf16 dz only feeds the GEMMs)
    torch.testing.assert_close(layer.bias.grad, (g.double() * x0.detach().double()).sum(0).float(), rtol=1e-3, atol=5e-3)
    # DotInteraction backward on the full batch, checked on 512 samples
    feats = [x0.detach()[:, f * D:(f + 1) * D].clone().requires_grad_(True) for f in range(F)]
    out = kl.DotInteraction(dtype="bfloat16")(feats)
    go = ((torch.rand(out.shape, device=DEV, generator=gen) - 0.5)).to(torch.bfloat16)
    out.backward(go)
    sl = slice(4096, 4608)
    X = torch.stack([f.detach()[sl].float() for f in feats], 1)        # [512, F, D]
    G = torch.zeros(512, F, F, device=DEV)
    ii, jj = torch.tril_indices(F, F, -1, device=DEV)
    G[:, ii, jj] = go[sl].float()
    dX = torch.matmul(G + G.transpose(1, 2), X)
    for f in (0, 13, 26):
        torch.testing.assert_close(feats[f].grad[sl].float(), dX[:, f], rtol=2 ** -6, atol=2e-2)


def test_c3_gemm_slices_against_fp64():
    """The six products of one C3 cross layer (65,536 x 3456, projection 512, bf16 in / fp32 accumulate) on the
    256x256-tile kernels, checked on row / column slices against float64 sums of the same bf16 operands:
    bf16 outputs to one rounding of the fp64 value (2^-8 relative) plus the fp32 accumulation error, fp32
    weight gradients to 2e-4 relative."""
    from keras_rs_amd import dense_ops as Dn

    gen = torch.Generator(device=DEV).manual_seed(21)
    d, p = 27 * D, 512
    rn = lambda *s: (torch.rand(*s, device=DEV, generator=gen) - 0.5).to(torch.bfloat16)  # noqa: E731
    x0, x, g = rn(B, d), rn(B, d), rn(B, d)
    U, V = (rn(d, p).float() * 0.1).to(torch.bfloat16), (rn(p, d).float() * 0.1).to(torch.bfloat16)
    bias = torch.rand(d, device=DEV, generator=gen) - 0.5
    rows = slice(40_000, 40_000 + 384)          # straddles a tile boundary (40,000 = 156.25 tiles)
    bf = lambda t: t.to(torch.bfloat16)         # noqa: E731
    tol16 = dict(rtol=2 ** -8 + 1e-4, atol=1e-3)

    h, _ = Dn.gemm(x, U.t().contiguous(), b_is_nk=True)                                   # h = x U
    h64 = x[rows].double() @ U.double()
    torch.testing.assert_close(h[rows].double(), h64, **tol16)
    y, u = Dn.gemm(h, V.t().contiguous(), b_is_nk=True, bias=bias, x0=x0, x=x, want_u=True)  # cross epilogue
    u64 = h[rows].double() @ V.double() + bias.double()
    torch.testing.assert_close(u[rows].double(), u64, **tol16)
    y64 = x0[rows].double() * u64 + x[rows].double()
    torch.testing.assert_close(y[rows].double(), y64, **tol16)
    dz = bf(g.float() * x0.float())
    dh, _ = Dn.gemm(dz, V, b_is_nk=True)                                                  # dh = dz V^T
    torch.testing.assert_close(dh[rows].double(), dz[rows].double() @ V.double().t(), **tol16)
    dx, _ = Dn.gemm(dh, U, b_is_nk=True, r=g, beta=1.0)                                   # dx = dh U^T + g
    torch.testing.assert_close(dx[rows].double(), dh[rows].double() @ U.double().t() + g[rows].double(), **tol16)
    cols = slice(2900, 3200)
    dk, _ = Dn.gemm(h, dz, a_is_km=True, out_dtype=torch.float32)                         # dV = h^T dz
    torch.testing.assert_close(dk[:, cols].double(), h.double().t() @ dz[:, cols].double(), rtol=2e-4, atol=2e-3)
    du, _ = Dn.gemm(x, dh, a_is_km=True, out_dtype=torch.float32)                         # dU = x^T dh
    torch.testing.assert_close(du[cols].double(), x[:, cols].double().t() @ dh.double(), rtol=2e-4, atol=2e-3)


def _c3_slice_against_the_oracle(tables, g, S, RS, bit_equal):
    """Body of the two full-size oracle tests below: forward on the first S samples, fused Adagrad of the FULL batch on the rows
    the first RS samples touch; fills `bit_equal` with the fractions of bit-identical elements."""
    from keras_rs_amd.embedding_ops import FusedBags
    from oracle import krs_oracle as ko
    from tests.helpers import to_f32, to_np

    LR, ACC0 = 0.01, 0.1
    f32 = tables[0].dtype == torch.float32
    odt, npdt = (ko.F32, np.float32) if f32 else (ko.BF16, np.uint16)
    as_f32 = (lambda x: x) if f32 else to_f32
    rtol_t = 1e-6 if f32 else 2.0 ** -7
    combs = [("sum", "mean", "sqrtn")[t % 3] for t in range(T)]
    ids = [torch.randint(0, V, (B, h), device=DEV, generator=g, dtype=torch.int32) for h in HOTS]
    flat = torch.cat([x.reshape(-1) for x in ids])
    nnz = flat.numel()
    tabs2 = [t.clone() for t in tables]
    slots = [torch.full((V, D), ACC0, device=DEV) for _ in range(T)]
    fb = FusedBags(tabs2, [(t, combs[t], t * D) for t in range(T)], slots=slots, lrs=[LR] * T)
    out, scale = fb.forward(flat, B, hots=HOTS, want_scale=True)
    # ---- forward slice through the oracle
    comp, cids = [], []
    for t in range(T):
        uniq, inv = torch.unique(ids[t][:S].long(), return_inverse=True)
        comp.append(np.ascontiguousarray(to_np(tables[t][uniq])))
        cids.append(inv.reshape(-1).to(torch.int32).cpu().numpy())
    feats = ko.make_features(list(range(T)), combs, [t * D for t in range(T)], hots=HOTS, batch=S)
    exp = np.zeros((S, T * D), npdt)
    exp_scale = np.zeros(T * S, np.float32)
    flags = ko.embed_bag_fwd_raw(ko.make_tables(comp), odt, feats, np.concatenate(cids), None, None, S, D, exp, exp_scale)
    assert flags == 0
    got = to_np(out[:S])
    fwd_equal = float((got == exp).mean())
    assert fwd_equal > 0.9999, fwd_equal
    np.testing.assert_allclose(as_f32(got), as_f32(exp), rtol=rtol_t, atol=1e-7)
    sc = scale.reshape(T, B)[:, :S].cpu().numpy()
    np.testing.assert_allclose(sc, exp_scale.reshape(T, S), rtol=1e-6, atol=0)
    # ---- fused Adagrad at full size, checked by the oracle on a subset of rows
    grad = (torch.rand(B, T * D, device=DEV, generator=g) - 0.5).to(tables[0].dtype)
    ws = fb.plan_backward(flat, B, hots=HOTS, global_order=False)
    fb.backward_fused("adagrad", ws, grad, B, nnz, hots=HOTS, bag_scale=scale)
    torch.cuda.synchronize()
    rows_sel, ids_c, lens = [], [], []
    for t in range(T):
        R = torch.unique(ids[t][:RS].long())
        mask = torch.isin(ids[t].long(), R)
        sel = ids[t].long()[mask]                           # sample-major, position-minor: the feature's lookup order
        ids_c.append(torch.searchsorted(R, sel).to(torch.int32).cpu().numpy())
        lens.append(mask.sum(1).cpu().numpy().astype(np.int64))
        rows_sel.append(R)
    offsets = np.concatenate([[0], np.cumsum(np.concatenate(lens))]).astype(np.int64)
    n_sel = int(offsets[-1])
    assert 100_000 < n_sel < 2_000_000, n_sel
    feats_csr = ko.make_features(list(range(T)), combs, [t * D for t in range(T)])
    dense = [np.zeros((len(r), D), np.float32) for r in rows_sel]
    gnp = np.ascontiguousarray(to_np(grad))
    ko.embed_bag_bwd_dense(ko.make_tables(dense), feats_csr, np.concatenate(ids_c), offsets, None,
                           np.ascontiguousarray(scale.cpu().numpy()), gnp, B, D)
    worst = 1.0
    for t in range(T):
        R = rows_sel[t]
        tab = np.ascontiguousarray(to_np(tables[t][R]))             # the rows BEFORE the update
        acc = np.full((len(R), D), ACC0, np.float32)
        ko.apply_optimizer(tab, acc, dense[t], None, LR, "adagrad")
        got_t, got_a = to_np(tabs2[t][R]), slots[t][R].cpu().numpy()
        np.testing.assert_allclose(got_a, acc, rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(as_f32(got_t), as_f32(tab), rtol=rtol_t, atol=1e-7)
        worst = min(worst, float((got_t == tab).mean()))
        assert not np.array_equal(got_t, to_np(tables[t][R]))      # the rows did move
    assert worst > 0.999, worst
    bit_equal.update(forward=fwd_equal, update=worst)
    # rows outside every lookup of the batch keep value and accumulator (checked on the heaviest table)
    f = 20
    untouched = torch.ones(V, dtype=torch.bool, device=DEV)
    untouched[ids[f].reshape(-1).long()] = False
    assert torch.equal(tabs2[f][untouched], tables[f][untouched]) and bool((slots[f][untouched] == ACC0).all())


def test_c3_full_size_meets_the_oracle_on_a_slice(c3):
    """Round-3 review: at C3 full size the pooled output and the fused Adagrad update met the oracle only through
    properties and a torch slice.  Here the ORACLE itself (oracle/krs_oracle.c) is the checker, at the real sizes --
    26 x 1 M x 128 bf16 tables, batch 65,536, the ml_perf bag lengths, sum / mean / sqrtn combiners:
      * forward: krs_oracle_embed_bag_fwd on the first 4096 samples of the batch (0.9 M lookups; the tables it needs are
        the rows those samples touch, handed to it as compact tables with remapped ids -- same rows, same order);
      * backward: the fused Adagrad update of the FULL batch on a subset of rows (the rows the first 512 samples touch,
        ~0.4 M of the 14 M lookups land on them, from anywhere in the batch): krs_oracle_embed_bag_bwd_dense over exactly
        those lookups in their original order + krs_oracle_apply_optimizer(adagrad), against the rows and accumulators
        K2 left in the real tables."""
    tables, _, g = c3
    _c3_slice_against_the_oracle(tables, g, 4096, 512, {})


def test_c3_size_fp32_tables_meet_the_oracle_bit_for_bit():
    """Round-4 review (next #5b): the same check on the fp32 path at C3 size -- 26 x 1 M x 128 FP32 tables (13.3 GB),
    fp32 activations and gradients, batch 65,536, the ml_perf bag lengths: the pooled output of the first 2048 samples and
    the fused Adagrad update of the full batch on the rows of 256 samples against oracle/krs_oracle.c.  fp32 has no
    rounding step to hide behind: kernel and oracle accumulate with fmaf in ascending position, so the fraction of
    bit-identical elements is asserted (> 99.99 % forward, > 99.9 % of the updated rows; both are 100 % when this was
    written) next to the 1e-6 tolerance."""
    g = torch.Generator(device=DEV).manual_seed(4242)
    tables = [torch.rand(V, D, device=DEV, generator=g) * 0.1 - 0.05 for _ in range(T)]
    eq = {}
    _c3_slice_against_the_oracle(tables, g, 2048, 256, eq)
    assert eq["forward"] > 0.9999 and eq["update"] > 0.999, eq
    del tables
    torch.cuda.empty_cache()
