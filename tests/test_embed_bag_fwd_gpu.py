"""K1 parity: krs_embed_bag_fwd (HIP, through the C ABI) vs the CPU oracle."""

import numpy as np
import pytest
import torch

from tests.helpers import make_bags, oracle_embed_fwd, to_f32, to_np

pytestmark = pytest.mark.gpu

TORCH_DT = {"f32": torch.float32, "bf16": torch.bfloat16}
NP_DT = {"f32": np.float32, "bf16": np.uint16}


def _run_case(dim, tdt, odt, csr, use_w, combiners, n_tables=3, batch=37, max_hot=5, seed=0,
              id_dtype=np.int32, extra_cols=0, shared=False):
    from keras_rs_amd.embedding_ops import FusedBags

    rng = np.random.default_rng(seed)
    dev = torch.device("cuda:0")
    vocabs_t = [int(v) for v in rng.integers(5, 90, size=n_tables)]
    tables = [torch.from_numpy(rng.uniform(-1, 1, (v, dim)).astype(np.float32)).to(TORCH_DT[tdt]).to(dev)
              for v in vocabs_t]
    # features: one per table (+ one extra on table 0 when `shared`)
    tix = list(range(n_tables)) + ([0] if shared else [])
    n_feats = len(tix)
    specs = [(tix[f], combiners[f % len(combiners)], extra_cols + f * dim) for f in range(n_feats)]
    out_cols = extra_cols + n_feats * dim + extra_cols
    bags = make_bags(rng, n_feats, batch, [vocabs_t[t] for t in tix], max_hot, csr, id_dtype=id_dtype)

    fb = FusedBags(tables, specs)
    out = torch.full((batch, out_cols), 7.0, dtype=TORCH_DT[odt], device=dev)
    ids = torch.from_numpy(bags["ids"]).to(dev)
    offs = None if bags["offsets"] is None else torch.from_numpy(bags["offsets"]).to(dev)
    w = torch.from_numpy(bags["weights"]).to(dev) if use_w else None
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    _, scale = fb.forward(ids, batch, hots=bags["hots"], offsets=offs, weights=w, out=out,
                          want_scale=True, err_flag=flag)
    torch.cuda.synchronize()
    assert int(flag.item()) == 0

    exp, exp_scale, _ = oracle_embed_fwd([to_np(t) for t in tables], specs, bags, batch, dim, out_cols,
                                         NP_DT[odt], use_w)
    got = to_np(out)
    lo, hi = extra_cols, extra_cols + n_feats * dim
    if odt == "bf16":
        # fp32 accumulate in the same order, one rounding: allow 1 bf16 ulp for a/den vs FMA contraction
        np.testing.assert_allclose(to_f32(got[:, lo:hi]), to_f32(exp[:, lo:hi]), rtol=2 ** -7, atol=1e-6)
    else:
        np.testing.assert_allclose(got[:, lo:hi], exp[:, lo:hi], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(scale.cpu().numpy(), exp_scale, rtol=1e-6, atol=0)
    # columns outside the features' slots are untouched
    if extra_cols:
        assert np.all(to_f32(got[:, :lo]) == 7.0) and np.all(to_f32(got[:, hi:]) == 7.0)


@pytest.mark.parametrize("dim", [4, 6, 7, 20, 32, 64, 96, 128, 256, 320])
@pytest.mark.parametrize("tdt,odt", [("f32", "f32"), ("bf16", "bf16"), ("bf16", "f32"), ("f32", "bf16")])
def test_dims_dtypes_dense(dim, tdt, odt):
    _run_case(dim, tdt, odt, csr=False, use_w=True, combiners=["sum", "mean", "sqrtn"])


@pytest.mark.parametrize("dim", [6, 64, 128])
@pytest.mark.parametrize("use_w", [False, True])
@pytest.mark.parametrize("comb", ["sum", "mean", "sqrtn"])
def test_csr_with_empty_bags(dim, use_w, comb):
    _run_case(dim, "f32", "f32", csr=True, use_w=use_w, combiners=[comb], batch=53, max_hot=9)


def test_int64_ids_and_shared_table_and_padded_output():
    _run_case(128, "bf16", "bf16", csr=False, use_w=False, combiners=["sum"], id_dtype=np.int64,
              extra_cols=8, shared=True)
    _run_case(64, "f32", "f32", csr=True, use_w=True, combiners=["mean"], id_dtype=np.int64,
              extra_cols=4, shared=True)


def test_long_bags_and_batch_not_multiple_of_wave():
    _run_case(128, "bf16", "bf16", csr=False, use_w=True, combiners=["sum", "mean"], batch=131, max_hot=100)
    _run_case(128, "f32", "f32", csr=True, use_w=False, combiners=["sqrtn"], batch=3, max_hot=70)


def test_gather_is_bit_exact_for_hot1_sum():
    """hot=1 / sum / no weights is a pure row gather: output rows == table rows, bit for bit."""
    from keras_rs_amd.embedding_ops import FusedBags

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    for dt in (torch.float32, torch.bfloat16):
        tab = torch.randn(1000, 128, generator=g).to(dt).to(dev)
        ids = torch.randint(0, 1000, (4096,), generator=g, dtype=torch.int32).to(dev)
        out, _ = FusedBags([tab], [(0, "sum", 0)]).forward(ids, 4096, hots=[1])
        assert torch.equal(out, tab[ids.long()])


def test_out_of_range_ids_raise_flag_and_contribute_nothing():
    from keras_rs_amd import _lib as L
    from keras_rs_amd.embedding_ops import FusedBags

    dev = torch.device("cuda:0")
    tab = torch.ones(10, 64, device=dev)
    ids = torch.tensor([[1, 10], [-1, 2], [3, 4]], dtype=torch.int32, device=dev).reshape(-1)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    out, _ = FusedBags([tab], [(0, "sum", 0)]).forward(ids, 3, hots=[2], err_flag=flag)
    assert int(flag.item()) & L.FLAG_ID_OUT_OF_RANGE
    assert torch.equal(out[:, 0].cpu(), torch.tensor([1.0, 1.0, 2.0]))


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_one_lookup_per_bag_on_average_but_uneven_hots(dt):
    """nnz == bags selects the pure-gather kernel; features whose hot is not 1 must still be right."""
    from keras_rs_amd.embedding_ops import FusedBags

    rng = np.random.default_rng(9)
    dev = torch.device("cuda:0")
    batch, dim = 70, 64
    tables = [torch.from_numpy(rng.uniform(-1, 1, (40, dim)).astype(np.float32)).to(TORCH_DT[dt]).to(dev)
              for _ in range(3)]
    hots = [2, 0, 1]
    specs = [(0, "mean", 0), (1, "sum", dim), (2, "sqrtn", 2 * dim)]
    ids = np.concatenate([rng.integers(0, 40, batch * h) for h in hots]).astype(np.int32)
    out, scale = FusedBags(tables, specs).forward(torch.from_numpy(ids).to(dev), batch, hots=hots, want_scale=True)
    bags = dict(ids=ids, offsets=None, hots=hots, weights=None, nnz=len(ids))
    exp, exp_scale, _ = oracle_embed_fwd([to_np(t) for t in tables], specs, bags, batch, dim, 3 * dim, NP_DT[dt], False)
    np.testing.assert_allclose(to_f32(to_np(out)), to_f32(exp), rtol=2 ** -7 if dt == "bf16" else 1e-6, atol=1e-6)
    np.testing.assert_allclose(scale.cpu().numpy(), exp_scale, rtol=1e-6)


@pytest.mark.parametrize("dt,dim", [("bf16", 128), ("f32", 64), ("f32", 256)])
def test_bags_longer_than_one_staging_window(dt, dim):
    """A group's lookup stream is staged in LDS windows of 8*LPR ids: bags of 300 / 700 lookups span
    several windows (dense and CSR, with weights, mean combiner so the divisor crosses windows too)."""
    from keras_rs_amd.embedding_ops import FusedBags

    rng = np.random.default_rng(21)
    dev = torch.device("cuda:0")
    batch = 9
    tables = [torch.from_numpy(rng.uniform(-1, 1, (500, dim)).astype(np.float32)).to(TORCH_DT[dt]).to(dev)
              for _ in range(2)]
    specs = [(0, "mean", 0), (1, "sum", dim), (0, "sqrtn", 2 * dim)]
    tol = dict(rtol=2 ** -7, atol=1e-4) if dt == "bf16" else dict(rtol=2e-6, atol=2e-6)
    # dense: hots 1, 300, 700
    hots = [1, 300, 700]
    ids = np.concatenate([rng.integers(0, 500, batch * h) for h in hots]).astype(np.int32)
    w = rng.uniform(0, 1, ids.shape[0]).astype(np.float32)
    out, _ = FusedBags(tables, specs).forward(torch.from_numpy(ids).to(dev), batch, hots=hots,
                                              weights=torch.from_numpy(w).to(dev))
    bags = dict(ids=ids, offsets=None, hots=hots, weights=w, nnz=len(ids))
    exp, _, _ = oracle_embed_fwd([to_np(t) for t in tables], specs, bags, batch, dim, 3 * dim, NP_DT[dt], True)
    np.testing.assert_allclose(to_f32(to_np(out)), to_f32(exp), **tol)
    # CSR: ragged lengths up to 900, some empty
    lens = rng.integers(0, 900, size=(3, batch))
    lens[0, 3] = 0
    offsets = np.concatenate([[0], np.cumsum(lens.reshape(-1))]).astype(np.int32)
    ids = rng.integers(0, 500, int(lens.sum())).astype(np.int32)
    w = rng.uniform(0, 1, ids.shape[0]).astype(np.float32)
    out, _ = FusedBags(tables, specs).forward(torch.from_numpy(ids).to(dev), batch,
                                              offsets=torch.from_numpy(offsets).to(dev),
                                              weights=torch.from_numpy(w).to(dev))
    bags = dict(ids=ids, offsets=offsets, hots=None, weights=w, nnz=len(ids))
    exp, _, _ = oracle_embed_fwd([to_np(t) for t in tables], specs, bags, batch, dim, 3 * dim, NP_DT[dt], True)
    np.testing.assert_allclose(to_f32(to_np(out)), to_f32(exp), **tol)


@pytest.mark.parametrize("rows", [64, 128])
@pytest.mark.parametrize("hots", [[1, 3, 9], [2, 2, 2]])
def test_hot_rows_staged_in_lds_give_identical_outputs(rows, hots):
    """KRS_EMBED_OPT_HOTROWS: lookups of rows 0 .. n-1 are served from an LDS copy through flat loads -- the pooled
    outputs must be bit-identical to the plain kernel's (same fp32 accumulation order), with ids concentrated on the
    staged rows, a table smaller than the staged window, and workgroups that straddle two features (no staging there)."""
    import ctypes as C

    from keras_rs_amd import _lib as L
    from keras_rs_amd.embedding_ops import FusedBags

    rng = np.random.default_rng(rows)
    dev = torch.device("cuda:0")
    B, D = 1999, 128
    vocabs = [5000, 40, 700]
    tables = [torch.from_numpy(rng.uniform(-1, 1, (v, D)).astype(np.float32)).to(torch.bfloat16).to(dev) for v in vocabs]
    ids = torch.from_numpy(np.concatenate([
        np.minimum((rng.uniform(0, 1, B * h) ** 4 * vocabs[t]).astype(np.int64), vocabs[t] - 1) for t, h in enumerate(hots)
    ]).astype(np.int32)).to(dev)
    fb = FusedBags(tables, [(t, "sum", t * D) for t in range(3)])
    try:
        L.check(L.lib().krs_embed_set_option(C.c_int(3), C.c_int(0)), "set_option")
        ref, _ = fb.forward(ids, B, hots=hots)
        L.check(L.lib().krs_embed_set_option(C.c_int(3), C.c_int(rows)), "set_option")
        got, _ = fb.forward(ids, B, hots=hots)
        torch.cuda.synchronize()
    finally:
        L.lib().krs_embed_set_option(C.c_int(3), C.c_int(0))
    assert torch.equal(ref, got)
