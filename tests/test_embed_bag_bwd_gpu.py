"""K2 parity: sort-based index-scatter gradient (dense / sparse / fused SGD+Adagrad) vs the oracle."""

import numpy as np
import pytest
import torch

from oracle import krs_oracle as ko
from tests.helpers import make_bags, to_f32, to_np

pytestmark = pytest.mark.gpu
TORCH_DT = {"f32": torch.float32, "bf16": torch.bfloat16}


def _setup(dim, tdt, gdt, csr, use_w, combiners, n_tables=3, batch=41, max_hot=6, seed=0, shared=True,
           vocab_hi=60):
    from keras_rs_amd.embedding_ops import FusedBags

    rng = np.random.default_rng(seed)
    dev = torch.device("cuda:0")
    vocabs_t = [int(v) for v in rng.integers(3, vocab_hi, size=n_tables)]
    tables = [torch.from_numpy(rng.uniform(-1, 1, (v, dim)).astype(np.float32)).to(TORCH_DT[tdt]).to(dev)
              for v in vocabs_t]
    tix = list(range(n_tables)) + ([0] if shared else [])
    n_feats = len(tix)
    specs = [(tix[f], combiners[f % len(combiners)], 2 + f * dim) for f in range(n_feats)]
    cols = 2 + n_feats * dim + 3
    bags = make_bags(rng, n_feats, batch, [vocabs_t[t] for t in tix], max_hot, csr)
    grad = torch.from_numpy(rng.uniform(0, 1, (batch, cols)).astype(np.float32)).to(TORCH_DT[gdt]).to(dev)
    slots = [torch.full(t.shape, 0.1, dtype=torch.float32, device=dev) for t in tables]
    fb = FusedBags(tables, specs, slots=slots, lrs=[0.01 * (i + 1) for i in range(n_tables)])
    ids = torch.from_numpy(bags["ids"]).to(dev)
    offs = None if bags["offsets"] is None else torch.from_numpy(bags["offsets"]).to(dev)
    w = torch.from_numpy(bags["weights"]).to(dev) if use_w else None
    out = torch.empty((batch, cols), dtype=TORCH_DT[tdt], device=dev)
    _, scale = fb.forward(ids, batch, hots=bags["hots"], offsets=offs, weights=w, out=out, want_scale=True)
    ws = fb.plan_backward(ids, batch, hots=bags["hots"], offsets=offs)

    # oracle dense gradient
    feats_np = ko.make_features([t for t, _, _ in specs], [c for _, c, _ in specs], [c for _, _, c in specs],
                                hots=bags["hots"], batch=batch)
    de = [np.zeros((v, dim), np.float32) for v in vocabs_t]
    ko.embed_bag_bwd_dense(ko.make_tables(de), feats_np, bags["ids"], bags["offsets"],
                           bags["weights"] if use_w else None, scale.cpu().numpy(), to_np(grad), batch, dim)
    return dict(fb=fb, ws=ws, grad=grad, batch=batch, nnz=bags["nnz"], hots=bags["hots"], w=w, scale=scale,
                de=de, tables=tables, slots=slots, bags=bags, vocabs=vocabs_t)


@pytest.mark.parametrize("dim", [6, 7, 32, 64, 128, 256])
@pytest.mark.parametrize("tdt,gdt", [("f32", "f32"), ("bf16", "bf16")])
@pytest.mark.parametrize("csr", [False, True])
def test_dense_gradient(dim, tdt, gdt, csr):
    s = _setup(dim, tdt, gdt, csr, use_w=True, combiners=["sum", "mean", "sqrtn"])
    got = s["fb"].backward_dense(s["ws"], s["grad"], s["batch"], s["nnz"], hots=s["hots"], weights=s["w"],
                                 bag_scale=s["scale"])
    torch.cuda.synchronize()
    for g, e in zip(got, s["de"]):
        # same summation order (ascending position) and fmaf: agreement far inside 1e-5
        np.testing.assert_allclose(g.cpu().numpy(), e, rtol=1e-6, atol=1e-6)


def test_dense_gradient_is_deterministic_and_handles_heavy_duplicates():
    s = _setup(128, "f32", "f32", False, use_w=False, combiners=["sum"], n_tables=2, batch=300, max_hot=40,
               vocab_hi=8)
    a = s["fb"].backward_dense(s["ws"], s["grad"], s["batch"], s["nnz"], hots=s["hots"], bag_scale=s["scale"])
    b = s["fb"].backward_dense(s["ws"], s["grad"], s["batch"], s["nnz"], hots=s["hots"], bag_scale=s["scale"])
    for x, y, e in zip(a, b, s["de"]):
        assert torch.equal(x, y)
        np.testing.assert_allclose(x.cpu().numpy(), e, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("dim", [20, 128])
def test_sparse_gradient(dim):
    s = _setup(dim, "f32", "f32", True, use_w=True, combiners=["mean"])
    rows, vals = s["fb"].backward_sparse(s["ws"], s["grad"], s["batch"], s["nnz"], hots=s["hots"],
                                         weights=s["w"], bag_scale=s["scale"])
    rows = rows.cpu().numpy()
    vals = vals.cpu().numpy()
    assert np.all(np.diff(rows) > 0)  # unique, ascending global rows
    dense = np.concatenate(s["de"], axis=0)
    touched = np.zeros(dense.shape[0], bool)
    touched[rows] = True
    np.testing.assert_allclose(vals, dense[rows], rtol=1e-6, atol=1e-6)
    assert np.all(dense[~touched] == 0)


@pytest.mark.parametrize("kind", ["sgd", "adagrad"])
@pytest.mark.parametrize("tdt,gdt,dim", [("f32", "f32", 128), ("bf16", "bf16", 128), ("f32", "f32", 7),
                                         ("bf16", "f32", 64), ("f32", "bf16", 64)])
def test_fused_optimizers(kind, tdt, gdt, dim):
    s = _setup(dim, tdt, gdt, False, use_w=True, combiners=["sum", "mean"])
    exp_tables = [to_np(t).copy() for t in s["tables"]]
    exp_slots = [x.cpu().numpy().copy() for x in s["slots"]]
    s["fb"].backward_fused(kind, s["ws"], s["grad"], s["batch"], s["nnz"], hots=s["hots"], weights=s["w"],
                           bag_scale=s["scale"])
    torch.cuda.synchronize()
    # which rows were looked up (per table)
    ids, hots, bags = s["bags"]["ids"], s["hots"], s["bags"]
    for t in range(len(exp_tables)):
        touched = np.zeros(s["vocabs"][t], np.uint8)
        base = 0
        for f, (tt, _, _) in enumerate(s["fb"].features):
            n = s["batch"] * hots[f]
            if tt == t:
                touched[ids[base:base + n]] = 1
            base += n
        ko.apply_optimizer(exp_tables[t], exp_slots[t], s["de"][t], touched, s["fb"].lrs[t], kind)
        got = to_np(s["tables"][t])
        if tdt == "bf16":
            np.testing.assert_allclose(to_f32(got), to_f32(exp_tables[t]), rtol=2 ** -7, atol=1e-6)
        else:
            np.testing.assert_allclose(got, exp_tables[t], rtol=1e-6, atol=1e-6)
        if kind == "adagrad":
            np.testing.assert_allclose(s["slots"][t].cpu().numpy(), exp_slots[t], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("tdt,gdt,dim", [("f32", "f32", 128), ("bf16", "bf16", 128), ("f32", "f32", 7),
                                         ("bf16", "f32", 64), ("f32", "bf16", 96)])
@pytest.mark.parametrize("heavy", [False, True])
def test_fused_rowwise_adagrad(tdt, gdt, dim, heavy):
    # opt-in variant (not a reference optimizer): ONE accumulator per row = running mean over the columns of g^2.
    # Checked against the oracle twin on the oracle's dense gradient; `heavy`: segments longer than 128 / 2048
    # lookups take the workgroup-per-row and the chunked paths
    kw = dict(n_tables=2, batch=700, max_hot=40, vocab_hi=6) if heavy else {}
    s = _setup(dim, tdt, gdt, False, use_w=not heavy, combiners=["sum", "mean"], **kw)
    dev = s["grad"].device
    fb = s["fb"]
    fb.slots = [torch.full((t.shape[0],), 0.1, dtype=torch.float32, device=dev) for t in s["tables"]]
    fb._tab_key = None
    exp_tables = [to_np(t).copy() for t in s["tables"]]
    exp_acc = [np.full(t.shape[0], 0.1, np.float32) for t in s["tables"]]
    for step in range(2):
        fb.backward_fused("adagrad_rowwise", s["ws"], s["grad"], s["batch"], s["nnz"], hots=s["hots"], weights=s["w"],
                          bag_scale=s["scale"])
    torch.cuda.synchronize()
    ids, hots = s["bags"]["ids"], s["hots"]
    n_piece = 8 if gdt == "bf16" else 4
    for t in range(len(exp_tables)):
        touched = np.zeros(s["vocabs"][t], np.uint8)
        base = 0
        for f, (tt, _, _) in enumerate(fb.features):
            n = s["batch"] * hots[f]
            if tt == t:
                touched[ids[base:base + n]] = 1
            base += n
        for step in range(2):
            ko.apply_optimizer(exp_tables[t], exp_acc[t], s["de"][t], touched, fb.lrs[t], "adagrad_rowwise",
                               (n_piece, 0, 0, 0))
        tol = dict(rtol=2 ** -7, atol=1e-6) if tdt == "bf16" else dict(rtol=2e-5 if heavy else 1e-6, atol=1e-6)
        np.testing.assert_allclose(to_f32(to_np(s["tables"][t])), to_f32(exp_tables[t]), **tol)
        np.testing.assert_allclose(fb.slots[t].cpu().numpy(), exp_acc[t], rtol=1e-5 if heavy else 1e-6, atol=1e-7)
        assert np.all(fb.slots[t].cpu().numpy()[touched == 0] == np.float32(0.1))


def test_rowwise_adagrad_equals_exact_adagrad_when_columns_agree():
    # relation to the exact rule: where |g| is the same in every column of a row, mean_j g_j^2 = g_j^2 and both
    # optimizers take the same step
    import keras_rs_amd.layers as kl

    rng = np.random.default_rng(2)
    ids = rng.integers(0, 30, (16, 3)).astype(np.int32)
    sign = torch.from_numpy(rng.choice([-1.0, 1.0], (16, 8)).astype(np.float32)).cuda()
    mag = torch.from_numpy(rng.uniform(0.5, 2, (16, 1)).astype(np.float32)).cuda()
    res = []
    for opt in (kl.Adagrad(0.1, 0.1), kl.RowwiseAdagrad(0.1, 0.1)):
        t = kl.TableConfig("t", 30, 8, placement="sparsecore", optimizer=opt, combiner="sum",
                           initializer=kl.distributed_embedding.base.RandomUniform(-1, 1, seed=3))
        layer = kl.DistributedEmbedding({"a": kl.FeatureConfig("a", t, (16, 3), (16, 8))})
        (layer({"a": ids})["a"] * sign * mag).sum().backward()
        res.append(layer.get_embedding_tables()["t"].clone())
        sd = layer.state_dict()
        assert sd["sparsecore_t_slot"].shape == ((30,) if isinstance(opt, kl.RowwiseAdagrad) else (30, 8))
    # rows hit by several samples see sums of +-mag with differing magnitudes per column: compare the others
    counts = np.bincount(ids.reshape(-1), minlength=30)
    once = torch.from_numpy(np.nonzero(counts == 1)[0]).cuda()
    assert once.numel() > 3
    torch.testing.assert_close(res[0][once], res[1][once], rtol=1e-6, atol=1e-6)


ADAM = (0.9, 0.999, 1e-7)
FTRL = (-0.5, 0.02, 0.01, 0.3)   # learning_rate_power, l1, l2, beta


@pytest.mark.parametrize("kind", ["adam", "ftrl"])
@pytest.mark.parametrize("tdt,gdt,dim", [("f32", "f32", 128), ("bf16", "bf16", 128), ("f32", "f32", 7),
                                         ("f32", "bf16", 64)])
def test_fused_adam_and_ftrl_two_steps(kind, tdt, gdt, dim):
    # SURVEY.md section 8f.2.  Two consecutive updates (Adam's bias correction depends on the step);
    # untouched rows must keep value and slots (lazy semantics).
    s = _setup(dim, tdt, gdt, False, use_w=True, combiners=["sum", "mean"])
    fb = s["fb"]
    dev = s["tables"][0].device
    fb.slots = [torch.zeros((2,) + tuple(t.shape), dtype=torch.float32, device=dev) for t in s["tables"]]
    if kind == "ftrl":
        for sl in fb.slots:
            sl[0].fill_(0.1)
    exp_tables = [to_np(t).copy() for t in s["tables"]]
    exp_slots = [x.cpu().numpy().copy() for x in fb.slots]
    ids, hots = s["bags"]["ids"], s["hots"]
    for step in (1, 2):
        hyper = ADAM + (float(np.sqrt(1 - ADAM[1] ** step) / (1 - ADAM[0] ** step)),) if kind == "adam" else FTRL
        fb.backward_fused(kind, s["ws"], s["grad"], s["batch"], s["nnz"], hots=hots, weights=s["w"],
                          bag_scale=s["scale"], hyper=hyper)
        torch.cuda.synchronize()
        for t in range(len(exp_tables)):
            touched = np.zeros(s["vocabs"][t], np.uint8)
            base = 0
            for f, (tt, _, _) in enumerate(fb.features):
                n = s["batch"] * hots[f]
                if tt == t:
                    touched[ids[base:base + n]] = 1
                base += n
            before = exp_tables[t].copy()
            ko.apply_optimizer(exp_tables[t], exp_slots[t], s["de"][t], touched, fb.lrs[t], kind, hyper)
            assert np.array_equal(before[touched == 0], exp_tables[t][touched == 0])
            got = to_np(s["tables"][t])
            if tdt == "bf16":
                np.testing.assert_allclose(to_f32(got), to_f32(exp_tables[t]), rtol=2 ** -7, atol=1e-5)
                exp_tables[t] = got.copy()  # follow the device's rounding into the next step
            else:
                np.testing.assert_allclose(got, exp_tables[t], rtol=2e-5, atol=2e-6)
            # FTRL's linear term carries (n'^-p - n^-p) / lr * w: a difference of close numbers times 1 / lr
            np.testing.assert_allclose(fb.slots[t].cpu().numpy(), exp_slots[t], rtol=2e-5, atol=2e-5 if kind == "ftrl" else 1e-6)
    with pytest.raises(Exception, match="hyper"):
        fb.backward_fused(kind, s["ws"], s["grad"], s["batch"], s["nnz"], hots=hots)


@pytest.mark.parametrize("tdt,gdt,dim,batch", [("f32", "f32", 128, 4000), ("bf16", "bf16", 64, 4000),
                                               ("f32", "f32", 32, 30000)])
def test_hot_rows_take_the_workgroup_path(tdt, gdt, dim, batch):
    """Tiny vocabularies / skewed ids: segments longer than kLongSeg (128) are summed by whole workgroups
    in chunks of kChunk (2048) lookups (bag_apply_long_kernel), rows spanning several chunks are
    finished from their partial rows in chunk order (bag_apply_finish_kernel; batch 30000 gives rows
    with ~10-40 chunks); dense, sparse and fused forms must still match the oracle."""
    s = _setup(dim, tdt, gdt, False, use_w=True, combiners=["sum", "mean"], n_tables=2, batch=batch, max_hot=5,
               vocab_hi=6, shared=True)
    fb = s["fb"]
    dense = fb.backward_dense(s["ws"], s["grad"], s["batch"], s["nnz"], hots=s["hots"], weights=s["w"],
                              bag_scale=s["scale"])
    again = fb.backward_dense(s["ws"], s["grad"], s["batch"], s["nnz"], hots=s["hots"], weights=s["w"],
                              bag_scale=s["scale"])
    for g, g2, e in zip(dense, again, s["de"]):
        assert torch.equal(g, g2)  # still deterministic
        np.testing.assert_allclose(g.cpu().numpy(), e, rtol=2e-5 * (batch / 4000), atol=(1e-3 if gdt == "bf16" else 2e-4) * (batch / 4000))
    rows, vals = fb.backward_sparse(s["ws"], s["grad"], s["batch"], s["nnz"], hots=s["hots"], weights=s["w"],
                                    bag_scale=s["scale"])
    np.testing.assert_allclose(vals.cpu().numpy(), np.concatenate(s["de"], 0)[rows.cpu().numpy()],
                               rtol=2e-5 * (batch / 4000), atol=(1e-3 if gdt == "bf16" else 2e-4) * (batch / 4000))
    exp_tables = [to_np(t).copy() for t in s["tables"]]
    exp_slots = [x.cpu().numpy().copy() for x in s["slots"]]
    fb.backward_fused("adagrad", s["ws"], s["grad"], s["batch"], s["nnz"], hots=s["hots"], weights=s["w"],
                      bag_scale=s["scale"])
    for t in range(len(exp_tables)):
        ko.apply_optimizer(exp_tables[t], exp_slots[t], s["de"][t], None, fb.lrs[t], "adagrad")
        np.testing.assert_allclose(to_f32(to_np(s["tables"][t])), to_f32(exp_tables[t]),
                                   rtol=2 ** -7 if tdt == "bf16" else 1e-5, atol=1e-5)
        np.testing.assert_allclose(s["slots"][t].cpu().numpy(), exp_slots[t], rtol=1e-4, atol=1e-3)
    # Adam through the same hot-row path (two slot planes)
    fb.slots = [torch.zeros((2,) + tuple(t.shape), dtype=torch.float32, device=t.device) for t in s["tables"]]
    exp_tables = [to_np(t).copy() for t in s["tables"]]
    exp_slots = [x.cpu().numpy().copy() for x in fb.slots]
    hyper = ADAM + (float(np.sqrt(1 - ADAM[1]) / (1 - ADAM[0])),)
    fb.backward_fused("adam", s["ws"], s["grad"], s["batch"], s["nnz"], hots=s["hots"], weights=s["w"],
                      bag_scale=s["scale"], hyper=hyper)
    for t in range(len(exp_tables)):
        ko.apply_optimizer(exp_tables[t], exp_slots[t], s["de"][t], None, fb.lrs[t], "adam", hyper)
        np.testing.assert_allclose(to_f32(to_np(s["tables"][t])), to_f32(exp_tables[t]),
                                   rtol=2 ** -7 if tdt == "bf16" else 1e-4, atol=1e-4)


@pytest.mark.parametrize("kind", ["sgd", "adagrad", "adam", "adagrad_rowwise"])
@pytest.mark.parametrize("tdt,gdt,dim", [("bf16", "bf16", 128), ("f32", "f32", 64), ("bf16", "f32", 32), ("f32", "bf16", 24)])
def test_apply_kernel_gives_the_same_bits_on_misaligned_table_and_slot_bases(kind, tdt, gdt, dim):
    """bag_apply_fast_kernel has ONE access form -- under-aligned wide vector accesses, the memory pipeline runs in
    unaligned-access mode -- so table / slot base pointers that are NOT 16-byte aligned (2 or 4 bytes off) must give the
    bits of aligned ones: segments of 1 .. ~40 lookups, weights and bag scales, every fused optimizer.  (Round 3 / 4 also
    compared it bit for bit with the round-1 kernel behind KRS_EMBED_OPT_APPLY; that kernel was deleted in round 5.  The
    bf16-table / fp32-gradient pair runs bag_apply_generic.)"""
    from keras_rs_amd.embedding_ops import FusedBags

    rng = np.random.default_rng(11)
    dev = torch.device("cuda:0")
    batch, n_tables = 257, 3
    vocabs = [37, 900, 5]
    hots = [3, 9, 2, 1]
    tix = [0, 1, 2, 1]

    def alloc(shape, dt, fill, misaligned):
        # a view that starts one element into a larger buffer when `misaligned` (2 or 4 bytes off 16-byte alignment)
        n = int(np.prod(shape))
        big = torch.empty(n + 8, dtype=dt, device=dev)
        v = big[1:1 + n].view(shape) if misaligned else big[:n].view(shape)
        v.copy_(fill)
        return v

    def build(misaligned):
        g = np.random.default_rng(5)
        tables = [alloc((v, dim), TORCH_DT[tdt], torch.from_numpy(g.uniform(-1, 1, (v, dim)).astype(np.float32)).to(dev),
                        misaligned) for v in vocabs]
        planes = {"sgd": None, "adagrad": 1, "adam": 2, "adagrad_rowwise": 0}[kind]
        if planes is None:
            slots = [None] * n_tables
        elif planes == 0:
            slots = [alloc((v,), torch.float32, torch.full((v,), 0.1, device=dev), misaligned) for v in vocabs]
        elif planes == 1:
            slots = [alloc((v, dim), torch.float32, torch.full((v, dim), 0.1, device=dev), misaligned) for v in vocabs]
        else:
            slots = [alloc((2, v, dim), torch.float32, torch.zeros((2, v, dim), device=dev), misaligned) for v in vocabs]
        fb = FusedBags(tables, [(tix[f], ["sum", "mean", "sqrtn", "sum"][f], 3 + f * dim) for f in range(4)],
                       slots=slots, lrs=[0.01, 0.02, 0.03])
        return tables, slots, fb

    ids = torch.from_numpy(np.concatenate([rng.integers(0, vocabs[tix[f]], batch * hots[f]) for f in range(4)]
                                          ).astype(np.int32)).to(dev)
    w = torch.from_numpy(rng.uniform(0.1, 1, ids.numel()).astype(np.float32)).to(dev)
    cols = 3 + 4 * dim + 1
    grad = torch.from_numpy(rng.uniform(-1, 1, (batch, cols)).astype(np.float32)).to(TORCH_DT[gdt]).to(dev)
    res = []
    for misaligned in (False, True):
        tables, slots, fb = build(misaligned)
        before = [t.clone() for t in tables]
        out = torch.empty((batch, cols), dtype=TORCH_DT[tdt], device=dev)
        _, scale = fb.forward(ids, batch, hots=hots, weights=w, out=out, want_scale=True)
        ws = fb.plan_backward(ids, batch, hots=hots, global_order=False)
        hyper = (0.9, 0.999, 1e-7, 0.3) if kind == "adam" else None
        fb.backward_fused(kind, ws, grad, batch, ids.numel(), hots=hots, weights=w, bag_scale=scale, hyper=hyper)
        torch.cuda.synchronize()
        assert any(not torch.equal(x, y) for x, y in zip(before, tables)), "the update did not run"
        res.append(([t.clone() for t in tables], [None if x is None else x.clone() for x in slots]))
    for a, b in zip(res[0][0], res[1][0]):
        assert torch.equal(a, b)
    for a, b in zip(res[0][1], res[1][1]):
        assert (a is None and b is None) or torch.equal(a, b)


def test_more_features_than_the_descriptor_cache_take_the_any_shape_kernel():
    """The vector apply kernels cache feature / table descriptors in LDS (512 of each); a wider model -- the reference
    has no limit on the number of features -- runs bag_apply_generic: same dense gradient and fused SGD update as the
    oracle."""
    s = _setup(8, "f32", "f32", False, use_w=True, combiners=["sum", "mean", "sqrtn"], n_tables=520, batch=7, max_hot=3,
               shared=True, vocab_hi=12)
    assert len(s["fb"].features) > 512
    got = s["fb"].backward_dense(s["ws"], s["grad"], s["batch"], s["nnz"], hots=s["hots"], weights=s["w"],
                                 bag_scale=s["scale"])
    torch.cuda.synchronize()
    for g, e in zip(got, s["de"]):
        np.testing.assert_allclose(g.cpu().numpy(), e, rtol=1e-6, atol=1e-6)
    before = [t.clone() for t in s["tables"]]
    s["fb"].backward_fused("sgd", s["ws"], s["grad"], s["batch"], s["nnz"], hots=s["hots"], weights=s["w"],
                           bag_scale=s["scale"])
    torch.cuda.synchronize()
    for t, (b, e) in enumerate(zip(before, s["de"])):
        exp = b.cpu().numpy() - np.float32(s["fb"].lrs[t]) * e
        np.testing.assert_allclose(s["tables"][t].cpu().numpy(), exp, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("csr", [False, True])
def test_thousands_of_out_of_range_ids_are_dropped(csr):
    """A trailing run of invalid keys longer than one hot-row chunk (2048 lookups): it is listed as a multi-chunk
    segment and must be skipped by every apply kernel (it used to reach bag_apply_finish_kernel, which named a table
    row from the invalid key).  CSR form with lookups behind the last bag as well (the padded tail of a
    static-capacity exchange: positions no bag covers, whose plan values are never written)."""
    from keras_rs_amd.embedding_ops import FusedBags

    dev = torch.device("cuda:0")
    rng = np.random.default_rng(3)
    V, D, B, hot = 300, 64, 2048, 4
    table = torch.from_numpy(rng.uniform(-1, 1, (V, D)).astype(np.float32)).to(dev)
    slot = torch.full((V, D), 0.1, device=dev)
    ids_np = rng.integers(0, V, B * hot).astype(np.int32)
    bad = rng.permutation(B * hot)[:5000]
    ids_np[bad] = -1 - (bad % 7)                        # 5000 out-of-range ids: one invalid run of > 2 chunks
    grad = torch.from_numpy(rng.uniform(-1, 1, (B, D)).astype(np.float32)).to(dev)
    fb = FusedBags([table], [(0, "sum", 0)], slots=[slot], lrs=[0.05])
    if csr:
        pad = np.full(3000, -1, np.int32)               # behind the last bag
        ids = torch.from_numpy(np.concatenate([ids_np, pad])).to(dev)
        offsets = torch.arange(0, B * hot + 1, hot, dtype=torch.int32, device=dev)
        ws = fb.plan_backward(ids, B, offsets=offsets)
        fb.backward_fused("adagrad", ws, grad, B, ids.numel())
    else:
        ids = torch.from_numpy(ids_np).to(dev)
        ws = fb.plan_backward(ids, B, hots=[hot])
        fb.backward_fused("adagrad", ws, grad, B, ids.numel(), hots=[hot])
    torch.cuda.synchronize()
    # oracle: the same bags with the invalid lookups removed (weight 0 contributes nothing; dense gradient + Adagrad)
    w = (ids_np >= 0).astype(np.float32)
    safe = np.where(ids_np >= 0, ids_np, 0).astype(np.int32)
    dense = np.zeros((V, D), np.float32)
    f = ko.make_features([0], ["sum"], [0], hots=[hot], batch=B)
    ko.embed_bag_bwd_dense(ko.make_tables([dense]), f, safe, None, w, None, grad.cpu().numpy(), B, D)
    exp_t = rng.uniform(0, 0, (V, D)).astype(np.float32)  # placeholder, replaced below
    rng2 = np.random.default_rng(3)
    exp_t = rng2.uniform(-1, 1, (V, D)).astype(np.float32)
    acc = np.full((V, D), 0.1, np.float32)
    touched = np.zeros(V, np.uint8)
    touched[ids_np[ids_np >= 0]] = 1
    ko.apply_optimizer(exp_t, acc, dense, touched, 0.05, "adagrad")
    np.testing.assert_allclose(table.cpu().numpy(), exp_t, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(slot.cpu().numpy(), acc, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("vocabs,dim", [([37, 900, 5, 70000], 64),        # 17 key bits: two passes
                                        ([200, 31], 32),                   # 8 bits: one pass (first = last)
                                        ([1_200_000, 3, 2_100_000], 8)])   # 22 bits: three passes
@pytest.mark.parametrize("with_bad", [False, True])
def test_table_segmented_plan_equals_global_plan(vocabs, dim, with_bad):
    """krs_embed_bag_bwd_plan_tables (per-table sort on the id, 8-byte intermediate pairs, final pass rebuilds the
    64-bit values) against the global sort (KRS_EMBED_OPT_PLAN = 1): the fused Adagrad update and the dense gradient
    they lead to must be bit-identical -- problems that end in partial tiles, two features on one table (neighbours),
    1 / 2 / 3 passes, out-of-range ids (which end their table's run instead of the array)."""
    import ctypes as C

    from keras_rs_amd import _lib as L
    from keras_rs_amd.embedding_ops import FusedBags

    rng = np.random.default_rng(5)
    dev = torch.device("cuda:0")
    batch = 1031
    tix = [0, 0] + list(range(1, len(vocabs)))             # table 0 has two neighbouring features
    hots = [3, 1] + [7, 2, 5][: len(vocabs) - 1]
    ids_np = np.concatenate([rng.integers(0, vocabs[tix[f]], batch * hots[f]) for f in range(len(tix))]).astype(np.int32)
    if with_bad:
        bad = rng.permutation(len(ids_np))[:300]
        ids_np[bad] = np.where(bad % 2 == 0, -5, 2_000_000_000)
    ids = torch.from_numpy(ids_np).to(dev)
    cols = len(tix) * dim
    grad = torch.from_numpy(rng.uniform(-1, 1, (batch, cols)).astype(np.float32)).to(dev)
    w = torch.from_numpy(rng.uniform(0.1, 1, len(ids_np)).astype(np.float32)).to(dev)
    res = []
    try:
        for variant in (0, 1):
            L.check(L.lib().krs_embed_set_option(C.c_int(2), C.c_int(variant)), "krs_embed_set_option")
            g = np.random.default_rng(9)
            tables = [torch.from_numpy(g.uniform(-1, 1, (v, dim)).astype(np.float32)).to(dev) for v in vocabs]
            slots = [torch.full((v, dim), 0.1, device=dev) for v in vocabs]
            fb = FusedBags(tables, [(tix[f], "sum", f * dim) for f in range(len(tix))], slots=slots,
                           lrs=[0.01 * (t + 1) for t in range(len(vocabs))])
            err = torch.zeros(1, dtype=torch.int32, device=dev)
            ws = fb.plan_backward(ids, batch, hots=hots, err_flag=err, global_order=False)
            assert bool(int(err.item()) & 1) == with_bad
            dense = fb.backward_dense(ws, grad, batch, ids.numel(), hots=hots, weights=w)
            fb.backward_fused("adagrad", ws, grad, batch, ids.numel(), hots=hots, weights=w)
            torch.cuda.synchronize()
            res.append((tables, slots, dense))
    finally:
        L.lib().krs_embed_set_option(C.c_int(2), C.c_int(0))
    for part in range(3):
        for a, b in zip(res[0][part], res[1][part]):
            assert torch.equal(a, b)
    assert not torch.equal(res[0][0][0], torch.from_numpy(np.random.default_rng(9).uniform(-1, 1, (vocabs[0], dim)).astype(np.float32)).to(dev))


def test_compact_form_refuses_a_table_segmented_plan_and_default_plan_serves_it():
    """the table-segmented sort leaves out-of-range ids at the end of every TABLE's run, the compact
    (sparse) form counts ONE trailing invalid run.  `plan_backward` defaults to the global sort (every apply form accepts
    it); a plan made with global_order=False is refused by backward_sparse (n_unique = -1 from the mode word inside the
    workspace -- also when the workspace was copied elsewhere; the round-3 host registry of workspace addresses is gone,
    ADVICE r4: a re-used address could refuse a good plan)."""
    from keras_rs_amd import _lib as L
    from keras_rs_amd.embedding_ops import FusedBags

    rng = np.random.default_rng(11)
    dev = torch.device("cuda:0")
    batch, dim, vocabs, hots = 257, 16, [50, 70, 30], [3, 2, 4]
    ids_np = np.concatenate([rng.integers(0, vocabs[f], batch * hots[f]) for f in range(3)]).astype(np.int32)
    ids_np[5] = -1                       # out of range in the FIRST table (not the last run of the array)
    ids_np[batch * 3 + 7] = 10_000       # ... and in the second
    ids = torch.from_numpy(ids_np).to(dev)
    tables = [torch.from_numpy(rng.uniform(-1, 1, (v, dim)).astype(np.float32)).to(dev) for v in vocabs]
    fb = FusedBags(tables, [(f, "sum", f * dim) for f in range(3)])
    grad = torch.from_numpy(rng.uniform(-1, 1, (batch, 3 * dim)).astype(np.float32)).to(dev)
    ws = fb.plan_backward(ids, batch, hots=hots)                       # default: global order
    rows, vals = fb.backward_sparse(ws, grad, batch, ids.numel(), hots=hots)
    dense = fb.backward_dense(ws, grad, batch, ids.numel(), hots=hots)
    full = torch.cat(dense, dim=0)
    touched = torch.zeros(sum(vocabs), dtype=torch.bool, device=dev)
    touched[rows] = True
    assert rows.numel() == int((full.abs().sum(1) > 0).sum()) and torch.equal(full[rows], vals)
    assert not bool(full[~touched].any())
    ws_t = fb.plan_backward(ids, batch, hots=hots, global_order=False)  # table-segmented
    d2 = fb.backward_dense(ws_t, grad, batch, ids.numel(), hots=hots)
    for a, b in zip(dense, d2):
        assert torch.equal(a, b)
    with pytest.raises(L.KrsError):
        fb.backward_sparse(ws_t, grad, batch, ids.numel(), hots=hots)
    moved = ws_t.clone()                                                 # the fact travels with the bytes
    with pytest.raises(L.KrsError):
        fb.backward_sparse(moved, grad, batch, ids.numel(), hots=hots)
    ws_t.copy_(ws)                       # a GLOBAL plan copied over the address that held the table-segmented one is served
    rows2, vals2 = fb.backward_sparse(ws_t, grad, batch, ids.numel(), hots=hots)
    assert torch.equal(rows2, rows) and torch.equal(vals2, vals)
