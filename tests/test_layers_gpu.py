"""The reference's own layer tests, replayed on the MI355X layers (HIP through the C ABI):
known-answer vectors (tests/golden/kat.json), gradients against an independent plain-torch
composition of the same formulas, DistributedEmbedding correctness / shared tables / one
training step."""

import json
import math
import os

import numpy as np
import pytest
import torch

from tests.helpers import to_f32, to_np

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat.json")))
TOL = dict(atol=1e-6, rtol=1e-6)  # keras_rs/src/testing/test_case.py:47-77


def _layers():
    import keras_rs_amd.layers as kl

    return kl


@pytest.mark.parametrize("case", KAT["feature_cross"]["cases"], ids=lambda c: c["name"])
def test_feature_cross_kat(case):
    kl = _layers()
    fc = KAT["feature_cross"]
    x0 = torch.tensor(fc["x0"], device=DEV)
    x = torch.tensor(fc["x"], device=DEV)
    layer = kl.FeatureCross(projection_dim=case["projection_dim"], diag_scale=case["diag_scale"],
                            kernel_initializer="ones")
    out = layer(x0) if case["one_input"] else layer(x0, x)
    np.testing.assert_allclose(out.detach().cpu().numpy(), np.array(case["expected"], np.float32), **TOL)
    assert [list(w.shape) for w in layer.weights] == case["weight_shapes"]


def test_feature_cross_pre_activation_callable():
    kl = _layers()
    fc = KAT["feature_cross"]
    x0 = torch.tensor(fc["x0"], device=DEV)
    x = torch.tensor(fc["x"], device=DEV)
    out = kl.FeatureCross(pre_activation=torch.zeros_like)(x0, x)  # feature_cross_test.py:75-79
    np.testing.assert_allclose(out.detach().cpu().numpy(), np.array(fc["pre_activation_zero"]["expected"]), **TOL)


def _torch_cross(x0, x, down, kernel, bias, diag, act):
    h = x if down is None else x @ down
    z = h @ kernel + (0 if bias is None else bias)
    u = {None: lambda t: t, "relu": torch.relu, "sigmoid": torch.sigmoid, "tanh": torch.tanh}[act](z)
    return x0 * (u + diag * x) + x


@pytest.mark.parametrize("p", [None, 8])
@pytest.mark.parametrize("act", [None, "relu", "tanh", "sigmoid"])
@pytest.mark.parametrize("shape", [(37, 24), (4, 5, 16)])
def test_feature_cross_forward_and_gradients_fp32(p, act, shape):
    kl = _layers()
    g = torch.Generator().manual_seed(0)
    x0 = torch.randn(shape, generator=g).to(DEV).requires_grad_()
    x = torch.randn(shape, generator=g).to(DEV).requires_grad_()
    layer = kl.FeatureCross(projection_dim=p, diag_scale=0.3, pre_activation=act, bias_initializer="uniform")
    y = layer(x0, x)
    gy = torch.randn(shape, generator=g).to(DEV)
    y.backward(gy)
    got = [x0.grad, x.grad] + [w.grad for w in layer.weights]
    # independent composition on fp64 copies
    r0, r = x0.detach().double().requires_grad_(), x.detach().double().requires_grad_()
    ws = [w.detach().double().requires_grad_() for w in layer.weights]
    down, kern, bias = (ws[0], ws[1], ws[2]) if p is not None else (None, ws[0], ws[1])
    ry = _torch_cross(r0, r, down, kern, bias, 0.3, act)
    ry.backward(gy.double())
    np.testing.assert_allclose(y.detach().cpu().numpy(), ry.detach().cpu().numpy(), rtol=1e-5, atol=1e-5)
    for a, b in zip(got, [r0.grad, r.grad] + [w.grad for w in ws]):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=2e-5, atol=2e-5)


def test_feature_cross_same_tensor_twice_and_dcn_stack():
    kl = _layers()
    g = torch.Generator().manual_seed(1)
    x0 = torch.randn(33, 32, generator=g).to(DEV).requires_grad_()
    layers = [kl.FeatureCross(projection_dim=8) for _ in range(3)]
    xl = x0
    for layer in layers:  # DCNBlock.call, examples/ml_perf/model.py:332-336
        xl = layer(x0, xl)
    xl.sum().backward()
    r0 = x0.detach().double().requires_grad_()
    rl = r0
    for layer in layers:
        w = [t.detach().double() for t in layer.weights]
        rl = _torch_cross(r0, rl, w[0], w[1], w[2], 0.0, None)
    rl.sum().backward()
    np.testing.assert_allclose(xl.detach().cpu().numpy(), rl.detach().cpu().numpy(), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(x0.grad.cpu().numpy(), r0.grad.cpu().numpy(), rtol=5e-5, atol=5e-5)


def _stack_reference(layers, x0, top, diag, act):
    """float64 composition of `top(xl)` through the stack; returns d/dx0 and the weight gradients."""
    r0 = x0.detach().double().requires_grad_()
    ws = [[t.detach().double().requires_grad_() for t in layer.weights] for layer in layers]
    outs, rl = [], r0
    for w in ws:
        rl = _torch_cross(r0, rl, w[0], w[1], w[2], diag, act)
        outs.append(rl)
    top(outs).backward()
    return r0.grad, [[t.grad for t in w] for w in ws]


@pytest.mark.parametrize("shape", [(45, 40), (3, 7, 24)])
def test_cross_stack_hands_dx0_down_the_layers(shape):
    """A stack on one x0 passes dL/dx0 from layer to layer (autograd.Dx0Relay) instead of leaving three
    adds to autograd: same numbers as the float64 composition, also when an intermediate output has a
    second consumer, and a buffer left behind by a partial backward pass is not picked up later."""
    kl = _layers()
    g = torch.Generator().manual_seed(4)
    x0 = torch.randn(shape, generator=g).to(DEV).requires_grad_()
    layers = [kl.FeatureCross(projection_dim=8, diag_scale=0.2, pre_activation="tanh", bias_initializer="uniform")
              for _ in range(3)]
    gy = torch.randn(shape, generator=g).to(DEV)

    def run(top):
        x0.grad = None
        for layer in layers:
            for w in layer.weights:
                w.grad = None
        outs, xl = [], x0
        for layer in layers:
            xl = layer(x0, xl)
            outs.append(xl)
        return outs, top(outs)

    def check(top, rtol=5e-5):
        ref0, refw = _stack_reference(layers, x0, top, 0.2, "tanh")
        np.testing.assert_allclose(x0.grad.cpu().numpy(), ref0.cpu().numpy(), rtol=rtol, atol=rtol)
        for layer, rw in zip(layers, refw):
            for w, r in zip(layer.weights, rw):
                if r is not None:
                    np.testing.assert_allclose(w.grad.cpu().numpy(), r.cpu().numpy(), rtol=rtol, atol=rtol)

    plain = lambda outs: (outs[-1] * gy.to(outs[-1].dtype)).sum()                      # noqa: E731
    outs, loss = run(plain)
    assert all(hasattr(o, "_krs_dx0_relay") for o in outs)
    loss.backward()
    check(plain)
    # the middle output feeds the next layer AND the loss
    forked = lambda outs: (outs[-1] * gy.to(outs[-1].dtype)).sum() + (outs[1] ** 2).sum()  # noqa: E731
    outs, loss = run(forked)
    loss.backward()
    check(forked)
    # partial pass first (gradient w.r.t. the middle output only: the top layer leaves its dx0 term behind),
    # then the full pass on the same graph, then a pass that stops below the top layer
    outs, loss = run(plain)
    torch.autograd.grad(loss, [outs[1]], retain_graph=True)
    loss.backward(retain_graph=True)
    check(plain)
    outs, loss = run(plain)
    torch.autograd.grad(loss, [outs[1]], retain_graph=True)       # leaves a buffer on the middle layer's relay
    x0.grad = None
    for layer in layers:
        for w in layer.weights:
            w.grad = None
    (outs[1] * gy).sum().backward()                               # must not pick it up
    lower = lambda o: (o[1] * gy.to(o[1].dtype)).sum()            # noqa: E731
    ref0, _ = _stack_reference(layers[:2], x0, lower, 0.2, "tanh")
    np.testing.assert_allclose(x0.grad.cpu().numpy(), ref0.cpu().numpy(), rtol=5e-5, atol=5e-5)


def test_feature_cross_bf16_policy():
    kl = _layers()
    g = torch.Generator().manual_seed(2)
    x0 = torch.randn(256, 128, generator=g).to(DEV)
    layer = kl.FeatureCross(projection_dim=64, dtype="mixed_bfloat16")
    y = layer(x0.bfloat16(), x0.bfloat16())
    assert y.dtype == torch.bfloat16 and layer.weights[0].dtype == torch.float32
    w = [t.detach().bfloat16().float() for t in layer.weights[:2]] + [layer.weights[2].detach()]
    xb = x0.bfloat16().float()
    ref = _torch_cross(xb, xb, w[0], w[1], w[2], 0.0, None)
    np.testing.assert_allclose(y.detach().float().cpu().numpy(), ref.cpu().numpy(), rtol=2 ** -6, atol=3e-2)


@pytest.mark.parametrize("case", KAT["dot_interaction"]["cases"],
                         ids=lambda c: f"self{int(c['self_interaction'])}_skip{int(c['skip_gather'])}")
def test_dot_interaction_kat_and_grad(case):
    kl = _layers()
    feats = [torch.tensor(f, device=DEV, requires_grad=True) for f in KAT["dot_interaction"]["inputs"]]
    layer = kl.DotInteraction(case["self_interaction"], case["skip_gather"])
    out = layer(feats)
    np.testing.assert_allclose(out.detach().cpu().numpy(), np.array(case["expected"], np.float32), rtol=1e-6, atol=1e-5)
    assert tuple(out.shape) == layer.compute_output_shape([(1, 5)] * 3)
    w = torch.arange(1, out.numel() + 1, device=DEV, dtype=torch.float32).reshape(out.shape)
    (out * w).sum().backward()
    ref = [f.detach().double().requires_grad_() for f in feats]
    X = torch.stack(ref, 1)
    P = X @ X.transpose(1, 2)
    F = 3
    if case["skip_gather"]:
        mask = torch.tril(torch.ones(F, F, dtype=torch.float64, device=DEV), 0 if case["self_interaction"] else -1)
        r = (P * mask).reshape(1, F * F)
    else:
        idx = [i * F + j for i in range(F) for j in range(i + 1 if case["self_interaction"] else i)]
        r = P.reshape(1, F * F)[:, idx]
    (r * w.double()).sum().backward()
    for a, b in zip(feats, ref):
        np.testing.assert_allclose(a.grad.cpu().numpy(), b.grad.cpu().numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("case", KAT["embed_reduce"]["cases"],
                         ids=lambda c: f"{c['kind']}_{c['combiner']}_{'w' if c['weights'] else 'now'}")
def test_embed_reduce_kat(case):
    kl = _layers()
    er = KAT["embed_reduce"]
    layer = kl.EmbedReduce(er["input_dim"], er["output_dim"], combiner=case["combiner"])
    w = None if case["weights"] is None else np.array(case["weights"], np.float32)
    if case["kind"] == "bag":
        inputs = kl.Ragged(np.array(case["ids"], np.int32), np.array(case["offsets"], np.int32))
        res = layer(inputs, None if w is None else kl.Ragged(w, inputs.row_offsets))
    else:
        res = layer(torch.tensor(case["ids"], device=DEV), None if w is None else torch.tensor(w, device=DEV))
    assert tuple(res.shape) == (2, er["output_dim"])
    e = layer.embeddings.detach().cpu().numpy().astype(np.float64)
    exp = np.stack([sum(c * e[r] for r, c in terms) / div for terms, div in zip(case["terms"], case["divisor"])])
    np.testing.assert_allclose(res.detach().cpu().numpy(), exp.astype(np.float32), **TOL)


def test_embed_reduce_out_of_range_raises_and_dense_gradient():
    kl = _layers()
    layer = kl.EmbedReduce(10, 8, combiner="mean")
    with pytest.raises(IndexError):
        layer(torch.tensor([[1, 10]], device=DEV))
    ids = torch.tensor([[1, 2, 2], [3, 1, 1]], device=DEV)
    w = torch.tensor([[1.0, 2.0, 0.5], [1.0, 1.0, 3.0]], device=DEV)
    out = layer(ids, w)
    g = torch.randn(2, 8, device=DEV)
    out.backward(g)
    e = layer.embeddings.detach().double().requires_grad_()
    ref = (e[ids] * w.double()[..., None]).sum(1) / w.double().sum(1, keepdim=True)
    ref.backward(g.double())
    np.testing.assert_allclose(layer.embeddings.grad.cpu().numpy(), e.grad.cpu().numpy(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("combiner", ["sum", "mean", "sqrtn"])
def test_embed_reduce_weights_of_lower_and_higher_rank(combiner):
    # embed_reduce.py:181-190 accepts weights of any rank up to that of the embedded inputs and appends the
    # missing axes at the END (:244-248): [B] weights with [B, L] ids are one weight per row.  Expected values:
    # a float64 restatement of :253-274 on the expanded weights.
    kl = _layers()
    rng = np.random.default_rng(5)
    B, L, V, D = 6, 4, 11, 8
    layer = kl.EmbedReduce(V, D, combiner=combiner)
    ids = rng.integers(0, V, (B, L)).astype(np.int32)
    layer(torch.tensor(ids, device=DEV))
    e = layer.embeddings.detach().cpu().numpy().astype(np.float64)

    def ref(idv, w):
        x = e[idv]
        w = w.reshape(w.shape + (1,) * (x.ndim - w.ndim)).astype(np.float64)   # NOT broadcast: the divisor sums it as is
        if x.ndim <= 2:
            return x * w if combiner == "sum" else x
        s = (x * w).sum(-2)
        if combiner == "sum":
            return s
        d = w.sum(-2) if combiner == "mean" else np.sqrt((w * w).sum(-2))
        return np.where(d != 0, s / np.where(d != 0, d, 1), 0)

    w_row = rng.uniform(-1, 1, B).astype(np.float32)
    w_row[2] = 0.0                                                    # divide_no_nan
    got = layer(torch.tensor(ids, device=DEV), torch.tensor(w_row, device=DEV))
    np.testing.assert_allclose(got.detach().cpu().numpy(), ref(ids, w_row), rtol=1e-5, atol=1e-6)
    w_full = rng.uniform(0.1, 1, (B, L, D)).astype(np.float32)         # one weight per embedding column
    got = layer(torch.tensor(ids, device=DEV), torch.tensor(w_full, device=DEV))
    np.testing.assert_allclose(got.detach().cpu().numpy(), ref(ids, w_full), rtol=1e-5, atol=1e-6)
    ids1 = ids[:, 0].copy()
    w_bd = rng.uniform(0.1, 1, (B, D)).astype(np.float32)
    got = layer(torch.tensor(ids1, device=DEV), torch.tensor(w_bd, device=DEV))
    np.testing.assert_allclose(got.detach().cpu().numpy(), ref(ids1, w_bd), rtol=1e-5, atol=1e-6)
    with pytest.raises(ValueError):                                    # [L]-shaped weights do not match axis 0
        layer(torch.tensor(ids, device=DEV), torch.tensor(np.ones(L + 1, np.float32), device=DEV))


@pytest.mark.parametrize("placement", ["default_device", "sparsecore"])
def test_distributed_embedding_reports_out_of_range_ids_lazily(placement):
    # out-of-range ids are never clamped; the lookup contributes nothing and the layer
    # raises -- without a host sync in the step: at a later call once the flag has arrived, or on request
    kl = _layers()
    t = kl.TableConfig("t", 20, 8, placement=placement, optimizer="sgd", combiner="sum")
    layer = kl.DistributedEmbedding({"a": kl.FeatureConfig("a", t, (4, 2), (4, 8))})
    good = np.array([[1, 2], [3, 4], [5, 6], [7, 8]], np.int32)
    bad = good.copy()
    bad[2, 1] = 20
    out = layer({"a": good})["a"]
    layer.check_ids(wait=True)                                         # nothing to report
    tab = layer.get_embedding_tables()["t"].clone()
    out_bad = layer({"a": bad})["a"]
    # the bad lookup contributed nothing (not row 0, not the last row)
    torch.testing.assert_close(out_bad[2].float(), tab[5].float())
    torch.testing.assert_close(out_bad[0].float(), out[0].float())
    with pytest.raises(IndexError):
        layer.check_ids(wait=True)
    layer.check_ids(wait=True)                                         # the flag was consumed
    layer({"a": bad})
    torch.cuda.synchronize()
    with pytest.raises(IndexError):                                    # ... or it surfaces at the next call
        layer({"a": good})
    layer({"a": good})
    layer.check_ids(wait=True)


@pytest.mark.parametrize("kind", ["adagrad", "adam"])
def test_distributed_embedding_state_dict_carries_optimizer_state(kind):
    # the reference's slot variables and `_iterations` are layer variables (jax/distributed_embedding.py:316-345):
    # a checkpoint taken after k steps resumes exactly -- tables, accumulators / moments, update count
    kl = _layers()

    def make():
        opt = kl.Adagrad(0.1, 0.1) if kind == "adagrad" else kl.Adam(0.05)
        t = kl.TableConfig("t", 40, 8, placement="sparsecore", optimizer=opt, combiner="mean")
        layer = kl.DistributedEmbedding({"a": kl.FeatureConfig("a", t, (8, 3), (8, 8))})
        layer.build(None)
        return layer

    rng = np.random.default_rng(9)
    batches = [rng.integers(0, 40, (8, 3)).astype(np.int32) for _ in range(4)]
    grads = [torch.from_numpy(rng.uniform(-1, 1, (8, 8)).astype(np.float32)).to(DEV) for _ in range(4)]

    def step(layer, i):
        (layer({"a": batches[i]})["a"] * grads[i]).sum().backward()

    a = make()
    for i in range(2):
        step(a, i)
    sd = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in a.state_dict().items()}
    assert any(k.endswith("_slot") for k in sd) and sd["_extra_state"]["iterations"] == {"sparsecore/0": 2}
    b = make()
    b.load_state_dict(sd)
    for i in range(2, 4):
        step(a, i)
        step(b, i)
    torch.testing.assert_close(b.get_embedding_tables()["t"], a.get_embedding_tables()["t"], rtol=0, atol=0)
    for k, v in a.state_dict().items():
        if isinstance(v, torch.Tensor):
            torch.testing.assert_close(b.state_dict()[k], v, rtol=0, atol=0)
    # a resumed run without the state would differ (Adam's bias correction / Adagrad's accumulator restart)
    c = make()
    c.set_embedding_tables({"t": sd["sparsecore_t_embeddings"]})
    for i in range(2, 4):
        step(c, i)
    assert not torch.equal(c.get_embedding_tables()["t"], a.get_embedding_tables()["t"])


def test_table_stacking_by_name_and_checkpoint_file(tmp_path):
    # jax/distributed_embedding.py:413-453: named stacks.  Here: one contiguous [sum V, D] parameter per stack, the
    # tables are row windows of it; results equal the unstacked layer's; a checkpoint written to a file restores
    # tables, optimizer slots and the update count (jax/distributed_embedding_test.py:471-552 save / restore)
    kl = _layers()
    from keras_rs_amd.layers import base

    def make(stacking):
        opt = kl.Adagrad(0.1, 0.1)
        tcs = [kl.TableConfig(n, v, 8, placement="sparsecore", optimizer=opt, combiner="sum",
                              initializer=base.RandomUniform(-1, 1, seed=5 + i))
               for i, (n, v) in enumerate([("a", 11), ("b", 7), ("c", 13)])]
        layer = kl.DistributedEmbedding({n: kl.FeatureConfig(n, tc, (6, 2), (6, 8)) for n, tc in zip("abc", tcs)},
                                        table_stacking=stacking)
        layer.build(None)
        return layer

    rng = np.random.default_rng(0)
    ids = {n: rng.integers(0, v, (6, 2)).astype(np.int32) for n, v in zip("abc", (11, 7, 13))}
    gr = {n: torch.from_numpy(rng.uniform(-1, 1, (6, 8)).astype(np.float32)).to(DEV) for n in "abc"}

    def step(layer):
        out = layer(ids)
        sum((out[n] * gr[n]).sum() for n in "abc").backward()
        return out

    plain, stacked = make("auto"), make([["a", "c"]])
    assert "sparsecore_stack_a_c" in stacked.state_dict() and stacked.state_dict()["sparsecore_stack_a_c"].shape == (24, 8)
    assert "sparsecore_b_embeddings" in stacked.state_dict() and "sparsecore_a_embeddings" not in stacked.state_dict()
    o1, o2 = step(plain), step(stacked)
    for n in "abc":
        assert torch.equal(o1[n], o2[n])
    t1, t2 = plain.get_embedding_tables(), stacked.get_embedding_tables()
    for n in "abc":
        assert torch.equal(t1[n], t2[n])
    # the windows alias the stack
    st = stacked.state_dict()["sparsecore_stack_a_c"]
    assert torch.equal(st[:11], t2["a"]) and torch.equal(st[11:], t2["c"])
    # checkpoint file round trip
    path = str(tmp_path / "emb.pt")
    torch.save(stacked.state_dict(), path)
    resumed = make([["a", "c"]])
    resumed.load_state_dict(torch.load(path))
    step(stacked), step(resumed)
    for k, v in stacked.state_dict().items():
        if isinstance(v, torch.Tensor):
            assert torch.equal(resumed.state_dict()[k], v), k
    assert resumed.get_extra_state() == stacked.get_extra_state() == {"iterations": {"sparsecore/0": 2}}
    with pytest.raises(ValueError):
        make([["a", "nope"]])
    with pytest.raises(ValueError):
        make("always")


def _de_configs(placement, optimizer="sgd", combiner="mean"):
    kl = _layers()
    de = KAT["distributed_embedding"]
    t = kl.TableConfig("table", de["vocabulary_size"], de["embedding_dim"], placement=placement,
                       optimizer=optimizer, combiner=combiner)
    return t


@pytest.mark.parametrize("placement", ["default_device", "sparsecore", "auto"])
@pytest.mark.parametrize("combiner", ["sum", "mean", "sqrtn"])
@pytest.mark.parametrize("kind", ["dense1d", "dense2d", "ragged"])
@pytest.mark.parametrize("use_w", [False, True])
def test_distributed_embedding_correctness(placement, combiner, kind, use_w):
    # distributed_embedding_test.py:411-599: closed forms through the nested API, batch 16
    kl = _layers()
    de = KAT["distributed_embedding"]
    B, D = de["batch"], de["embedding_dim"]
    t = _de_configs(placement, combiner=combiner)
    n = B // 2
    if kind == "dense1d":
        x = np.array(de["dense1d_ids_pattern"] * n, np.int32)
        w = np.array([1.0, 2.0] * n, np.float32)
        fc = kl.FeatureConfig("f", t, (B,), (B, D))
    elif kind == "dense2d":
        x = np.array([[2, 3], [4, 5]] * n, np.int32)
        w = np.array([[1.0, 2.0], [3.0, 4.0]] * n, np.float32)
        fc = kl.FeatureConfig("f", t, (B, 2), (B, D))
    else:
        x = np.empty(B, dtype=object)
        w = np.empty(B, dtype=object)
        for i in range(n):
            x[2 * i], x[2 * i + 1] = np.array([1], np.int32), np.array([2, 3, 4, 5], np.int32)
            w[2 * i], w[2 * i + 1] = np.array([1.0], np.float32), np.array([1.0, 2.0, 3.0, 4.0], np.float32)
        fc = kl.FeatureConfig("f", t, (B, 4), (B, D))
    layer = kl.DistributedEmbedding({"group": {"f": fc}})
    res = layer({"group": {"f": x}}, {"group": {"f": w}} if use_w else None)["group"]["f"]
    assert tuple(res.shape) == (B, D)
    e = layer.get_embedding_tables()["table"].cpu().numpy().astype(np.float64)
    rows = []
    for b in range(B):
        ids = np.atleast_1d(x[b])
        ww = np.atleast_1d(w[b]).astype(np.float64) if use_w else np.ones(len(ids))
        if kind == "dense1d" and combiner != "sum":
            ww = np.ones(len(ids))
        s = (ww[:, None] * e[ids]).sum(0)
        if kind != "dense1d":
            if combiner == "mean":
                s = s / ww.sum()
            elif combiner == "sqrtn":
                s = s / math.sqrt((ww * ww).sum())
        rows.append(s)
    np.testing.assert_allclose(res.detach().cpu().numpy(), np.stack(rows).astype(np.float32), rtol=1e-5, atol=1e-6)


def test_distributed_embedding_shared_table_and_preprocessed_call():
    kl = _layers()
    t = _de_configs("default_device", combiner="sum")
    fcs = [kl.FeatureConfig(f"f{i}", t, (16, 1), (16, 7)) for i in range(3)]
    layer = kl.DistributedEmbedding(fcs)
    xs = [np.full((16, 1), i + 1, np.int32) for i in range(3)]
    pre = layer.preprocess(xs)
    outs = layer(pre)
    assert len(layer.weights) == 1
    e = layer.get_embedding_tables()["table"]
    for i, o in enumerate(outs):
        assert torch.equal(o, e[i + 1].expand(16, 7))


@pytest.mark.parametrize("placement,optimizer", [("default_device", "sgd"), ("sparsecore", "sgd"),
                                                 ("sparsecore", "adagrad")])
def test_distributed_embedding_training_step_matches_formula(placement, optimizer):
    # one step with loss = sum(out * g): table rows move by the optimizer formula of
    # jax/test_utils.py:474-497 (fused) / by a torch SGD step on the dense gradient
    kl = _layers()
    lr = 0.1
    opt = kl.SGD(lr) if optimizer == "sgd" else kl.Adagrad(lr, 0.1)
    t = kl.TableConfig("table", 23, 7, placement=placement, optimizer=opt, combiner="sum")
    layer = kl.DistributedEmbedding({"a": kl.FeatureConfig("a", t, (16, 2), (16, 7))})
    x = np.random.default_rng(0).integers(0, 23, (16, 2)).astype(np.int32)
    before = layer({"a": x})["a"].detach().clone()
    table0 = layer.get_embedding_tables()["table"].clone()
    g = torch.rand(16, 7, device=DEV)
    out = layer({"a": x})["a"]
    (out * g).sum().backward()
    if placement == "default_device":
        with torch.no_grad():
            for p in layer.weights:
                p -= lr * p.grad
    after = layer({"a": x})["a"].detach()
    assert not torch.allclose(before, after)  # distributed_embedding_test.py:240-384
    dense = torch.zeros(23, 7, dtype=torch.float64, device=DEV)
    dense.index_add_(0, torch.from_numpy(x.reshape(-1)).long().to(DEV), g.double().repeat_interleave(2, 0))
    if optimizer == "sgd":
        exp = table0.double() - lr * dense
    else:
        acc = 0.1 + dense * dense
        exp = torch.where(dense != 0, table0.double() - lr * dense / acc.sqrt(), table0.double())
    np.testing.assert_allclose(layer.get_embedding_tables()["table"].cpu().numpy(), exp.cpu().numpy(),
                               rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("policy", ["float32", "mixed_bfloat16"])
@pytest.mark.parametrize("b,d_in,units,act", [(64, 13, 512, "relu"), (300, 520, 264, "sigmoid"), (33, 256, 1, None),
                                              (257, 512, 256, "tanh"), (16, 24, 8, lambda t: t * t)])
def test_dense_layer_matches_torch(policy, b, d_in, units, act):
    # the Dense blocks of the DLRM MLPs (examples/ml_perf/model.py:214-262) on
    # krs_gemm's bias + activation epilogue; forward and all three gradients against plain torch
    kl = _layers()
    from keras_rs_amd.layers import base as kl_base

    layer = kl.Dense(units, activation=act, kernel_initializer=kl_base.GlorotUniform(seed=3),
                     bias_initializer=kl_base.RandomUniform(-0.1, 0.1, seed=4), dtype=policy)
    x = (torch.rand(b, d_in, device=DEV) - 0.5).requires_grad_(True)
    y = layer(x)
    assert tuple(y.shape) == (b, units) and [tuple(w.shape) for w in layer.weights] == [(d_in, units), (units,)]
    gy = torch.rand(b, units, device=DEV)
    (y.float() * gy).sum().backward()
    fn = {"relu": torch.relu, "sigmoid": torch.sigmoid, "tanh": torch.tanh, None: lambda t: t}.get(act, act)
    bf = policy != "float32"
    xr = x.detach().clone().requires_grad_(True)
    kr = layer.kernel.detach().clone().requires_grad_(True)
    br = layer.bias.detach().clone().requires_grad_(True)
    xin, kin = (xr.bfloat16().float(), kr.bfloat16().float()) if bf else (xr, kr)
    yr = fn(xin @ kin + br)
    (yr * gy).sum().backward()
    tol = dict(rtol=2 ** -6, atol=3e-2) if bf else dict(rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(y.detach().float().cpu().numpy(), yr.detach().cpu().numpy(), **tol)
    gtol = dict(rtol=2 ** -5, atol=0.15) if bf else dict(rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(x.grad.cpu().numpy(), xr.grad.cpu().numpy(), **gtol)
    np.testing.assert_allclose(layer.kernel.grad.cpu().numpy(), kr.grad.cpu().numpy(), **gtol)
    np.testing.assert_allclose(layer.bias.grad.cpu().numpy(), br.grad.cpu().numpy(), **gtol)
    assert kl.Dense.from_config(layer.get_config()).units == units if act is None or isinstance(act, str) else True


def test_dlrm_dcn_v2_example_trains():
    # the reference's ml_perf model assembled from the MI355X layers (examples/dlrm_dcn_v2.py): a few
    # steps on a fixed batch lower the BCE loss, tables and dense weights move, nothing is NaN
    import importlib.util

    spec = importlib.util.spec_from_file_location(
        "dlrm_dcn_v2", os.path.join(os.path.dirname(os.path.dirname(__file__)), "examples", "dlrm_dcn_v2.py"))
    ex = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ex)
    hots = [3, 1, 2, 5]
    model = ex.build_model(256, 500, hots, embedding_dim=32, projection=16, cross_layers=2, bottom=(64, 32),
                           top=(64, 32, 1), table_optimizer=_layers().Adagrad(0.05, 0.1))
    x, y = ex.synthetic_batch(256, 13, 500, hots, torch.device(DEV))
    box = [None]
    before = None
    losses = []
    for _ in range(6):
        losses.append(float(ex.train_step(model, box, x, y)))
        if before is None:
            before = {k: v.clone() for k, v in model.embedding_layer.get_embedding_tables().items()}
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    after = model.embedding_layer.get_embedding_tables()
    assert any(not torch.equal(before[k], after[k]) for k in before)


def test_threaded_data_loader_feeds_preprocessed_host_batches():
    # host ids -> loader threads (preprocess + asynchronous upload on their own
    # streams) -> layer call; results equal the direct call on the same ids
    from keras_rs_amd.data import ThreadedDataLoader

    kl = _layers()
    t = kl.TableConfig("t", 50, 8, placement="sparsecore", optimizer="sgd", combiner="mean")
    layer = kl.DistributedEmbedding({"a": kl.FeatureConfig("a", t, (32, 4), (32, 8)),
                                     "b": kl.FeatureConfig("b", t, (32, 1), (32, 8))})
    layer.build(None)
    rng = np.random.default_rng(1)
    batches = [({"large_emb_inputs": {"a": rng.integers(0, 50, (32, 4)).astype(np.int32),
                                      "b": rng.integers(0, 50, (32, 1)).astype(np.int32)},
                 "dense_input": rng.random((32, 3)).astype(np.float32)}, rng.integers(0, 2, 32)) for _ in range(6)]
    loader = ThreadedDataLoader(layer.preprocess, batches, num_workers=2)
    seen = 0
    for x, y in loader:
        assert x["dense_input"].is_cuda and y.is_cuda
        out = layer(x["large_emb_inputs"])
        # find the host batch this one came from (workers may reorder)
        k = next(i for i, (bx, _) in enumerate(batches) if np.array_equal(bx["dense_input"], x["dense_input"].cpu().numpy()))
        ref = layer(batches[k][0]["large_emb_inputs"])
        assert torch.equal(out["a"], ref["a"]) and torch.equal(out["b"], ref["b"])
        seen += 1
    assert seen == 6
    loader.stop()


@pytest.mark.parametrize("optimizer", ["adam", "ftrl"])
def test_distributed_embedding_adam_and_ftrl_steps_match_the_keras_formulas(optimizer):
    # the default TableConfig.optimizer is "adam" (distributed_embedding_config.py:57); two training
    # steps on 'sparsecore' tables against a float64 transcription of the Keras update rules
    kl = _layers()
    opt = kl.Adam(0.05, 0.9, 0.999, 1e-7) if optimizer == "adam" else kl.Ftrl(0.05, -0.5, 0.1, 0.01, 0.02, 0.1)
    t = kl.TableConfig("table", 23, 8, placement="sparsecore", optimizer=opt, combiner="sum")
    layer = kl.DistributedEmbedding({"a": kl.FeatureConfig("a", t, (16, 2), (16, 8))})
    x = np.random.default_rng(0).integers(0, 23, (16, 2)).astype(np.int32)
    layer.build(None)
    W = layer.get_embedding_tables()["table"].double().cpu().numpy()
    w_start = W.copy()
    m = np.zeros_like(W)
    v = np.zeros_like(W)
    n = np.full_like(W, 0.1)
    z = np.zeros_like(W)
    touched = np.zeros(23, bool)
    touched[x.reshape(-1)] = True
    for step in (1, 2):
        g = torch.rand(16, 8, device=DEV, generator=torch.Generator(device=DEV).manual_seed(step))
        (layer({"a": x})["a"] * g).sum().backward()
        dense = np.zeros_like(W)
        np.add.at(dense, x.reshape(-1), np.repeat(g.double().cpu().numpy(), 2, axis=0))
        if optimizer == "adam":
            corr = np.sqrt(1 - 0.999 ** step) / (1 - 0.9 ** step)
            m[touched] += (dense[touched] - m[touched]) * 0.1
            v[touched] += (dense[touched] ** 2 - v[touched]) * 0.001
            W[touched] -= 0.05 * corr * m[touched] / (np.sqrt(v[touched]) + 1e-7)
        else:
            n_new = n + dense * dense
            z_new = z + dense - (np.sqrt(n_new) - np.sqrt(n)) / 0.05 * W
            quad = np.sqrt(n_new) / 0.05 + 2 * (0.02 + 0.1 / (2 * 0.05))
            w_new = (np.clip(z_new, -0.01, 0.01) - z_new) / quad
            W[touched], n[touched], z[touched] = w_new[touched], n_new[touched], z_new[touched]
        np.testing.assert_allclose(layer.get_embedding_tables()["table"].cpu().numpy(), W, rtol=2e-5, atol=2e-6)
    assert np.array_equal(W[~touched], w_start[~touched])  # lazy: rows never looked up do not move


def test_fused_optimizer_follows_a_learning_rate_schedule():
    # jax/config_conversion.py:136-176: callable learning rates are evaluated per step; here SGD with
    # lr(step) = 0.5 / (1 + step) over three updates of one row
    kl = _layers()

    class ScheduledSGD:
        def __init__(self):
            self.learning_rate = lambda step: 0.5 / (1 + step)
    ScheduledSGD.__name__ = "SGD"
    t = kl.TableConfig("table", 5, 8, placement="sparsecore", optimizer=ScheduledSGD(), combiner="sum")
    layer = kl.DistributedEmbedding({"a": kl.FeatureConfig("a", t, (1, 1), (1, 8))})
    x = np.array([[2]], np.int32)
    layer.build(None)
    w = layer.get_embedding_tables()["table"][2].double().cpu().numpy()
    for step in range(3):
        layer({"a": x})["a"].sum().backward()          # gradient of the row: all ones
        w = w - 0.5 / (1 + step)
        np.testing.assert_allclose(layer.get_embedding_tables()["table"][2].cpu().numpy(), w, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("lead", [0, 8, 4])
def test_concat_features_uses_the_slab_and_trains_like_torch_cat(lead):
    # SURVEY.md section 8f.3 (concat-free layout): concat_features([dense, *embeddings]) == torch.cat,
    # forward values, gradient of `dense`, and the fused table update, for lead == width of the dense
    # head (slab returned in place), lead == 0 (two-piece concat) and a mismatching lead (same)
    kl = _layers()
    rng = np.random.default_rng(5)
    ids = {k: rng.integers(0, 40, (12, h)).astype(np.int32) for k, h in (("a", 3), ("b", 1), ("c", 2))}
    w_dot = torch.rand(12, 6, device=DEV)

    def run(use_helper):
        tabs = [kl.TableConfig(f"t{k}", 40, 8, placement="sparsecore", optimizer=kl.SGD(0.5), combiner="sum",
                               initializer=kl_base.RandomUniform(-1, 1, seed=3 + i)) for i, k in enumerate("abc")]
        layer = kl.DistributedEmbedding({k: kl.FeatureConfig(k, t, ids[k].shape, (12, 8)) for k, t in zip("abc", tabs)},
                                        slab_lead_cols=lead if use_helper else 0)
        dense = torch.linspace(-1, 1, 12 * 8, device=DEV).reshape(12, 8).requires_grad_(True)
        emb = layer(ids)
        feats = [dense] + [emb[k] for k in "abc"]
        inter = kl.DotInteraction()(feats)
        x0 = kl.concat_features(feats) if use_helper else torch.cat(feats, dim=-1)
        if use_helper and lead == 8:
            assert x0.data_ptr() == emb["a"]._krs_slab[0].data_ptr()  # the slab itself, nothing copied
        y = kl.FeatureCross(kernel_initializer=kl_base.GlorotUniform(seed=1))(x0, x0)
        ((y * y).sum() + (inter * w_dot).sum()).backward()
        return x0.detach().clone(), dense.grad.clone(), layer.get_embedding_tables()

    from keras_rs_amd.layers import base as kl_base

    x_a, gd_a, t_a = run(True)
    x_b, gd_b, t_b = run(False)
    assert torch.equal(x_a, x_b)
    np.testing.assert_allclose(gd_a.cpu().numpy(), gd_b.cpu().numpy(), rtol=1e-5, atol=1e-6)
    for k in t_a:
        np.testing.assert_allclose(t_a[k].cpu().numpy(), t_b[k].cpu().numpy(), rtol=1e-5, atol=1e-6)
    # the reserved columns are handed out once per forward pass: a second concat with another head is a copy
    if lead == 8:
        t = kl.TableConfig("t", 40, 8, placement="sparsecore", optimizer="sgd", combiner="sum")
        layer = kl.DistributedEmbedding({"a": kl.FeatureConfig("a", t, (12, 1), (12, 8))}, slab_lead_cols=8)
        emb = layer({"a": ids["b"]})
        d1, d2 = torch.ones(12, 8, device=DEV), torch.full((12, 8), 2.0, device=DEV)
        c1 = kl.concat_features([d1, emb["a"]])
        c2 = kl.concat_features([d2, emb["a"]])
        assert c1.data_ptr() != c2.data_ptr() and torch.all(c1[:, :8] == 1) and torch.all(c2[:, :8] == 2)
        assert torch.equal(c1[:, 8:], c2[:, 8:])
    # anything that is not "all features of one slab, in order, at the end" is a plain concat
    plain = [torch.ones(2, 3, device=DEV), torch.zeros(2, 2, device=DEV)]
    assert torch.equal(kl.concat_features(plain), torch.cat(plain, dim=-1))


def test_set_embedding_tables_and_mixed_width_groups():
    kl = _layers()
    t1 = kl.TableConfig("t1", 30, 8, placement="default_device", combiner="sum")
    t2 = kl.TableConfig("t2", 40, 16, placement="sparsecore", optimizer="sgd", combiner="mean")
    layer = kl.DistributedEmbedding({"a": kl.FeatureConfig("a", t1, (4, 2), (4, 8)),
                                     "b": kl.FeatureConfig("b", t2, (4, 3), (4, 16))})
    layer.build(None)
    layer.set_embedding_tables({"t1": np.ones((30, 8), np.float32), "t2": np.full((40, 16), 2.0, np.float32)})
    out = layer({"a": np.zeros((4, 2), np.int32), "b": np.zeros((4, 3), np.int32)})
    assert torch.all(out["a"] == 2.0) and torch.all(out["b"] == 2.0)


def test_embedding_layer_and_readme_quickstart_shape():
    # README.md:46-76 / feature_cross.py:68-86: Embedding(32, 6) -> FeatureCross x2
    kl = _layers()
    emb = kl.Embedding(32, 6)
    ids = torch.randint(0, 32, (2,), device=DEV)
    x0 = emb(ids)
    x1 = kl.FeatureCross()(x0, x0)
    x2 = kl.FeatureCross()(x0, x1)
    assert tuple(x2.shape) == (2, 6)
    assert torch.equal(x0, emb.embeddings[ids.long()])
    seq = emb(torch.randint(0, 32, (3, 4), device=DEV))
    assert tuple(seq.shape) == (3, 4, 6)


@pytest.mark.parametrize("optimizer", ["sgd", "adagrad", "adam", "ftrl"])
def test_sharded_layer_world1_on_hip_matches_unsharded_layer(optimizer):
    """The sharded path (bucketise -> owner-side partial pooling -> home-side sum, and the CSR-form fused
    apply) on the HIP kernels, degenerate world of 1: same outputs and same updated tables as
    DistributedEmbedding, also through layers.concat_features with a reserved dense slot."""
    kl = _layers()
    from keras_rs_amd.sharded import ShardedDistributedEmbedding

    opt = {"sgd": kl.SGD(0.1), "adagrad": kl.Adagrad(0.1, 0.1), "adam": kl.Adam(0.1),
           "ftrl": kl.Ftrl(0.1, -0.5, 0.1, 0.01, 0.02, 0.3)}[optimizer]
    rng = np.random.default_rng(5)
    V, D, B = [50, 31], 16, 24
    hots = [1, 3, 2]
    tix = [0, 1, 0]

    def make(cls, **kw):
        tcs = [kl.TableConfig(f"t{i}", V[i], D, optimizer=opt, combiner=["mean", "sum"][i], placement="sparsecore")
               for i in range(2)]
        fcs = {f"f{i}": kl.FeatureConfig(f"f{i}", tcs[tix[i]], (B, hots[i]), (B, D)) for i in range(3)}
        return cls(fcs, **kw)

    full = {f"t{i}": rng.uniform(-1, 1, (V[i], D)).astype(np.float32) for i in range(2)}
    ids = {f"f{i}": rng.integers(0, V[tix[i]], (B, hots[i])).astype(np.int32) for i in range(3)}
    w = {f"f{i}": rng.uniform(0.1, 1, (B, hots[i])).astype(np.float32) for i in range(3)}
    g = {f"f{i}": torch.rand(B, D, device=DEV) for i in range(3)}
    results = []
    dense = torch.rand(B, 4, device=DEV)
    gx = torch.rand(B, 4 + 3 * D, device=DEV)
    for layer in (make(kl.DistributedEmbedding), make(ShardedDistributedEmbedding, slab_lead_cols=4)):
        layer.build(None)
        layer.set_embedding_tables(full)
        out = layer(ids, w)
        x0 = kl.concat_features([dense] + [out[k] for k in out])
        assert torch.equal(x0, torch.cat([dense] + [out[k] for k in out], dim=-1))
        (sum((out[k] * g[k]).sum() for k in out) + (x0 * gx).sum()).backward()
        results.append(({k: v.detach().cpu().numpy() for k, v in out.items()},
                        {k: v.cpu().numpy() for k, v in layer.get_embedding_tables().items()}))
    for k in results[0][0]:
        np.testing.assert_allclose(results[1][0][k], results[0][0][k], rtol=1e-6, atol=1e-6)
    for k in results[0][1]:
        np.testing.assert_allclose(results[1][1][k], results[0][1][k], rtol=1e-5, atol=1e-6)
        assert not np.allclose(results[0][1][k], full[k])


@pytest.mark.parametrize("policy", ["float32", "mixed_bfloat16"])
def test_dot_interaction_gradient_joins_the_slab_gradient_in_the_kernel(policy):
    # autograd.SlabGradRelay: with [dense, *embeddings] going to DotInteraction first and to concat_features
    # second (the DLRM order), the interaction's gradient of the embeddings is added into the concat's gradient
    # inside krs_dot_interaction_bwd_accumulate.  Same numbers as the separate add (which runs when the concat
    # result retains its gradient, or when the interaction is called after the concat).
    from keras_rs_amd.autograd import SlabGradRelay
    from keras_rs_amd.layers import base as kl_base

    kl = _layers()
    rng = np.random.default_rng(11)
    B, D_ = 24, 16
    ids = {k: rng.integers(0, 50, (B, h)).astype(np.int32) for k, h in (("a", 2), ("b", 1), ("c", 4))}
    w_dot = torch.rand(B, 6, device=DEV)
    dt = torch.bfloat16 if policy == "mixed_bfloat16" else torch.float32

    def run(mode):
        tabs = [kl.TableConfig(f"t{k}", 50, D_, placement="sparsecore", optimizer=kl.SGD(0.25), combiner="sum",
                               initializer=kl_base.RandomUniform(-1, 1, seed=7 + i)) for i, k in enumerate("abc")]
        layer = kl.DistributedEmbedding({k: kl.FeatureConfig(k, t, ids[k].shape, (B, D_)) for k, t in zip("abc", tabs)},
                                        slab_lead_cols=D_, dtype=policy)
        dense = torch.linspace(-1, 1, B * D_, device=DEV).reshape(B, D_).to(dt).requires_grad_(True)
        emb = layer(ids)
        feats = [dense] + [emb[k] for k in "abc"]
        if mode == "dot_last":
            x0 = kl.concat_features(feats)
            inter = kl.DotInteraction(dtype=policy)(feats)
        else:
            inter = kl.DotInteraction(dtype=policy)(feats)
            x0 = kl.concat_features(feats)
        extra = None
        if mode == "second_concat":      # the slab gets a second consumer: its gradients are summed by autograd,
            extra = kl.concat_features(feats)       # so nothing may be added into one of them in place
        if mode == "retain":
            x0.retain_grad()
        if mode == "watch_view":
            emb["b"].retain_grad()
        y = kl.FeatureCross(kernel_initializer=kl_base.GlorotUniform(seed=1), dtype=policy)(x0, x0)
        if mode == "dropped":     # a model's forward returns and the concat result is referenced by the graph only
            del x0, feats, emb
        before = SlabGradRelay.joined
        loss = (y.float() ** 2).sum() + (inter.float() * w_dot).sum()
        if extra is not None:
            loss = loss + 0.5 * (extra.float() ** 2).sum()
        loss.backward()
        took = SlabGradRelay.joined - before
        gx0 = x0.grad.clone() if mode == "retain" else None
        return took, dense.grad.float().clone(), {k: v.float().clone() for k, v in layer.get_embedding_tables().items()}, gx0

    took, gd, tabs, _ = run("joined")
    assert took == 1
    took_d, gd_d, tabs_d, _ = run("dropped")
    assert took_d == 1 and torch.equal(gd, gd_d) and all(torch.equal(tabs[k], tabs_d[k]) for k in tabs)
    tol = dict(rtol=2 ** -6, atol=2e-2) if dt == torch.bfloat16 else dict(rtol=1e-5, atol=1e-5)
    for mode in ("retain", "dot_last", "watch_view"):
        took_m, gd_m, tabs_m, gx0 = run(mode)
        assert took_m == 0
        torch.testing.assert_close(gd, gd_m, **tol)
        for k in tabs:
            torch.testing.assert_close(tabs[k], tabs_m[k], **tol)
        if gx0 is not None:   # the retained gradient of the concat is the cross layer's alone
            assert gx0.shape == (B, 4 * D_) and torch.isfinite(gx0.float()).all()
    # a second concat of the same features: the join must step aside, and the gradients must hold the extra term
    took_s, gd_s, tabs_s, _ = run("second_concat")
    assert took_s == 0 and not torch.equal(gd_s, gd)
    # ... the same numbers as a plain-torch composition of that loss on the joined run's inputs (dense gradient only:
    # d(0.5 |extra|^2)/d(dense) = dense)
    dense0 = torch.linspace(-1, 1, B * D_, device=DEV).reshape(B, D_).to(dt).float()
    torch.testing.assert_close(gd_s, gd + dense0, **tol)


def test_prepared_casts_behind_the_optimizer_step():
    """optim.Adagrad(prepare_casts=True) prepares the compute-dtype copies of every kernel a layer has asked for, for
    the next forward, in one launch (krs_cast_transpose_many): the copies equal the casts of the UPDATED weights, are
    handed out once, and are dropped when the weight changed through torch in between (version bump); training with and
    without them agrees (to the last-bit noise of the atomically summed bias gradients)."""
    from keras_rs_amd import dense_ops as D
    from keras_rs_amd.layers import base as kl_base
    from keras_rs_amd.optim import Adagrad

    kl = _layers()

    def make():
        return [kl.FeatureCross(projection_dim=16, kernel_initializer=kl_base.GlorotUniform(seed=1), dtype="mixed_bfloat16"),
                kl.FeatureCross(projection_dim=16, kernel_initializer=kl_base.GlorotUniform(seed=2), dtype="mixed_bfloat16"),
                kl.Dense(8, activation="relu", kernel_initializer=kl_base.GlorotUniform(seed=3), dtype="mixed_bfloat16")]

    g = torch.Generator(device=DEV).manual_seed(5)
    xs = [torch.randn(96, 64, device=DEV, generator=g).to(torch.bfloat16) for _ in range(4)]

    def run(prepare):
        layers = make()
        opt = None
        outs = []
        for x in xs:
            y = layers[2](layers[1](x, layers[0](x, x)))
            outs.append(y.detach().float().clone())
            y.float().pow(2).sum().backward()
            if opt is None:
                opt = Adagrad([p for l in layers for p in l.parameters()], lr=0.05, initial_accumulator_value=0.1,
                              prepare_casts=prepare)
            opt.step()
            opt.zero_grad(set_to_none=True)
        return layers, outs

    layers, o1 = run(True)
    _, o0 = run(False)
    for a, b in zip(o0, o1):
        torch.testing.assert_close(a, b, rtol=2e-2, atol=1e-3)
    assert not torch.equal(o1[0], o1[3])
    mats = [p for l in layers for p in l.parameters() if p.dim() == 2]
    assert len(mats) == 5
    for p in mats:
        key, plain, trans = p._krs_cast                                  # prepared by the last step()
        assert torch.equal(plain, p.detach().to(torch.bfloat16)) and torch.equal(trans, plain.t().contiguous())
        a, b = D.cast_transpose(p, torch.bfloat16)
        assert a is plain and b is trans and p._krs_cast is None          # handed out once
        a2, b2 = D.cast_transpose(p, torch.bfloat16)                     # ... then the ordinary path
        assert a2 is not plain and torch.equal(a2, plain) and torch.equal(b2, trans)
    assert D.refresh_casts(mats) == 5
    with torch.no_grad():
        mats[0].mul_(0.5)                                                # through torch: version bump
    a, _ = D.cast_transpose(mats[0], torch.bfloat16)
    assert torch.equal(a, mats[0].detach().to(torch.bfloat16))


def test_weight_gradients_on_the_second_stream_are_the_same_gradients():
    """autograd.WGRAD_SIDE_STREAM: dK / dU of the cross layers are computed on a second stream that the main stream
    rejoins at the end of the backward pass -- same kernels, same inputs: every gradient bit-identical to the
    one-stream schedule, also when the optimizer step follows immediately (the rejoin must order it)."""
    from keras_rs_amd import autograd as A
    from keras_rs_amd.layers import base as kl_base
    from keras_rs_amd.optim import Adagrad

    kl = _layers()
    g = torch.Generator(device=DEV).manual_seed(3)
    x0 = torch.randn(4096, 512, device=DEV, generator=g).to(torch.bfloat16)
    A_rows = 1024          # (the side stream is taken from this many rows on: lowered for the test)

    def run(side):
        old, old_rows = A.WGRAD_SIDE_STREAM, A.WGRAD_SIDE_MIN_ROWS
        A.WGRAD_SIDE_STREAM, A.WGRAD_SIDE_MIN_ROWS = side, A_rows
        try:
            layers = [kl.FeatureCross(projection_dim=64, kernel_initializer=kl_base.GlorotUniform(seed=i), dtype="mixed_bfloat16",
                                      use_bias=False) for i in range(3)]
            x = x0.clone().requires_grad_()
            xl = x
            for layer in layers:
                xl = layer(x, xl)
            xl.float().pow(2).mean().backward()
            grads = [p.grad.clone() for layer in layers for p in layer.parameters()] + [x.grad.clone()]
            opt = Adagrad([p for layer in layers for p in layer.parameters()], lr=0.01, initial_accumulator_value=0.1)
            opt.step()                                       # right behind the backward: reads the side stream's results
            torch.cuda.synchronize()
            return grads + [p.detach().clone() for layer in layers for p in layer.parameters()]
        finally:
            A.WGRAD_SIDE_STREAM, A.WGRAD_SIDE_MIN_ROWS = old, old_rows

    a, b = run(True), run(False)
    assert len(a) == len(b) == 13
    for u, v in zip(a, b):
        assert torch.equal(u, v)
    # gradient ACCUMULATION (a second backward into existing .grad: an add kernel on the main stream) and a user hook on
    # a weight both read the gradient before the end of the pass: the layer must keep those on the main stream
    layer = kl.FeatureCross(projection_dim=64, kernel_initializer=kl_base.GlorotUniform(seed=9), dtype="mixed_bfloat16")
    xa = x0.clone().requires_grad_()
    seen = []
    A.WGRAD_SIDE_MIN_ROWS = A_rows
    for rep in range(2):
        layer(xa, xa).float().pow(2).mean().backward()
        if rep == 0:
            g1 = [p.grad.clone() for p in layer.parameters()]
            layer.kernel.register_hook(lambda g: seen.append(g.clone()))
    torch.cuda.synchronize()
    for p, g in zip(layer.parameters(), g1):
        torch.testing.assert_close(p.grad, 2 * g, rtol=1e-5, atol=1e-8)
    A.WGRAD_SIDE_MIN_ROWS = 32768
    assert len(seen) == 1 and torch.equal(seen[0], g1[[n for n, _ in layer.named_parameters()].index("kernel")])


def test_second_stream_is_opt_in_and_keeps_away_from_weights_with_a_second_reader():
    """the weight-gradient stream raced when the main stream read dK / dU before the end-of-backward
    rejoin.  (a) it is opt-in now (set_wgrad_side_stream / KRS_WGRAD_SIDE=1); (b) bf16 VARIABLES (dtype="bfloat16"): the
    casts of the fp32 products to the weights' dtype run on that stream too -- gradients equal the one-stream ones bit
    for bit; (c) a weight with a regulariser (its penalty is a second gradient contribution, summed by autograd on the
    main stream) and (d) a layer applied twice in one graph keep their gradients on the main stream."""
    import os

    from keras_rs_amd import autograd as A
    from keras_rs_amd.layers import base as kl_base

    kl = _layers()
    # (the module default is off: `KRS_WGRAD_SIDE` unset; an earlier test of this session may have run the example's
    #  train_step, which opts in for its process -- every case below sets the switch itself)
    src = open(A.__file__).read()
    assert 'environ.get("KRS_WGRAD_SIDE", "0")' in src and "KRS_WGRAD_SIDE" not in os.environ
    g = torch.Generator(device=DEV).manual_seed(4)
    x0 = torch.randn(4096, 512, device=DEV, generator=g).to(torch.bfloat16)
    old_rows = A.WGRAD_SIDE_MIN_ROWS
    A.WGRAD_SIDE_MIN_ROWS = 1024
    used = []
    real_stream = A._wgrad_stream

    def spy(device):
        used.append(1)
        return real_stream(device)

    A._wgrad_stream = spy
    try:
        def run(side, dtype, reg=None, twice=False):
            old = A.set_wgrad_side_stream(side)
            try:
                layers = [kl.FeatureCross(projection_dim=64, kernel_initializer=kl_base.GlorotUniform(seed=i), dtype=dtype,
                                          kernel_regularizer=reg) for i in range(2)]
                x = x0.clone().requires_grad_()
                xl = x
                for layer in layers:
                    xl = layer(x, xl)
                    if twice:
                        xl = layer(x, xl)
                loss = xl.float().pow(2).mean()
                if reg is not None:
                    loss = loss + sum(sum(layer.losses) for layer in layers)
                loss.backward()
                torch.cuda.synchronize()
                return [p.grad.clone() for layer in layers for p in layer.parameters()] + [x.grad.clone()]
            finally:
                A.set_wgrad_side_stream(old)

        del used[:]
        a = run(True, "bfloat16")
        assert used, "bf16 variables: the second stream was expected to be taken"
        b = run(False, "bfloat16")
        assert all(t.dtype == torch.bfloat16 for t in a[:-1][::3])       # (down_kernel, kernel are bf16 variables)
        for u, v in zip(a, b):
            assert torch.equal(u, v)
        for kw in (dict(reg="l2"), dict(twice=True)):
            del used[:]
            a = run(True, "mixed_bfloat16", **kw)
            assert not used, f"{kw}: a weight with a second reader must stay on the main stream"
            b = run(False, "mixed_bfloat16", **kw)
            for u, v in zip(a, b):
                assert torch.equal(u, v)
        # the penalty really is in the gradient: L2(0.01) adds 0.02 * w
        layer = kl.FeatureCross(projection_dim=64, dtype="mixed_bfloat16", kernel_regularizer=kl_base.L2(0.01), use_bias=False)
        x = x0.clone()
        layer(x, x)
        sum(layer.losses).backward()
        torch.testing.assert_close(layer.kernel.grad, 0.02 * layer.kernel.detach(), rtol=1e-6, atol=1e-9)
    finally:
        A._wgrad_stream = real_stream
        A.WGRAD_SIDE_MIN_ROWS = old_rows


def test_distributed_embedding_mixed_placement_and_update_stats_argument():
    """distributed_embedding_test.py:654-723 (placements intermixed: every feature must come back from ITS table -- the
    widths differ, and here the values are checked too) and :760-771 (`update_stats=True` is accepted at construction)."""
    kl = _layers()
    B = 16
    t1 = kl.TableConfig("table1", 50, 16, placement="default_device")
    t2 = kl.TableConfig("table2", 50, 32, placement="sparsecore")
    t3 = kl.TableConfig("table3", 50, 64, placement="default_device")
    cfg = {"feature1": kl.FeatureConfig("feature1", t1, (B, 1), (B, 16)),
           "feature2": kl.FeatureConfig("feature2", t2, (B, 1), (B, 32)),
           "feature3": kl.FeatureConfig("feature3", t3, (B, 1), (B, 64))}
    layer = kl.DistributedEmbedding(cfg, update_stats=True)
    assert layer.update_stats is True
    rng = np.random.default_rng(2)
    ids = {k: rng.integers(0, 50, (B, 1)).astype(np.int32) for k in cfg}
    res = layer(ids)
    tables = {k: v.cpu().numpy() for k, v in layer.get_embedding_tables().items()}
    for k, t, d in (("feature1", "table1", 16), ("feature2", "table2", 32), ("feature3", "table3", 64)):
        assert tuple(res[k].shape) == (B, d)
        np.testing.assert_array_equal(res[k].detach().cpu().numpy(), tables[t][ids[k][:, 0]])   # one id per bag: its row


def test_cross_stack_backward_with_the_elementwise_pass_in_the_product_epilogue():
    """Round 4 (krs_gemm_cross_bwd inside autograd.CrossLayerFn): in a stack `xl = layer(x0, xl)` the data-gradient
    product of layer l+1 runs the elementwise backward of layer l in its epilogue.  Every gradient must be what the
    separate pass gives -- bit for bit for the data path and the kernels' gradients, to summation order for the bias
    gradients -- for a stack whose bottom layer is fed x0 itself (the direct term folds into dL/dx0) and with an
    activation; and when an intermediate output has a SECOND consumer (autograd sums two gradients for it) the layers
    notice that the product's G was not all of dL/dy and correct through the elementwise kernel."""
    from keras_rs_amd import autograd as A
    from keras_rs_amd.layers import base as kl_base

    kl = _layers()
    g = torch.Generator(device=DEV).manual_seed(21)
    B, d, p = 16384, 768, 256                       # 64 x 3 tiles of 256 x 256: the fused ring kernel applies
    x0 = (torch.randn(B, d, device=DEV, generator=g) * 0.5).to(torch.bfloat16)

    def run(fuse, pre_activation=None, tap=False):
        old = A.FUSE_CROSS_BWD
        A.FUSE_CROSS_BWD = fuse
        try:
            layers = [kl.FeatureCross(projection_dim=p, kernel_initializer=kl_base.GlorotUniform(seed=40 + i),
                                      bias_initializer=kl_base.RandomUniform(-0.1, 0.1, seed=50 + i),
                                      pre_activation=pre_activation, dtype="mixed_bfloat16") for i in range(3)]
            x = x0.clone().requires_grad_()
            xl, outs = x, []
            for layer in layers:
                xl = layer(x, xl)
                outs.append(xl)
            loss = xl.float().pow(2).mean()
            if tap:
                loss = loss + outs[0].float().mean() * 3.0       # a second consumer of the first layer's output
            loss.backward()
            torch.cuda.synchronize()
            named = [(f"{i}.{n}", q.grad.clone()) for i, layer in enumerate(layers) for n, q in layer.named_parameters()]
            return named + [("x", x.grad.clone())]
        finally:
            A.FUSE_CROSS_BWD = old

    calls = []
    from keras_rs_amd import dense_ops as D

    real = D.gemm_cross_bwd

    def spy(*a, **k):
        calls.append(k.get("fold_direct"))
        return real(*a, **k)

    D.gemm_cross_bwd = spy
    try:
        for kw in (dict(), dict(pre_activation="relu")):
            del calls[:]
            a = run(True, **kw)
            assert calls == [False, True], calls       # layer 3 -> layer 2, layer 2 -> layer 1 (the bottom: x is x0)
            b = run(False, **kw)
            for (name, u), (_, v) in zip(a, b):
                if name.endswith("bias"):
                    torch.testing.assert_close(u, v, rtol=1e-5, atol=1e-6)
                elif name == "x":
                    # dL/dx0: the top layer's term joins inside the fused epilogue, one bf16 rounding fewer than the
                    # separate passes (KRS_FUSE_TOP_DX0): equal to one ulp, and bit-equal with that switch off (below)
                    torch.testing.assert_close(u.float(), v.float(), rtol=2.0 ** -7, atol=1e-7)
                else:
                    assert torch.equal(u, v), name
            old_top, A.FUSE_TOP_DX0 = A.FUSE_TOP_DX0, False
            try:
                c = run(True, **kw)
            finally:
                A.FUSE_TOP_DX0 = old_top
            assert torch.equal(c[-1][1], b[-1][1])
        a, b = run(True, tap=True), run(False, tap=True)
        for (name, u), (_, v) in zip(a, b):
            # (the corrected dL/dx0 went through two bf16 roundings instead of one)
            torch.testing.assert_close(u.float(), v.float(), rtol=2.0 ** -6, atol=1e-6 if not name == "x" else 1e-7)
    finally:
        D.gemm_cross_bwd = real


def test_dense_layer_above_a_cross_stack_runs_the_top_layers_elementwise_backward_in_its_data_gradient():
    """The top MLP of the ml_perf model is fed the cross stack's output (examples/ml_perf/model.py:209-211): its first
    Dense layer's data-gradient product dx = dz K^T IS dL/dy of the top cross layer, so that layer's elementwise backward
    rides in the product's epilogue too (krs_gemm_cross_bwd without a residual).  Same gradients as the separate passes."""
    from keras_rs_amd import autograd as A
    from keras_rs_amd import dense_ops as D
    from keras_rs_amd.layers import base as kl_base

    kl = _layers()
    g = torch.Generator(device=DEV).manual_seed(31)
    B, d, p = 16384, 768, 256
    x0 = (torch.randn(B, d, device=DEV, generator=g) * 0.5).to(torch.bfloat16)
    calls = []
    real = D.gemm_cross_bwd

    deferred = []

    def spy(*a, **k):
        calls.append(a[2] is None)
        deferred.append((k.get("want_dx0", True), k.get("u_upper") is not None))
        return real(*a, **k)

    def run(fuse, n_cross):
        old, A.FUSE_CROSS_BWD = A.FUSE_CROSS_BWD, fuse
        try:
            cross = [kl.FeatureCross(projection_dim=p, kernel_initializer=kl_base.GlorotUniform(seed=60 + i),
                                     bias_initializer=kl_base.RandomUniform(-0.1, 0.1, seed=70 + i),
                                     dtype="mixed_bfloat16") for i in range(n_cross)]
            mlp = [kl.Dense(256, activation="relu", kernel_initializer=kl_base.GlorotUniform(seed=80), dtype="mixed_bfloat16"),
                   kl.Dense(1, activation="sigmoid", kernel_initializer=kl_base.GlorotUniform(seed=81), dtype="mixed_bfloat16")]
            x = x0.clone().requires_grad_()
            xl = x
            for layer in cross:
                xl = layer(x, xl)
            y = mlp[1](mlp[0](xl))
            y.float().mean().backward()
            torch.cuda.synchronize()
            return [(f"{i}.{n}", q.grad.clone()) for i, layer in enumerate(cross + mlp) for n, q in layer.named_parameters()] \
                + [("x", x.grad.clone())]
        finally:
            A.FUSE_CROSS_BWD = old

    D.gemm_cross_bwd = spy
    try:
        for n_cross, expect in ((2, [True, False]), (1, [True])):   # (one layer: it is also the bottom, x is x0)
            del calls[:], deferred[:]
            a = run(True, n_cross)
            assert calls == expect, calls                 # the Dense layer's product (no residual), then layer 2 -> layer 1
            # two layers: the top cross layer's own term of dL/dx0 is not written by the Dense layer's launch but computed
            # in the next one (u_upper); a single layer has no next launch
            assert deferred == ([(False, False), (True, True)] if n_cross == 2 else [(True, False)]), deferred
            b = run(False, n_cross)
            for (name, u), (_, v) in zip(a, b):
                if name.endswith("bias") and int(name[0]) < n_cross:
                    torch.testing.assert_close(u, v, rtol=1e-5, atol=1e-7)
                elif name == "x" and n_cross == 2:
                    # (that sum is rounded once instead of twice: one bf16 ulp; bit-equal with the switch off, below)
                    # -- of the larger of the two terms, which may cancel: bounded by the matrix's largest entry)
                    torch.testing.assert_close(u.float(), v.float(), rtol=2.0 ** -7,
                                               atol=2.0 ** -7 * float(v.float().abs().max()))
                else:
                    assert torch.equal(u, v), name
            old_top, A.FUSE_TOP_DX0 = A.FUSE_TOP_DX0, False
            try:
                del deferred[:]
                c = run(True, n_cross)
            finally:
                A.FUSE_TOP_DX0 = old_top
            assert all(d == (True, False) for d in deferred), deferred
            for (name, u), (_, v) in zip(c, b):
                if not name.endswith("bias"):
                    assert torch.equal(u, v), name
    finally:
        D.gemm_cross_bwd = real


def test_dense_above_a_cross_stack_whose_top_output_has_a_second_consumer():
    """The Dense layer's launch ran the top cross layer's elementwise backward and LEFT its term of dL/dx0 to that layer's
    own product (deferred); when the top output has another consumer, the gradient autograd hands that layer is not the
    Dense layer's G: the layer must notice (another tensor / a moved version) and redo its backward from the summed
    gradient.  Same gradients as the unfused passes, to the rounding of the two ways of summing."""
    from keras_rs_amd import autograd as A
    from keras_rs_amd.layers import base as kl_base

    kl = _layers()
    g = torch.Generator(device=DEV).manual_seed(37)
    B, d, p = 16384, 768, 256
    x0 = (torch.randn(B, d, device=DEV, generator=g) * 0.5).to(torch.bfloat16)

    def run(fuse):
        old, A.FUSE_CROSS_BWD = A.FUSE_CROSS_BWD, fuse
        try:
            cross = [kl.FeatureCross(projection_dim=p, kernel_initializer=kl_base.GlorotUniform(seed=90 + i),
                                     bias_initializer=kl_base.RandomUniform(-0.1, 0.1, seed=95 + i),
                                     dtype="mixed_bfloat16") for i in range(2)]
            mlp = kl.Dense(64, activation="relu", kernel_initializer=kl_base.GlorotUniform(seed=99), dtype="mixed_bfloat16")
            x = x0.clone().requires_grad_()
            xl = x
            for layer in cross:
                xl = layer(x, xl)
            loss = mlp(xl).float().mean() + xl.float().pow(2).mean() * 0.5      # the second consumer of the top output
            loss.backward()
            torch.cuda.synchronize()
            return [(f"{i}.{n}", q.grad.clone()) for i, layer in enumerate(cross + [mlp]) for n, q in layer.named_parameters()] \
                + [("x", x.grad.clone())]
        finally:
            A.FUSE_CROSS_BWD = old

    a, b = run(True), run(False)
    for (name, u), (_, v) in zip(a, b):
        scale = float(v.float().abs().max())
        torch.testing.assert_close(u.float(), v.float(), rtol=2.0 ** -6, atol=2.0 ** -7 * scale, msg=lambda m: f"{name}: {m}")


@pytest.mark.parametrize("policy,acts", [("mixed_bfloat16", ("relu", "relu", "relu", "sigmoid")), ("float32", ("tanh", "relu", None, "sigmoid"))])
def test_stacked_dense_layers_run_the_lower_layers_activation_backward_in_the_upper_data_gradient(policy, acts):
    """Round 6 (review item 5): the data-gradient product dx = dz K^T of a Dense layer on top of another Dense layer runs that
    layer's dz = dL/dy * act'(y) and bias gradient in its epilogue (krs_gemm_cross_bwd, dense form: dL/dy is never written).
    Every weight / bias / input gradient of a 4-layer stack (units that take the fused ring kernel in bf16 and the two-call
    form in fp32 / for the 1-unit layer) against the same stack with the fusion switched off: dz and dx bit for bit -- the
    derivative is applied to the same once-rounded product --, bias gradients to fp32 summation order."""
    from keras_rs_amd import autograd as A
    from keras_rs_amd.layers import base as kl_base

    kl = _layers()
    g = torch.Generator(device=DEV).manual_seed(41)
    B, d, units = 16384 + 40, 768, (512, 512, 256, 1)
    dt = torch.bfloat16 if policy == "mixed_bfloat16" else torch.float32
    x0 = (torch.randn(B, d, device=DEV, generator=g) * 0.5).to(dt)
    gy = (torch.randn(B, 1, device=DEV, generator=g) * 0.1).to(dt)

    def run(fuse):
        old, A.FUSE_DENSE_BWD = A.FUSE_DENSE_BWD, fuse
        try:
            mlp = [kl.Dense(u, activation=a, kernel_initializer=kl_base.GlorotUniform(seed=60 + i),
                            bias_initializer=kl_base.RandomUniform(-0.1, 0.1, seed=70 + i), dtype=policy)
                   for i, (u, a) in enumerate(zip(units, acts))]
            x = x0.clone().requires_grad_()
            h = x
            for layer in mlp:
                h = layer(h)
            h.backward(gy)
            torch.cuda.synchronize()
            return [(f"{i}.{n}", q.grad.clone()) for i, layer in enumerate(mlp) for n, q in layer.named_parameters()] + [("x", x.grad.clone())]
        finally:
            A.FUSE_DENSE_BWD = old

    a, b = run(True), run(False)
    for (name, u), (_, v) in zip(a, b):
        if name.endswith("bias"):
            torch.testing.assert_close(u, v, rtol=1e-4, atol=1e-4 * float(v.abs().max() + 1e-6), msg=lambda m: f"{name}: {m}")
        else:
            assert torch.equal(u, v), name
    # and against float64 autograd of the same stack (the fused path is what ships: this is the parity statement)
    if policy == "float32":
        mlp_ref = [kl.Dense(u, activation=a, kernel_initializer=kl_base.GlorotUniform(seed=60 + i),
                            bias_initializer=kl_base.RandomUniform(-0.1, 0.1, seed=70 + i), dtype=policy)
                   for i, (u, a) in enumerate(zip(units, acts))]
        for layer in mlp_ref:
            layer.build((B, d if layer is mlp_ref[0] else mlp_ref[mlp_ref.index(layer) - 1].units))
        xr = x0.double().requires_grad_()
        h = xr
        for layer, act in zip(mlp_ref, acts):
            h = h @ layer.kernel.double() + layer.bias.double()
            h = {"relu": torch.relu, "tanh": torch.tanh, "sigmoid": torch.sigmoid, None: lambda t: t}[act](h)
        h.backward(gy.double())
        # (fp32 against float64: a pre-activation within rounding of a relu's kink takes the other branch -- a handful of
        #  elements; everywhere else the two agree to fp32 accuracy)
        got, ref = dict(a)["x"].double(), xr.grad
        close = torch.isclose(got, ref, rtol=2e-4, atol=2e-6)
        assert float(close.double().mean()) > 0.999 and float((got - ref).abs().max()) < 1e-3


def test_a_dense_output_with_a_second_consumer_or_a_watcher_still_gets_the_right_gradient():
    """The upper Dense layer hands autograd the LOWER layer's dz instead of dL/dy.  With a second consumer of that output the
    engine sums the other gradient into it: the lower backward notices (another tensor) and sends the difference through the
    derivative.  With retain_grad() on the output nothing is fused and .grad is the real dL/dy."""
    from keras_rs_amd import autograd as A
    from keras_rs_amd.layers import base as kl_base

    kl = _layers()
    g = torch.Generator(device=DEV).manual_seed(43)
    B, d = 16384, 512
    x0 = (torch.randn(B, d, device=DEV, generator=g) * 0.5).to(torch.bfloat16)

    def run(fuse, watch):
        old, A.FUSE_DENSE_BWD = A.FUSE_DENSE_BWD, fuse
        try:
            lo = kl.Dense(512, activation="sigmoid", kernel_initializer=kl_base.GlorotUniform(seed=81), dtype="mixed_bfloat16")
            hi = kl.Dense(256, activation="relu", kernel_initializer=kl_base.GlorotUniform(seed=82), dtype="mixed_bfloat16")
            x = x0.clone().requires_grad_()
            y = lo(x)
            if watch:
                y.retain_grad()
            loss = hi(y).float().mean() + (0.0 if watch else y.float().pow(2).mean() * 0.5)     # the second consumer
            loss.backward()
            torch.cuda.synchronize()
            out = [(f"{i}.{n}", q.grad.clone()) for i, layer in enumerate((lo, hi)) for n, q in layer.named_parameters()] + [("x", x.grad.clone())]
            return out + ([("y", y.grad.clone())] if watch else [])
        finally:
            A.FUSE_DENSE_BWD = old

    for watch in (False, True):
        a, b = run(True, watch), run(False, watch)
        for (name, u), (_, v) in zip(a, b):
            if watch:
                assert torch.equal(u, v), name       # nothing was fused: the same launches
            else:
                scale = float(v.float().abs().max())
                torch.testing.assert_close(u.float(), v.float(), rtol=2.0 ** -6, atol=2.0 ** -7 * scale, msg=lambda m: f"{name}: {m}")


def test_training_step_leaves_no_cyclic_garbage_that_holds_device_tensors():
    """bench.py times its steps with the cyclic collector off: anything a step leaves in a reference cycle then stays
    allocated (round 4: a recursive closure of the output packing held the 453 MB lookup slab of every step, two fresh
    hipMallocs per step).  With the collector off, a few steps must not grow the allocator's reservation, and a collection
    afterwards must find no tensor."""
    import gc

    kl = _layers()
    from keras_rs_amd.layers import base as kl_base

    B, D, hots, vocabs = 4096, 32, [3, 1, 2], [500, 300, 1000]
    feats = {}
    for t in range(3):
        tc = kl.TableConfig(name=f"t{t}", vocabulary_size=vocabs[t], embedding_dim=D, optimizer=kl.Adagrad(0.05, 0.1),
                            initializer=kl_base.RandomUniform(-0.05, 0.05, seed=t), combiner="sum", placement="sparsecore")
        feats[f"f{t}"] = kl.FeatureConfig(f"f{t}", tc, (B, hots[t]), (B, D))
    emb = kl.DistributedEmbedding(feats, dtype="bfloat16", slab_lead_cols=D)
    dot = kl.DotInteraction(dtype="bfloat16")
    cross = [kl.FeatureCross(projection_dim=16, dtype="mixed_bfloat16") for _ in range(2)]
    g = torch.Generator(device=DEV).manual_seed(1)
    ids = {f"f{t}": torch.randint(0, vocabs[t], (B, hots[t]), device=DEV, generator=g, dtype=torch.int32) for t in range(3)}
    dense = torch.rand(B, D, device=DEV, generator=g).to(torch.bfloat16)
    pre = emb.preprocess(ids)

    def step():
        out = emb(pre)
        fs = [dense] + [out[k] for k in out]
        inter = dot(fs)
        x0 = kl.concat_features(fs)
        xl = x0
        for layer in cross:
            xl = layer(x0, xl)
        (xl.float().mean() + inter.float().mean()).backward()
        for layer in cross:
            for p in layer.parameters():
                p.grad = None

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    gc.collect()
    gc.disable()
    old_flags = gc.get_debug()
    try:
        before = torch.cuda.memory_reserved()
        for _ in range(6):
            step()
        torch.cuda.synchronize()
        assert torch.cuda.memory_reserved() == before
        gc.set_debug(gc.DEBUG_SAVEALL)
        gc.collect()
        leaked = [o for o in gc.garbage if isinstance(o, torch.Tensor)]
        assert not leaked, [(tuple(t.shape), t.dtype) for t in leaked][:8]
    finally:
        gc.set_debug(old_flags)
        gc.garbage.clear()
        gc.enable()


def test_a_forward_without_a_backward_leaves_no_stale_bookkeeping():
    """per-weight `_krs_pending_cross` and the per-bags plan-workspace flag were undone in backward only.
    A grad-enabled forward whose graph is dropped (a validation pass without no_grad, an exception) must give them back:
    otherwise the weight stays off the second stream for good and every later step allocates a fresh plan workspace."""
    import gc

    kl = _layers()
    B, D = 256, 32
    tc = kl.TableConfig(name="t", vocabulary_size=400, embedding_dim=D, optimizer=kl.Adagrad(0.05, 0.1), combiner="sum",
                        placement="sparsecore")
    emb = kl.DistributedEmbedding({"f": kl.FeatureConfig("f", tc, (B, 2), (B, D))}, dtype="bfloat16")
    cross = kl.FeatureCross(projection_dim=16, dtype="mixed_bfloat16")
    g = torch.Generator(device=DEV).manual_seed(3)
    ids = {"f": torch.randint(0, 400, (B, 2), device=DEV, generator=g, dtype=torch.int32)}
    pre = emb.preprocess(ids)
    bags = emb._groups["sparsecore"][0].bags

    def forward():
        x0 = emb(pre)["f"]
        return cross(x0, x0)

    y = forward()
    y.float().sum().backward()          # builds the weights, one ordinary step
    assert getattr(cross.kernel, "_krs_pending_cross", 0) == 0 and not getattr(bags, "_plan_ws_busy", False)
    ws0 = bags._plan_ws.data_ptr()
    for _ in range(3):                  # validation-style passes: grad mode on, no backward, outputs dropped
        y = forward()
        assert cross.kernel._krs_pending_cross == 1 and bags._plan_ws_busy
        del y
        gc.collect()
        assert cross.kernel._krs_pending_cross == 0 and cross.down_kernel._krs_pending_cross == 0
        assert not bags._plan_ws_busy
    y = forward()                       # the next training step re-uses the kept workspace and counts from zero
    assert cross.kernel._krs_pending_cross == 1 and bags._plan_ws.data_ptr() == ws0
    y.float().sum().backward()
    assert cross.kernel._krs_pending_cross == 0 and not bags._plan_ws_busy
    torch.cuda.synchronize()


def test_a_layer_with_more_features_than_one_fused_launch_takes_is_split_into_groups():
    """K2's vector apply kernel keeps 512 feature / table descriptors of a launch in LDS; a wider group used to
    fall to the any-shape kernel (no hot-row path).  The layer now splits such a group between tables -- 700 tables of one width
    here, two of them shared by two features each: two groups, every lookup and every fused SGD update equal to the oracle's,
    features of a shared table in one group."""
    from keras_rs_amd.layers import distributed_embedding as de
    from oracle import krs_oracle as ko

    kl = _layers()
    rng = np.random.default_rng(5)
    T, V, D, B = 700, 40, 8, 16
    tcs = [kl.TableConfig(f"t{t}", V, D, optimizer=kl.SGD(0.5), combiner="sum", placement="sparsecore") for t in range(T)]
    fcs = {f"f{t}": kl.FeatureConfig(f"f{t}", tcs[t], (B, 2), (B, D)) for t in range(T)}
    fcs["g0"] = kl.FeatureConfig("g0", tcs[3], (B, 1), (B, D))          # shared tables
    fcs["g1"] = kl.FeatureConfig("g1", tcs[650], (B, 3), (B, D))
    emb = kl.DistributedEmbedding(fcs)
    groups = emb._groups["sparsecore"]
    assert len(groups) == 2 and all(len(g.paths) <= de.MAX_GROUP_DESCRIPTORS and len(g.table_configs) <= de.MAX_GROUP_DESCRIPTORS for g in groups)
    assert sum(len(g.paths) for g in groups) == T + 2 and sum(len(g.table_configs) for g in groups) == T
    where = {p: gi for gi, g in enumerate(groups) for p in g.paths}
    assert where["g0"] == where["f3"] and where["g1"] == where["f650"]
    ids = {k: rng.integers(0, V, fc.input_shape).astype(np.int32) for k, fc in fcs.items()}
    out = emb(ids)
    tables = {k: v.cpu().numpy().copy() for k, v in emb.get_embedding_tables().items()}
    gout = {k: rng.uniform(-1, 1, (B, D)).astype(np.float32) for k in fcs}
    for k, fc in fcs.items():
        np.testing.assert_allclose(out[k].detach().cpu().numpy(), ko.embed_reduce(tables[fc.table.name], ids[k], None, "sum"), rtol=1e-6, atol=1e-6)
    sum((out[k] * torch.from_numpy(gout[k]).to(DEV)).sum() for k in fcs).backward()
    after = emb.get_embedding_tables()
    for t in (0, 3, 350, 511, 512, 650, 699):
        exp = tables[f"t{t}"].astype(np.float64).copy()
        for k, fc in fcs.items():
            if fc.table is tcs[t]:
                np.add.at(exp, ids[k].reshape(-1), -0.5 * np.repeat(gout[k], ids[k].shape[1], axis=0))
        np.testing.assert_allclose(after[f"t{t}"].cpu().numpy(), exp, rtol=1e-5, atol=1e-6)


def test_scheduled_learning_rates_of_many_tables_reach_the_kernels_through_device_memory():
    """krs_store_f32 carries at most 32 values per launch: 40 tables with their OWN schedules (two launches per update) and SGD,
    three eager updates -- every table must have moved by its own lr_t(step) * gradient at every step (the rates are written
    into the descriptor array on the device; a stale or misplaced rate shows as a wrong row)."""
    kl = _layers()
    rng = np.random.default_rng(9)
    T, V, D, B = 40, 30, 8, 12
    def make_schedule(t):          # (one positional argument: the update count -- jax/config_conversion.py:136-176)
        return lambda step: (0.5 + 0.01 * t) / (1.0 + step)

    sched = [make_schedule(t) for t in range(T)]
    tcs = [kl.TableConfig(f"t{t}", V, D, optimizer=kl.SGD(sched[t]), combiner="sum", placement="sparsecore") for t in range(T)]
    fcs = {f"f{t}": kl.FeatureConfig(f"f{t}", tcs[t], (B, 1), (B, D)) for t in range(T)}
    emb = kl.DistributedEmbedding(fcs)
    ids = {k: rng.integers(0, V, (B, 1)).astype(np.int32) for k in fcs}
    exp = None
    for step in range(3):
        out = emb(ids)
        if exp is None:
            exp = {k: v.cpu().numpy().astype(np.float64).copy() for k, v in emb.get_embedding_tables().items()}
        gout = {k: rng.uniform(-1, 1, (B, D)).astype(np.float32) for k in fcs}
        sum((out[k] * torch.from_numpy(gout[k]).to(DEV)).sum() for k in fcs).backward()
        for t in range(T):
            np.add.at(exp[f"t{t}"], ids[f"f{t}"].reshape(-1), -np.float32(sched[t](step)) * gout[f"f{t}"].astype(np.float64))
        got = emb.get_embedding_tables()
        for t in (0, 1, 31, 32, 39):
            np.testing.assert_allclose(got[f"t{t}"].cpu().numpy(), exp[f"t{t}"], rtol=2e-5, atol=2e-6, err_msg=f"step {step} table {t}")
    assert emb._groups["sparsecore"][0].step == 3
