"""keras_rs_amd.keras_adapter against tests/keras_stub.py (a stand-in for the Keras symbols it touches; keras is
not installed in the image).  Host part: constructor / error / weight-order / config contract of the reference's
tests (feature_cross_test.py:21-65, dot_interaction_test.py:93-105).  GPU part: the adapter layers give the same
numbers as the torch-native layers, forward and backward, with the reference's known-answer vectors."""

import json
import os

import numpy as np
import pytest
import torch

from tests import keras_stub

KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat.json")))


def _adapter(device="cpu"):
    from keras_rs_amd import keras_adapter as ka

    keras_stub.Layer.DEVICE = device
    return ka.make_layers(keras_stub)


def test_feature_cross_contract_on_host():
    A = _adapter()
    with pytest.raises(ValueError):                      # feature_cross_test.py:62-65
        A.FeatureCross(diag_scale=-1.0)
    for case in KAT["feature_cross"]["cases"]:
        layer = A.FeatureCross(projection_dim=case["projection_dim"], diag_scale=case["diag_scale"],
                               kernel_initializer="ones")
        layer.build((1, 3))
        assert [list(w.shape) for w in layer.weights] == case["weight_shapes"]      # :21-47 weight order
    layer = A.FeatureCross(projection_dim=2, use_bias=False)
    layer.build((4, 6))
    assert [w.shape for w in layer.weights] == [(6, 2), (2, 6)]
    cfg = A.FeatureCross(projection_dim=3, diag_scale=0.5, pre_activation="relu").get_config()
    for key in ("projection_dim", "diag_scale", "use_bias", "pre_activation", "kernel_initializer",
                "bias_initializer", "kernel_regularizer", "bias_regularizer"):           # feature_cross.py:196-222
        assert key in cfg
    assert cfg["pre_activation"] == "relu" and cfg["kernel_initializer"] == "glorot_uniform"
    with pytest.raises(ValueError):                      # shape mismatch is checked before any device work (:54-60)
        A.FeatureCross()(torch.zeros(12, 5), torch.zeros(12, 7))


def test_dot_interaction_contract_on_host():
    A = _adapter()
    layer = A.DotInteraction(self_interaction=True)
    assert layer.compute_output_shape([(8, 5)] * 3) == (8, 6)
    assert A.DotInteraction(skip_gather=True).compute_output_shape([(8, 5)] * 3) == (8, 9)
    with pytest.raises(ValueError):                      # dot_interaction_test.py:93-98
        layer([torch.zeros(3), torch.zeros(3)])
    with pytest.raises(ValueError):                      # :100-105
        layer([torch.zeros(1, 3), torch.zeros(1, 4)])
    assert layer.get_config()["self_interaction"] is True


def test_layers_refuses_without_keras():
    from keras_rs_amd import keras_adapter as ka

    with pytest.raises(ImportError):
        ka.layers()


@pytest.mark.gpu
def test_adapter_layers_match_native_layers_on_gpu():
    import keras_rs_amd.layers as kl

    dev = "cuda:0"
    A = _adapter(dev)
    fc = KAT["feature_cross"]
    x0 = torch.tensor(fc["x0"], device=dev)
    x = torch.tensor(fc["x"], device=dev)
    for case in fc["cases"]:
        layer = A.FeatureCross(projection_dim=case["projection_dim"], diag_scale=case["diag_scale"],
                               kernel_initializer="ones")
        out = layer(x0) if case["one_input"] else layer(x0, x)
        np.testing.assert_allclose(out.detach().cpu().numpy(), np.array(case["expected"], np.float32), atol=1e-6, rtol=1e-6)
    # a stack of two low-rank layers: forward and every gradient equal the torch-native layers' (same kernels)
    g = torch.Generator(device=dev).manual_seed(0)
    a0 = torch.randn(33, 24, device=dev, generator=g).requires_grad_()
    b0 = a0.detach().clone().requires_grad_()
    ad = [A.FeatureCross(projection_dim=8, pre_activation="tanh"), A.FeatureCross(projection_dim=8)]
    na = [kl.FeatureCross(projection_dim=8, pre_activation="tanh"), kl.FeatureCross(projection_dim=8)]
    ya = ad[1](a0, ad[0](a0, a0))
    for layer, ref in zip(ad, na):
        ref.build((33, 24))
        with torch.no_grad():
            for w, r in zip(layer.weights, ref.weights):
                r.copy_(w.value)
    yb = na[1](b0, na[0](b0, b0))
    assert torch.equal(ya, yb)
    go = torch.randn(33, 24, device=dev, generator=g)
    ya.backward(go)
    yb.backward(go)
    assert torch.equal(a0.grad, b0.grad)
    for layer, ref in zip(ad, na):
        for w, r in zip(layer.weights, ref.weights):
            # (the bias gradient is combined with fp32 atomics across workgroups: last-bit differences between runs)
            torch.testing.assert_close(w.value.grad, r.grad, rtol=1e-5, atol=1e-6)
    # DotInteraction known answers
    di = KAT["dot_interaction"]
    feats = [torch.tensor(np.asarray(f, np.float32).reshape(1, -1), device=dev) for f in di["inputs"]]
    if True:
        for case in di["cases"]:
            out = A.DotInteraction(self_interaction=case["self_interaction"], skip_gather=case["skip_gather"])(feats)
            np.testing.assert_allclose(out.cpu().numpy().reshape(-1), np.array(case["expected"], np.float32).reshape(-1), atol=1e-5, rtol=1e-5)
    # DistributedEmbedding wrapper: same numbers as the native layer, tables visible as Keras weights
    t = kl.TableConfig("t", 23, 7, placement="sparsecore", optimizer="sgd", combiner="mean")
    layer = A.DistributedEmbedding({"f": kl.FeatureConfig("f", t, (4, 2), (4, 7))})
    ids = np.array([[2, 3], [4, 5], [2, 2], [9, 1]], np.int32)
    out = layer({"f": ids})["f"]
    tab = layer.get_embedding_tables()["t"]
    torch.testing.assert_close(out, tab[torch.from_numpy(ids).long().to(dev)].mean(1), rtol=1e-5, atol=1e-6)
    assert any(w.shape == (23, 7) for w in layer.weights)


@pytest.mark.gpu
@pytest.mark.parametrize("opt", ["adagrad", "adam"])
def test_adapter_checkpoint_round_trip_keeps_slots_and_iterations(opt):
    """save_own_variables / load_own_variables (what .keras / .weights.h5 checkpoints call per layer): tables, the fused
    optimizer's slot planes AND its step count survive -- a restored layer continues bit for bit where the saved one
    stopped (Adam's bias correction depends on the step count, Adagrad on its accumulators)."""
    import keras_rs_amd.layers as kl

    dev = "cuda:0"
    A = _adapter(dev)

    def make():
        o = kl.Adagrad(0.1, 0.1) if opt == "adagrad" else kl.Adam(0.05)
        t = kl.TableConfig("t", 31, 8, placement="sparsecore", optimizer=o, combiner="sum")
        return A.DistributedEmbedding({"f": kl.FeatureConfig("f", t, (6, 3), (6, 8))})

    rng = np.random.default_rng(0)
    batches = [rng.integers(0, 31, (6, 3)).astype(np.int32) for _ in range(4)]
    grads = [torch.from_numpy(rng.uniform(-1, 1, (6, 8)).astype(np.float32)).to(dev) for _ in range(4)]

    def step(layer, i):
        out = layer({"f": batches[i]})["f"]
        (out * grads[i]).sum().backward()

    a = make()
    for i in range(2):
        step(a, i)
    store = {}
    a.save_own_variables(store)
    assert any(k.startswith("iterations/") for k in store) and sum(v.ndim >= 2 for v in store.values()) >= 2
    b = make()
    b.load_own_variables(store)
    for i in range(2, 4):
        step(a, i)
        step(b, i)
    torch.cuda.synchronize()
    assert torch.equal(a.get_embedding_tables()["t"], b.get_embedding_tables()["t"])
    sa, sb = a._impl.state_dict(), b._impl.state_dict()
    for k in sa:
        if k != "_extra_state":
            assert torch.equal(sa[k], sb[k]), k
    assert sa["_extra_state"] == sb["_extra_state"]
    c = make()
    for i in range(2, 4):      # a fresh layer (reset slots, step 0) does NOT reproduce it: the state matters
        step(c, i)
    assert not torch.equal(a.get_embedding_tables()["t"], c.get_embedding_tables()["t"])
