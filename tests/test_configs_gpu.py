"""BASELINE.json configs 1 and 2 as parity cases (not bench lines):
  C1  README quickstart: Embedding(32, 6) -> FeatureCross x2 -> Dense(10), batch 2      (README.md:46-76)
  C2  8 tables x 100,000 x 64 fp32, 3 full-rank FeatureCross on d = 512, fp32, batch 8192 (the stated
      configuration: the OpenMP oracle and the float64 torch composition take a few seconds on it).
The HIP path (layers -> C ABI) is compared with the oracle (forward) and with a float64 torch
composition of the same formulas on the CPU (gradients), to the north-star tolerance 1e-5."""

import numpy as np
import pytest
import torch

from oracle import krs_oracle as ko

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cross64(x0, x, kernel, bias):
    return x0 * (x @ kernel + bias) + x


def test_c1_readme_quickstart_forward_backward():
    import keras_rs_amd.layers as kl

    torch.manual_seed(0)
    emb = kl.Embedding(32, 6)
    c1, c2 = kl.FeatureCross(), kl.FeatureCross()
    head = torch.nn.Linear(6, 10).to(DEV)
    ids = torch.tensor([3, 17], device=DEV)
    x0 = emb(ids)
    x2 = c2(x0, c1(x0, x0))
    logits = head(x2)
    logits.sum().backward()
    # oracle forward
    e = emb.embeddings.detach().cpu().numpy()
    w1 = [w.detach().cpu().numpy() for w in c1.weights]
    w2 = [w.detach().cpu().numpy() for w in c2.weights]
    x0n = ko.embed_reduce(e, ids.cpu().numpy().astype(np.int32), None, "sum")
    x1n = ko.feature_cross(x0n, x0n, w1[0], w1[1])
    x2n = ko.feature_cross(x0n, x1n, w2[0], w2[1])
    np.testing.assert_allclose(x2.detach().cpu().numpy(), x2n, rtol=1e-6, atol=1e-6)
    # float64 gradients
    E = emb.embeddings.detach().double().cpu().requires_grad_()
    W = [w.detach().double().cpu().requires_grad_() for w in c1.weights + c2.weights]
    r0 = E[ids.cpu()]
    r2 = _cross64(r0, _cross64(r0, r0, W[0], W[1]), W[2], W[3])
    (r2 @ head.weight.detach().double().cpu().T + head.bias.detach().double().cpu()).sum().backward()
    np.testing.assert_allclose(emb.embeddings.grad.cpu().numpy(), E.grad.numpy(), rtol=1e-5, atol=1e-6)
    for got, ref in zip(c1.weights + c2.weights, W):
        np.testing.assert_allclose(got.grad.cpu().numpy(), ref.grad.numpy(), rtol=1e-5, atol=1e-6)


def test_c2_eight_tables_three_full_rank_cross_layers_fp32():
    import keras_rs_amd.layers as kl
    from keras_rs_amd.layers import base

    T, V, D, B = 8, 100_000, 64, 8192
    rng = np.random.default_rng(1338)
    tcs = [kl.TableConfig(f"t{t}", V, D, initializer=base.RandomUniform(-0.05, 0.05, seed=1337 + t),
                          combiner="sum", placement="default_device") for t in range(T)]
    fcs = {f"f{t}": kl.FeatureConfig(f"f{t}", tcs[t], (B,), (B, D)) for t in range(T)}
    emb = kl.DistributedEmbedding(fcs)
    ids = {f"f{t}": rng.integers(0, V, B).astype(np.int32) for t in range(T)}
    layers = [kl.FeatureCross(kernel_initializer=base.GlorotUniform(seed=7 + i), bias_initializer="uniform")
              for i in range(3)]
    out = emb(ids)
    x0 = torch.cat([out[k] for k in out], dim=-1)  # [B, 512]
    xl = x0
    for layer in layers:
        xl = layer(x0, xl)
    g = torch.from_numpy(rng.uniform(-1, 1, (B, T * D)).astype(np.float32)).to(DEV)
    xl.backward(g)

    # forward against the oracle
    tables = {k: v.cpu().numpy() for k, v in emb.get_embedding_tables().items()}
    x0n = np.concatenate([ko.embed_reduce(tables[f"t{t}"], ids[f"f{t}"], None, "sum") for t in range(T)], axis=1)
    np.testing.assert_array_equal(x0.detach().cpu().numpy(), x0n)  # the gather is bit-exact
    xn = x0n
    for layer in layers:
        w = [p.detach().cpu().numpy() for p in layer.weights]
        xn = ko.feature_cross(x0n, xn, w[0], w[1])
    np.testing.assert_allclose(xl.detach().cpu().numpy(), xn, rtol=1e-5, atol=1e-5)

    # gradients against float64 torch on the CPU
    W = [[p.detach().double().cpu().requires_grad_() for p in layer.weights] for layer in layers]
    r0 = torch.from_numpy(x0n).double().requires_grad_()
    rl = r0
    xs, us = [], []
    for k, b in W:
        u = rl @ k + b                      # (_cross64 spelled out: the pre-activation's gradient is looked at below)
        u.retain_grad()
        xs.append(rl.detach())
        us.append(u)
        rl = r0 * u + rl
    rl.backward(g.double().cpu())
    # (weight gradients are fp32 sums of 8192 products of magnitude <= 1: 1e-4 of the column scale, ~ 1e-4 absolute)
    for layer, (k, b) in zip(layers, W):
        np.testing.assert_allclose(layer.weights[0].grad.cpu().numpy(), k.grad.numpy(), rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(layer.weights[1].grad.cpu().numpy(), b.grad.numpy(), rtol=1e-4, atol=1e-4)
        # ... and the north star's 1e-5 as a bound on the gradient as a whole (largest error against largest entry)
        for got, ref in ((layer.weights[0].grad, k.grad), (layer.weights[1].grad, b.grad)):
            assert np.abs(got.cpu().numpy() - ref.numpy()).max() <= 1e-5 * np.abs(ref.numpy()).max()
    # ELEMENT-wise 1e-5 (round-3 review): every entry of dK = x^T dz is a sum of 8192 products; its fp32 error is bounded
    # relative to the sum of the products' MAGNITUDES (entries that cancel to ~0 have no meaningful relative error of
    # their own).  |got - ref|_ij <= 1e-5 * (|x|^T |dz|)_ij for every entry, likewise the bias gradient against sum |dz|.
    for layer, (k, b), x_l, u_l in zip(layers, W, xs, us):
        dz = u_l.grad
        scale_k = (x_l.abs().T @ dz.abs()).numpy()
        err_k = np.abs(layer.weights[0].grad.cpu().numpy().astype(np.float64) - k.grad.numpy())
        assert (err_k <= 1e-5 * scale_k + 1e-30).all(), float((err_k / (scale_k + 1e-30)).max())
        scale_b = dz.abs().sum(0).numpy()
        err_b = np.abs(layer.weights[1].grad.cpu().numpy().astype(np.float64) - b.grad.numpy())
        assert (err_b <= 1e-5 * scale_b + 1e-30).all(), float((err_b / (scale_b + 1e-30)).max())
    # embedding-table gradient = scatter-add of the x0 gradient slices (index work: exact rows, 1e-5 values)
    dx0 = r0.grad.numpy()
    for t in (0, 5):
        dense = np.zeros((V, D))
        np.add.at(dense, ids[f"f{t}"], dx0[:, t * D:(t + 1) * D])
        got = emb.weights[t].grad.cpu().numpy()
        assert np.array_equal(np.nonzero(np.abs(got).sum(1))[0], np.nonzero(np.abs(dense).sum(1))[0])
        np.testing.assert_allclose(got, dense, rtol=1e-4, atol=1e-5)
