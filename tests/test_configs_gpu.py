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
    # Weight gradients, element by element against the float64 composition (round-4 review, next #5a).  Every entry of
    # dK = x^T dz is an fp32 sum of 8192 signed products; what fp32 can promise is an error relative to the sum of the
    # products' MAGNITUDES, S_ij = (|x|^T |dz|)_ij -- an entry whose products cancel (|ref| << S) has no meaningful relative
    # error of its own.  The test SELECTS the entries per element instead of loosening the tolerance globally:
    #   (1) entries that do not cancel by more than 128x (|ref| >= S / 128; about 55 % of dK): the stated tolerance, rtol = 1e-5
    #       (measured on MI355X: worst relative error 5e-6 there; worst |err| / S over ALL entries 4.1e-8, the fp32 CPU twin 2.7e-8);
    #   (2) every entry: |got - ref| <= 1e-5 * S  (the magnitude-scaled 1e-5);
    #   (3) the cancelling entries only: the old element-wise bound rtol = 1e-4, atol = 1e-4;
    #   (4) the worst scaled error is no larger than 4x that of the SAME gradient evaluated in fp32 by torch on the CPU
    #       (the reference's own arithmetic on its CPU path): the HIP path is as close to float64 as fp32 gets.
    # The assertion messages carry the worst errors and the share of entries in each class.
    def check_sum_gradient(name, got, ref, scale, fp32_twin):
        got64 = got.astype(np.float64)
        err = np.abs(got64 - ref)
        well = np.abs(ref) * 128.0 >= scale
        worst_rel = float((err[well] / np.abs(ref[well])).max()) if well.any() else 0.0
        worst_scaled = float((err / (scale + 1e-300)).max())
        twin_scaled = float((np.abs(fp32_twin.astype(np.float64) - ref) / (scale + 1e-300)).max())
        msg = ("%s: %.1f %% of the entries do not cancel (|ref| >= S/128), their worst relative error is %.3g (bound 1e-5); worst "
               "|err| / S over all entries %.3g (bound 1e-5; the fp32 CPU composition: %.3g)"
               % (name, 100.0 * well.mean(), worst_rel, worst_scaled, twin_scaled))
        print(msg)
        assert well.mean() > 0.2, msg
        assert worst_rel <= 1e-5, msg
        assert (err <= 1e-5 * scale + 1e-30).all(), msg
        np.testing.assert_allclose(got64[~well], ref[~well], rtol=1e-4, atol=1e-4, err_msg=msg)
        assert worst_scaled <= 4.0 * twin_scaled + 1e-9, msg

    for li, (layer, (k, b), x_l, u_l) in enumerate(zip(layers, W, xs, us)):
        dz = u_l.grad
        twin_k = (x_l.float().T @ dz.float()).numpy()                 # the same sums in fp32 on the CPU
        twin_b = dz.float().sum(0).numpy()
        check_sum_gradient(f"layer {li} dK", layer.weights[0].grad.cpu().numpy(), k.grad.numpy(),
                           (x_l.abs().T @ dz.abs()).numpy(), twin_k)
        check_sum_gradient(f"layer {li} dbias", layer.weights[1].grad.cpu().numpy(), b.grad.numpy(), dz.abs().sum(0).numpy(), twin_b)
    # embedding-table gradient = scatter-add of the x0 gradient slices (index work: exact rows, 1e-5 values)
    dx0 = r0.grad.numpy()
    for t in (0, 5):
        dense = np.zeros((V, D))
        np.add.at(dense, ids[f"f{t}"], dx0[:, t * D:(t + 1) * D])
        got = emb.weights[t].grad.cpu().numpy()
        assert np.array_equal(np.nonzero(np.abs(got).sum(1))[0], np.nonzero(np.abs(dense).sum(1))[0])
        # values: a touched row is the sum of <= a few slices of dL/dx0, itself a 512-term fp32 sum per element (dz K^T) on top
        # of the direct terms: rtol = 1e-5 against float64 wherever the element does not cancel against the slice's scale
        # (|ref| >= 3e-2 x the largest entry of its row), the magnitude-scaled 1e-5 (of the row's largest entry) everywhere
        err = np.abs(got.astype(np.float64) - dense)
        row_max = np.abs(dense).max(1, keepdims=True)
        well = np.abs(dense) >= 3e-2 * row_max
        well &= row_max > 0
        worst_rel = float((err[well] / np.abs(dense[well])).max())
        worst_scaled = float((err / (row_max + 1e-300)).max())
        msg = "table %d: worst relative error %.3g on %.1f %% of the elements, worst |err| / row max %.3g" % (
            t, worst_rel, 100.0 * well.mean(), worst_scaled)
        print(msg)
        assert worst_rel <= 1e-5 and worst_scaled <= 1e-5, msg
