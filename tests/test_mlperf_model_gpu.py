"""Parity of the ASSEMBLED ml_perf model: examples/dlrm_dcn_v2.py on the HIP layers
against the oracle's restatement of examples/ml_perf/model.py:175-212 -- bottom MLP, the large embeddings in dict
insertion order, the small-table block (plain embeddings pooled with a sum over axis -2, concatenated last), the
cross stack `xl = layer(x0, xl)` (:332-336) and the top MLP.  The concat order is the contract under test: every
feature gets a distinct table, so a permuted slot changes the output.  fp32 policy, tolerance 1e-5 (north star);
gradients against a float64 torch composition of the same formulas, table updates against the oracle's SGD."""

import importlib.util
import os

import numpy as np
import pytest
import torch

from oracle import krs_oracle as ko

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _example():
    spec = importlib.util.spec_from_file_location("dlrm_dcn_v2", os.path.join(ROOT, "examples", "dlrm_dcn_v2.py"))
    ex = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ex)
    return ex


def _np(t):
    return t.detach().float().cpu().numpy()


def _dense_np(x, layers, final):
    for i, layer in enumerate(layers):
        k, b = _np(layer.kernel), _np(layer.bias)
        y, _ = ko.gemm(np.ascontiguousarray(x), np.ascontiguousarray(k), x.shape[0], k.shape[1], k.shape[0], bias=b,
                       act=final if i == len(layers) - 1 else "relu")
        x = y
    return x


def test_assembled_model_matches_reference_composition():
    import keras_rs_amd.layers as kl

    ex = _example()
    B, E = 64, 16
    hots = [3, 1, 2, 5, 1, 2]
    vocabs = [500, 7, 300, 900, 3, 40]                      # 7, 3 and 40 fall under the threshold: small tables
    lr = 0.1
    model = ex.build_model(B, vocabs, hots, embedding_dim=E, projection=8, cross_layers=2, bottom=(32, E),
                           top=(32, 16, 1), table_optimizer=kl.SGD(lr), embedding_threshold=50, dtype="float32",
                           embedding_dtype="float32")
    assert [f["name"] for f in model.small_emb_features] == ["cat_1", "cat_4", "cat_5"]
    rng = np.random.default_rng(3)
    ids = {t: rng.integers(0, vocabs[t], (B, hots[t])).astype(np.int32) for t in range(6)}
    large_keys = [0, 2, 3]
    small_keys = [1, 4, 5]
    inputs = {
        "dense_input": torch.from_numpy(rng.uniform(0, 0.9, (B, 13)).astype(np.float32)).to(DEV),
        "large_emb_inputs": {f"cat_{t:02d}_id": torch.from_numpy(ids[t]).to(DEV) for t in large_keys},
        "small_emb_inputs": {f"cat_{t:02d}_id": torch.from_numpy(ids[t]).to(DEV) for t in small_keys},
    }
    labels = torch.from_numpy((rng.uniform(0, 1, (B, 1)) < 0.3).astype(np.float32)).to(DEV)
    pred = model(inputs)                                               # builds the layers
    tables0 = {k: _np(v).copy() for k, v in model.embedding_layer.get_embedding_tables().items()}
    small0 = {k: _np(layer.embeddings).copy() for k, layer in model.small_embedding_layers.items()}

    # ---- forward: oracle composition in the reference's order (model.py:183-211) ----
    dense_out = _dense_np(_np(inputs["dense_input"]), list(model.bottom_mlp), "relu")
    large = [ko.embed_reduce(tables0[f"cat_{t}"], ids[t], None, "sum") for t in large_keys]
    small = [ko.embed_reduce(small0[f"cat_{t:02d}_id"], ids[t], None, "sum") for t in small_keys]   # Embedding + sum(axis=-2)
    x0 = np.concatenate([dense_out, *large, *small], axis=-1)          # model.py:204-207
    assert x0.shape == (B, E * 7)
    xl = x0
    for layer in model.dcn_block.layers:                               # model.py:332-336
        xl = ko.feature_cross(x0, xl, _np(layer.kernel), _np(layer.bias), _np(layer.down_kernel))
    exp = _dense_np(xl, list(model.top_mlp), "sigmoid")
    np.testing.assert_allclose(_np(pred), exp, rtol=1e-5, atol=1e-5)

    # ---- backward: float64 torch composition of the same formulas on the CPU ----
    loss = kl.binary_crossentropy(labels, pred)      # krs_bce_fwd_bwd: no torch arithmetic left in the step
    np.testing.assert_allclose(float(loss.detach()), float(ko.bce_fwd_bwd(_np(pred), _np(labels))[0]), rtol=2e-6)
    loss.backward()
    d64 = lambda a: torch.from_numpy(np.asarray(a, np.float64)).requires_grad_()  # noqa: E731
    P = {n: d64(_np(p)) for n, p in model.named_parameters() if p.requires_grad}
    T64 = {k: d64(v) for k, v in tables0.items()}

    def mlp64(x, prefix, n, final):
        for i in range(n):
            x = x @ P[f"{prefix}.{i}.kernel"] + P[f"{prefix}.{i}.bias"]
            x = final(x) if i == n - 1 else torch.relu(x)
        return x

    r_dense = mlp64(torch.from_numpy(_np(inputs["dense_input"]).astype(np.float64)), "bottom_mlp", 2, torch.relu)
    r_large = [T64[f"cat_{t}"][torch.from_numpy(ids[t]).long()].sum(1) for t in large_keys]
    r_small = [P[f"small_embedding_layers.cat_{t:02d}_id.embeddings"][torch.from_numpy(ids[t]).long()].sum(1)
               for t in small_keys]
    r0 = torch.cat([r_dense, *r_large, *r_small], dim=-1)
    rl = r0
    for i in range(2):
        U, K, b = (P[f"dcn_block.layers.{i}.{n}"] for n in ("down_kernel", "kernel", "bias"))
        rl = r0 * ((rl @ U) @ K + b) + rl
    r_pred = mlp64(rl, "top_mlp", 3, torch.sigmoid)
    y64 = torch.from_numpy(_np(labels).astype(np.float64))
    r_loss = -(y64 * torch.log(r_pred) + (1 - y64) * torch.log(1 - r_pred)).mean()
    r_loss.backward()
    np.testing.assert_allclose(float(loss.detach()), float(r_loss.detach()), rtol=1e-6)
    for n, p in model.named_parameters():
        if p.requires_grad:
            np.testing.assert_allclose(_np(p.grad), P[n].grad.numpy(), rtol=1e-4, atol=1e-6, err_msg=n)
            # the north star's 1e-5, as a statement about the gradient as a whole: the largest error against the
            # largest entry (element by element an fp32 sum with cancellation has no relative bound at all)
            ref = P[n].grad.numpy()
            assert np.abs(_np(p.grad) - ref).max() <= 1e-5 * np.abs(ref).max(), n
    # the large tables took their fused SGD step inside the backward: table - lr * dense gradient on the touched rows
    after = model.embedding_layer.get_embedding_tables()
    for t in large_keys:
        name = f"cat_{t}"
        exp_t = tables0[name].copy()
        touched = np.zeros(vocabs[t], np.uint8)
        touched[ids[t].reshape(-1)] = 1
        ko.apply_optimizer(exp_t, None, T64[name].grad.numpy().astype(np.float32), touched, lr, "sgd")
        np.testing.assert_allclose(_np(after[name]), exp_t, rtol=1e-5, atol=1e-6)


def test_two_runs_of_the_step_are_bit_identical():
    """Run-to-run reproducibility of the whole training step (mixed_bfloat16, the bench's policy): two models built from
    the same seeds and stepped three times on the same batches end with the same bits in every dense weight, bias,
    optimizer slot and table row -- no fp32 atomics left anywhere in the step (the bias gradients were the last:
    krs_colsum_workspace_bytes), K2 has one owner per row and sums in position order, split-K slabs are reduced in order."""
    import keras_rs_amd.layers as kl

    ex = _example()
    B, E = 4096, 32
    hots = [3, 1, 2, 5, 1, 2, 1, 7]
    vocabs = [5000, 70, 3000, 9000, 30, 400, 20000, 1000]
    rng = np.random.default_rng(11)
    batches = []
    for _ in range(3):
        ids = {t: torch.from_numpy(rng.integers(0, vocabs[t], (B, hots[t])).astype(np.int32)).to(DEV) for t in range(8)}
        dense = torch.from_numpy(rng.uniform(0, 0.9, (B, 13)).astype(np.float32)).to(DEV)
        labels = torch.from_numpy((rng.uniform(0, 1, (B, 1)) < 0.3).astype(np.float32)).to(DEV)
        batches.append((ids, dense, labels))

    def run():
        model = ex.build_model(B, vocabs, hots, embedding_dim=E, projection=64, cross_layers=3, bottom=(64, 32, E),
                               top=(128, 64, 1), table_optimizer=kl.Adagrad(learning_rate=0.05, initial_accumulator_value=0.1),
                               embedding_threshold=100)
        small = {f"cat_{t:02d}_id" for t in range(8) if vocabs[t] < 100}
        box, losses = [None], []
        for ids, dense, labels in batches:
            inputs = {"dense_input": dense,
                      "large_emb_inputs": {f"cat_{t:02d}_id": ids[t] for t in range(8) if f"cat_{t:02d}_id" not in small},
                      "small_emb_inputs": {f"cat_{t:02d}_id": ids[t] for t in range(8) if f"cat_{t:02d}_id" in small}}
            losses.append(ex.train_step(model, box, inputs, labels))
        torch.cuda.synchronize()
        state = {k: v.detach().clone() for k, v in model.state_dict().items() if isinstance(v, torch.Tensor)}
        state.update({f"table.{k}": v.detach().clone() for k, v in model.embedding_layer.get_embedding_tables().items()})
        state.update({f"opt.{i}.{k}": v.clone() for i, st in enumerate(box[0].state.values()) for k, v in st.items()
                      if isinstance(v, torch.Tensor)})
        return losses, state

    la, sa = run()
    lb, sb = run()
    assert [float(x) for x in la] == [float(x) for x in lb]
    assert sa.keys() == sb.keys() and len(sa) > 20
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
