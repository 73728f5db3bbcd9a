"""DLRM-DCN-v2 from the MI355X layers: the model of the reference's ml_perf example
(examples/ml_perf/model.py:30-262, 296-345) with `keras_rs` / `keras.layers` swapped for
`keras_rs_amd.layers` -- bottom MLP -> DistributedEmbedding -> concat -> FeatureCross stack -> top MLP
(sigmoid) -> binary cross-entropy (examples/ml_perf/main.py:201-210).

    python examples/dlrm_dcn_v2.py            # a few training steps on synthetic data (needs an MI355X)

The only edits against the reference model code: `layers.concat_features` instead of
`ops.concatenate` (the embeddings land directly in the interaction input, no copy) with
`slab_lead_cols` reserving the bottom-MLP slot, `kl.binary_crossentropy` for keras.losses.BinaryCrossentropy() and
`keras_rs_amd.optim.Adagrad` for the dense optimizer.
"""

from __future__ import annotations

import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import keras_rs_amd.layers as kl  # noqa: E402
from keras_rs_amd.layers import base  # noqa: E402


def mlp_layers(dims, intermediate_activation, final_activation, seed, dtype):
    """model.py:214-262: Dense stack, VarianceScaling(1.0, fan_in, uniform) kernels and biases."""
    init = lambda: base.VarianceScaling(scale=1.0, mode="fan_in", distribution="uniform", seed=seed)  # noqa: E731
    acts = [intermediate_activation] * (len(dims) - 1) + [final_activation]
    return torch.nn.Sequential(*[kl.Dense(d, activation=a, kernel_initializer=init(), bias_initializer=init(),
                                          dtype=dtype) for d, a in zip(dims, acts)])


class DCNBlock(torch.nn.Module):
    """model.py:296-345."""

    def __init__(self, num_layers, projection_dim, seed, dtype):
        super().__init__()
        self.layers = torch.nn.ModuleList(
            kl.FeatureCross(projection_dim=projection_dim, kernel_initializer=base.GlorotUniform(seed=seed),
                            bias_initializer="zeros", dtype=dtype) for _ in range(num_layers))

    def forward(self, x0):
        xl = x0
        for layer in self.layers:
            xl = layer(x0, xl)
        return xl


class DLRMDCNV2(torch.nn.Module):
    """model.py:30-212.  small_emb_features: the tables under `embedding_threshold` (main.py:135-141), a list of
    {"name", "new_name", "vocabulary_size"}: plain Embedding layers, LecunNormal(seed) tables, bags pooled
    with a sum over axis -2, concatenated behind the large embeddings (model.py:128-148, 185-207)."""

    def __init__(self, large_emb_feature_configs, embedding_dim, bottom_mlp_dims, top_mlp_dims, num_dcn_layers,
                 dcn_projection_dim, seed=1337, dtype="mixed_bfloat16", embedding_dtype="bfloat16",
                 small_emb_features=None):
        super().__init__()
        assert bottom_mlp_dims[-1] == embedding_dim, "the bottom MLP output is one more 'feature' of the interaction"
        self.bottom_mlp = mlp_layers(bottom_mlp_dims, "relu", "relu", seed, dtype)
        self.embedding_layer = kl.DistributedEmbedding(large_emb_feature_configs, table_stacking="auto",
                                                       dtype=embedding_dtype, name="embedding_layer",
                                                       slab_lead_cols=embedding_dim)
        self.small_emb_features = small_emb_features
        self.small_embedding_layers = None
        if small_emb_features:
            # EmbedReduce(combiner="sum") = keras.layers.Embedding followed by ops.sum(axis=-2) (model.py:195-198)
            # as ONE gather+pool launch per feature
            self.small_embedding_layers = torch.nn.ModuleDict({
                f"{f['new_name']}_id": kl.EmbedReduce(f["vocabulary_size"], embedding_dim, combiner="sum",
                                                      embeddings_initializer=base.LecunNormal(seed=seed),
                                                      dtype=dtype,   # (model.py:133-141: no dtype argument = the model's policy; fp32 variables)
                                                      name=f"small_embedding_layer_{f['new_name']}")
                for f in small_emb_features})
        self.dcn_block = DCNBlock(num_dcn_layers, dcn_projection_dim, seed, dtype)
        self.top_mlp = mlp_layers(top_mlp_dims, "relu", "sigmoid", seed, dtype)

    def forward(self, inputs):
        dense_output = self.bottom_mlp(inputs["dense_input"])                     # model.py:183
        large_embeddings = self.embedding_layer(inputs["large_emb_inputs"])      # model.py:184
        emb_dtype = next(iter(large_embeddings.values())).dtype
        x = kl.concat_features([dense_output.to(emb_dtype), *large_embeddings.values()])   # model.py:204-207
        if self.small_embedding_layers is not None:
            small = [self.small_embedding_layers[k](v).to(emb_dtype)                       # model.py:189-201
                     for k, v in inputs["small_emb_inputs"].items()]
            x = torch.cat([x, *small], dim=-1)
        x = self.dcn_block(x)
        return self.top_mlp(x)                                                    # model.py:211


def synthetic_batch(batch, n_dense, vocab, hots, device, seed=0):
    g = torch.Generator(device=device).manual_seed(seed)
    ids = {f"cat_{t:02d}_id": torch.randint(0, vocab, (batch, h), device=device, generator=g, dtype=torch.int32)
           for t, h in enumerate(hots)}
    dense = torch.rand(batch, n_dense, device=device, generator=g)
    labels = (torch.rand(batch, 1, device=device, generator=g) < 0.3).float()
    return {"dense_input": dense, "large_emb_inputs": ids}, labels


def build_model(batch, vocab, hots, embedding_dim=128, projection=512, cross_layers=3, table_optimizer=None,
                bottom=(512, 256, 128), top=(1024, 1024, 512, 256, 1), embedding_threshold=0, dtype="mixed_bfloat16",
                embedding_dtype="bfloat16"):
    """vocab: one size for every table or a list (the Criteo vocabularies of configs/v6e_8.py); tables smaller
    than `embedding_threshold` become small (plain, trainable) embeddings as in main.py:135-141."""
    opt = table_optimizer or kl.Adagrad(learning_rate=0.0034, initial_accumulator_value=0.1)  # configs/v6e_8.py
    vocabs = list(vocab) if isinstance(vocab, (list, tuple)) else [vocab] * len(hots)
    feats, small = {}, []
    for t, h in enumerate(hots):
        if vocabs[t] < embedding_threshold:
            small.append({"name": f"cat_{t}", "new_name": f"cat_{t:02d}", "vocabulary_size": vocabs[t]})
            continue
        tc = kl.TableConfig(name=f"cat_{t}", vocabulary_size=vocabs[t], embedding_dim=embedding_dim,
                            initializer=base.RandomUniform(-0.05, 0.05, seed=1337 + t), optimizer=opt,
                            combiner="sum", placement="sparsecore")
        feats[f"cat_{t:02d}_id"] = kl.FeatureConfig(f"cat_{t}", tc, (batch, h), (batch, embedding_dim))
    bottom = tuple(bottom[:-1]) + (embedding_dim,)
    return DLRMDCNV2(feats, embedding_dim, list(bottom), list(top), cross_layers, projection, dtype=dtype,
                     embedding_dtype=embedding_dtype, small_emb_features=small or None)


def train_step(model, opt_box, inputs, labels):
    """One step: forward, BCE, backward (table optimizers run inside it), dense optimizer step."""
    pred = model(inputs)
    loss = kl.binary_crossentropy(labels, pred)     # main.py:201-210, forward + backward in one pass (krs_bce_fwd_bwd)
    loss.backward()
    if opt_box[0] is None:  # the first step has built the layers
        dense_params = [p for p in model.parameters() if p.requires_grad]
        from keras_rs_amd.optim import Adagrad

        opt_box[0] = Adagrad(dense_params, lr=0.0034, initial_accumulator_value=0.1, prepare_casts=True)
    opt_box[0].step()
    opt_box[0].zero_grad(set_to_none=True)
    return loss.detach()


if __name__ == "__main__":
    dev = torch.device("cuda", 0)
    hots = [3, 2, 1, 2, 6, 1, 1, 1, 1, 7, 3, 8, 1, 6, 9, 5, 1, 1, 1, 12, 100, 27, 10, 3, 1, 1]
    model = build_model(8192, 100_000, hots)
    box = [None]
    for step in range(5):
        x, y = synthetic_batch(8192, 13, 100_000, hots, dev, seed=step)
        print(f"step {step}: loss {float(train_step(model, box, x, y)):.4f}")
