"""Keras 3 side of the binding: `keras.layers.Layer` subclasses of FeatureCross / DotInteraction / DistributedEmbedding
whose `call` runs the HIP kernels (keras_rs_amd.autograd) -- the code INTEGRATION.md section 2-3 sketches, as a module.

Requires the Keras **torch** backend on PyTorch-ROCm (KERAS_BACKEND=torch): there a `keras.Variable`'s `.value` is a
`torch.nn.Parameter` on the device and `call` receives `torch.Tensor`s, so the `torch.autograd.Function`s of this
package are the custom gradients (what `keras.ops.custom_gradient` would wrap on other backends).  Keras is not
installed in the build image; the module therefore takes the keras module as an argument:

    import keras, keras_rs_amd.keras_adapter as ka
    L = ka.make_layers(keras)                 # or ka.layers() which imports keras itself
    cross = L.FeatureCross(projection_dim=512)          # drop-in for keras_rs.layers.FeatureCross
    keras_rs.layers.FeatureCross = L.FeatureCross       # examples/dcn.py, examples/ml_perf/model.py unchanged

and tests/test_keras_adapter.py exercises it against a minimal stand-in for the three Keras symbols it touches
(`layers.Layer` with add_weight / build-on-first-call, `initializers.get`, `activations.get`).  Constructor arguments,
`layer.weights` order (feature_cross_test.py:21-47: [down_proj kernel], dense kernel, dense bias), config keys and
error behaviour follow keras_rs/src/layers/feature_interaction/feature_cross.py:93-222, dot_interaction.py:84-234 and
embedding/base_distributed_embedding.py:468-808.
"""

from __future__ import annotations

import types
from typing import Any


def _tensor(v):
    """keras.Variable -> backend tensor (torch backend: the Parameter itself); tensors pass through."""
    return getattr(v, "value", v)


def make_layers(keras) -> types.SimpleNamespace:
    import torch

    from keras_rs_amd import _lib as L
    from keras_rs_amd.autograd import CrossEpilogueFn, CrossLayerFn, DotInteractionFn, Dx0Relay
    from keras_rs_amd.layers import distributed_embedding as de_torch
    from keras_rs_amd.layers.feature_cross import _Linear

    fused_acts = {None: L.ACT_NONE, "linear": L.ACT_NONE, "relu": L.ACT_RELU, "sigmoid": L.ACT_SIGMOID,
                  "tanh": L.ACT_TANH}

    def compute_dtype(layer):
        name = getattr(getattr(layer, "dtype_policy", None), "compute_dtype", None) or "float32"
        return {"float32": torch.float32, "bfloat16": torch.bfloat16}[str(name)]

    class FeatureCross(keras.layers.Layer):
        """keras_rs.layers.FeatureCross on libkrs_hip.so (feature_cross.py:93-222)."""

        def __init__(self, projection_dim=None, diag_scale=0.0, use_bias=True, pre_activation=None,
                     kernel_initializer="glorot_uniform", bias_initializer="zeros", kernel_regularizer=None,
                     bias_regularizer=None, **kwargs: Any):
            super().__init__(**kwargs)
            self.projection_dim = projection_dim
            self.diag_scale = diag_scale
            self.use_bias = use_bias
            self._pre_activation_id = pre_activation
            self.pre_activation = keras.activations.get(pre_activation)
            self.kernel_initializer = keras.initializers.get(kernel_initializer)
            self.bias_initializer = keras.initializers.get(bias_initializer)
            self.kernel_regularizer = kernel_regularizer
            self.bias_regularizer = bias_regularizer
            self.supports_masking = True
            if self.diag_scale is not None and self.diag_scale < 0.0:  # feature_cross.py:124-128
                raise ValueError(f"`diag_scale` should be non-negative. Received: `diag_scale={self.diag_scale}`")
            self.down_kernel = self.kernel = self.bias = None

        def build(self, input_shape):
            d = int(input_shape[-1])
            clone = lambda i: i.clone() if hasattr(i, "clone") else i  # noqa: E731  (feature_cross.py:99-102)
            # the reference's sublayers: down_proj = Dense(p, use_bias=False), dense = Dense(d): same weight order
            if self.projection_dim is not None:
                self.down_kernel = self.add_weight(shape=(d, self.projection_dim), initializer=clone(self.kernel_initializer),
                                                   regularizer=self.kernel_regularizer, name="down_proj_kernel")
            k_in = d if self.projection_dim is None else self.projection_dim
            self.kernel = self.add_weight(shape=(k_in, d), initializer=clone(self.kernel_initializer),
                                          regularizer=self.kernel_regularizer, name="dense_kernel")
            if self.use_bias:
                self.bias = self.add_weight(shape=(d,), initializer=clone(self.bias_initializer),
                                            regularizer=self.bias_regularizer, name="dense_bias")
            self.built = True

        def call(self, x0, x=None):
            if x is None:
                x = x0
            if tuple(x0.shape) != tuple(x.shape):  # feature_cross.py:175-179
                raise ValueError("`x0` and `x` should have the same shape. Received: "
                                 f"`x.shape` = {tuple(x.shape)}, `x0.shape` = {tuple(x0.shape)}")
            L.require_device(x0, "FeatureCross input")
            lead, d = x0.shape[:-1], x0.shape[-1]
            same = x is x0
            x02 = x0.reshape(-1, d)
            x2 = x02 if same else x.reshape(-1, d)
            diag = float(self.diag_scale) if self.diag_scale else 0.0
            cd = compute_dtype(self)
            down = None if self.down_kernel is None else _tensor(self.down_kernel)
            kernel = _tensor(self.kernel)
            bias = None if self.bias is None else _tensor(self.bias).float()
            act_id = self._pre_activation_id
            if act_id is None or (isinstance(act_id, str) and act_id in fused_acts):
                relay = Dx0Relay(x02)
                y = CrossLayerFn.apply(x02, x2, down, kernel, bias, diag, fused_acts[act_id], cd, relay,
                                       None if same else getattr(x, "_krs_dx0_relay", None))
                out = y.reshape(*lead, d)
                out._krs_dx0_relay = relay
                return out
            h = x2.to(cd)                       # arbitrary callable: GEMMs on MFMA, the callable in between
            if down is not None:
                h = _Linear.apply(h, down, None, cd)
            u = self.pre_activation(_Linear.apply(h, kernel, bias, cd)).to(cd)
            return CrossEpilogueFn.apply(u, x02.to(cd), x2.to(cd), diag).reshape(*lead, d)

        def compute_output_shape(self, x0_shape, x_shape=None):
            return x0_shape

        def get_config(self):
            config = super().get_config()
            config.update({
                "projection_dim": self.projection_dim, "diag_scale": self.diag_scale, "use_bias": self.use_bias,
                "pre_activation": keras.activations.serialize(self.pre_activation),
                "kernel_initializer": keras.initializers.serialize(self.kernel_initializer),
                "bias_initializer": keras.initializers.serialize(self.bias_initializer),
                "kernel_regularizer": self.kernel_regularizer, "bias_regularizer": self.bias_regularizer,
            })
            return config

    class DotInteraction(keras.layers.Layer):
        """keras_rs.layers.DotInteraction on libkrs_hip.so (dot_interaction.py:84-234)."""

        def __init__(self, self_interaction=False, skip_gather=False, **kwargs: Any):
            super().__init__(**kwargs)
            self.self_interaction = self_interaction
            self.skip_gather = skip_gather

        def build(self, input_shape):
            self.built = True

        def call(self, inputs):
            shape = tuple(inputs[0].shape)
            for idx, t in enumerate(inputs):
                if len(shape) != 2:  # dot_interaction.py:156-160
                    raise ValueError("All feature tensors inside `inputs` should have rank 2. "
                                     f"Received rank {len(shape)} at index {idx}.")
                if tuple(t.shape) != shape:  # :162-167
                    raise ValueError("All feature tensors in `inputs` should have the same shape. Found at least one "
                                     f"conflict: shape = {shape} at index 0 and shape = {tuple(t.shape)} at index {idx}.")
            cd = compute_dtype(self)
            return DotInteractionFn.apply(self.self_interaction, self.skip_gather, None, 0,
                                          *[t if t.dtype == cd else t.to(cd) for t in inputs])

        def compute_output_shape(self, input_shape):
            n, batch = len(input_shape), input_shape[0][0]
            if self.skip_gather:
                return (batch, n * n)
            return (batch, n * (n + 1) // 2 if self.self_interaction else n * (n - 1) // 2)

        def get_config(self):
            config = super().get_config()
            config.update({"self_interaction": self.self_interaction, "skip_gather": self.skip_gather})
            return config

    class DistributedEmbedding(keras.layers.Layer):
        """keras_rs.layers.DistributedEmbedding (base_distributed_embedding.py:468-808) whose placements are served
        by keras_rs_amd.layers.DistributedEmbedding: the hook set `_sparsecore_{init,build,preprocess,call,
        get_embedding_tables}` (base:990-1042) is that class's; this wrapper gives it the Keras layer protocol.
        'default_device' tables are exposed as trainable Keras weights (the model optimizer updates them, as in the
        reference), 'sparsecore' tables as non-trainable ones (updated inside the backward,
        jax/embedding_lookup.py:174-273); the optimizer slots and step counts of the fused optimizers are written to /
        read from Keras checkpoints by save_own_variables / load_own_variables below."""

        def __init__(self, feature_configs, *, table_stacking="auto", update_stats=False, **kwargs: Any):
            super().__init__(**kwargs)
            pol = getattr(self, "dtype_policy", None)
            self._impl = de_torch.DistributedEmbedding(feature_configs, table_stacking=table_stacking,
                                                       update_stats=update_stats,
                                                       dtype=getattr(pol, "name", None) or "float32")
            self._feature_configs = feature_configs

        @classmethod
        def has_sparsecores(cls):
            return de_torch.DistributedEmbedding.has_sparsecores()

        def build(self, input_shapes=None):
            self._impl.build(input_shapes)
            track = getattr(self, "_track_variable", None)
            if track is not None:       # Keras 3: torch Parameters wrapped as keras Variables
                for name, p in self._impl.named_parameters():
                    track(keras.Variable(p, trainable=bool(p.requires_grad), name=name))
            self.built = True

        # Keras checkpoints (.keras / .weights.h5, model.save_weights) call these two per layer.  The tables are
        # tracked variables, but the fused optimizers' slot planes are module buffers and their step counts live in
        # the module's extra state: written through the default path alone, a restore would silently reset the
        # Adagrad / Adam / FTRL accumulators and Adam's bias-correction step.  So the layer writes its WHOLE state
        # itself: every entry of the torch state_dict (tables or stacks, slot planes; bf16 as float32, which holds
        # it exactly) plus the per-group iteration counts.  Counterpart of the slot variables the reference adds to
        # the layer (jax/distributed_embedding.py:316-345) so that they are checkpointed with it.
        # The store is used through the surface saving_lib's H5Entry / NpzIOStore entries offer and no further: item
        # assignment, item access read with `[...]`, `keys()` -- no `in`, no nested keys (a "/" would ask the store for a
        # group of its own): the step counts travel as "iterations__<group>".
        def save_own_variables(self, store):
            import numpy as np

            sd = self._impl.state_dict()
            extra = sd.pop("_extra_state", None)
            for k, v in sd.items():
                store[k.replace("/", "__")] = (v.float() if v.dtype == torch.bfloat16 else v).detach().cpu().numpy()
            its = (extra or {}).get("iterations", {})
            for k, n in its.items():
                store["iterations__" + str(k).replace("/", "__")] = np.asarray(int(n), np.int64)

        def load_own_variables(self, store):
            if not self.built:
                self.build(None)
            sd = self._impl.state_dict()
            extra = sd.pop("_extra_state", None)
            have = set(store.keys())

            def stored(key, legacy):
                """The store's name for `key`: the "__" spelling, or the one an earlier revision wrote ("/" kept, step counts
                as "iterations/<group>") -- checkpoints of both revisions load."""
                name = key.replace("/", "__")
                return name if name in have else (legacy if legacy in have else None)

            names = {k: stored(k, k) for k in sd}
            missing = [k for k, n in names.items() if n is None]
            if missing:
                raise ValueError(f"DistributedEmbedding.load_own_variables: the checkpoint lacks {missing}")
            new = {k: torch.as_tensor(store[names[k]][...]).to(v.dtype) for k, v in sd.items()}
            its, absent = {}, []
            for k in (extra or {}).get("iterations", {}):
                n = stored("iterations__" + str(k), "iterations/" + str(k))
                if n is None:
                    absent.append(k)
                else:
                    its[k] = int(store[n][...])
            if absent:
                # the update counts drive Adam's bias correction and every learning-rate schedule: restoring the slot planes
                # with a count of 0 would apply step-1 constants to trained moments
                raise ValueError(f"DistributedEmbedding.load_own_variables: the checkpoint lacks the optimizer iteration "
                                 f"count(s) of {absent}")
            if extra is not None:     # (a module without extra state rejects the key under strict loading)
                new["_extra_state"] = {"iterations": its}
            self._impl.load_state_dict(new)

        def preprocess(self, inputs, weights=None, training=False):
            return self._impl.preprocess(inputs, weights, training)

        def call(self, inputs, weights=None, training=False):
            return self._impl(inputs, weights, training)

        def get_embedding_tables(self):
            return self._impl.get_embedding_tables()

        def set_embedding_tables(self, tables):
            self._impl.set_embedding_tables(tables)

        def compute_output_shape(self, input_shapes):
            return self._impl.compute_output_shape(input_shapes)

        def get_config(self):
            config = super().get_config()
            inner = self._impl.get_config()
            config.update({k: inner[k] for k in ("feature_configs", "tables", "table_stacking")})
            return config

    return types.SimpleNamespace(FeatureCross=FeatureCross, DotInteraction=DotInteraction,
                                 DistributedEmbedding=DistributedEmbedding)


def layers() -> types.SimpleNamespace:
    """The adapter classes on the installed keras (torch backend)."""
    try:
        import keras
    except ImportError as e:  # pragma: no cover - keras is absent from the build image
        raise ImportError("keras_rs_amd.keras_adapter needs Keras 3 (KERAS_BACKEND=torch); the torch-native layers in "
                          "keras_rs_amd.layers have no such dependency") from e
    if keras.backend.backend() != "torch":
        raise RuntimeError("keras_rs_amd.keras_adapter runs on the Keras torch backend (KERAS_BACKEND=torch)")
    return make_layers(keras)
