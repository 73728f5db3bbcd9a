"""Dense on MI355X: the counterpart of the keras.layers.Dense blocks around the hot path
(bottom / top MLP of the DLRM-DCN model, examples/ml_perf/model.py:105-163, 214-262; SURVEY.md
section 8f.3).  y = activation(x @ kernel + bias) runs as ONE krs_gemm launch with the bias +
activation epilogue; relu / sigmoid / tanh / linear are fused, any other callable is applied on the
host framework after a linear launch.
"""

from __future__ import annotations

from typing import Any

import torch

from keras_rs_amd import _lib as L
from keras_rs_amd.autograd import DenseActRelay, DenseFn
from keras_rs_amd.layers import base

_FUSED_ACTS = {None: L.ACT_NONE, base.linear: L.ACT_NONE, base.relu: L.ACT_RELU,
               base.sigmoid: L.ACT_SIGMOID, base.tanh: L.ACT_TANH}


class Dense(base.Layer):
    """Args as keras.layers.Dense: units, activation, use_bias, kernel_initializer, bias_initializer
    (+ dtype / name).  Weights: kernel [input_dim, units], bias [units] (fp32)."""

    def __init__(self, units: int, activation=None, use_bias: bool = True, kernel_initializer="glorot_uniform",
                 bias_initializer="zeros", **kwargs: Any):
        super().__init__(**kwargs)
        if int(units) <= 0:
            raise ValueError(f"`units` must be a positive integer. Received: units={units}")
        self.units = int(units)
        self.activation = base.get_activation(activation)
        self.use_bias = use_bias
        self.kernel_initializer = base.get_initializer(kernel_initializer)
        self.bias_initializer = base.get_initializer(bias_initializer)
        for w in ("kernel", "bias"):
            self.register_parameter(w, None)

    def build(self, input_shape, *_) -> None:
        d = int(input_shape[-1])
        self.kernel = self.add_weight((d, self.units), base.clone_initializer(self.kernel_initializer), "kernel")
        if self.use_bias:
            self.bias = self.add_weight((self.units,), base.clone_initializer(self.bias_initializer), "bias",
                                        dtype=torch.float32)
        self.built = True

    def call(self, x: torch.Tensor) -> torch.Tensor:
        L.require_device(x, "Dense input")
        lead, d = x.shape[:-1], x.shape[-1]
        if d != self.kernel.shape[0]:
            raise ValueError(f"Dense: input feature size {d} does not match the kernel {tuple(self.kernel.shape)}")
        fused = self.activation in _FUSED_ACTS
        # (an input that is the output of a FeatureCross stack carries that layer's relay: the data gradient of this
        #  layer then runs the cross layer's elementwise backward in its epilogue, autograd.DenseFn)
        # (likewise an input that is the output of another Dense layer carries a DenseActRelay: that layer's activation
        #  backward and bias gradient then ride in this layer's data-gradient product; this layer's own output gets one)
        two_d = x.dim() == 2
        relay_out = DenseActRelay() if (two_d and fused) else None
        y = DenseFn.apply(x.reshape(-1, d), self.kernel, self.bias,
                          _FUSED_ACTS[self.activation] if fused else L.ACT_NONE, self.compute_dtype,
                          getattr(x, "_krs_dx0_relay", None) if two_d else None,
                          getattr(x, "_krs_dense_relay", None) if two_d else None, relay_out)
        if not fused:
            y = self.activation(y)
        y = y.reshape(*lead, self.units)
        if relay_out is not None:
            import weakref

            relay_out.out_ref = weakref.ref(y)
            y._krs_dense_relay = relay_out
        return y

    def compute_output_shape(self, input_shape):
        return tuple(input_shape[:-1]) + (self.units,)

    def get_config(self) -> dict:
        config = super().get_config()
        config.update({
            "units": self.units,
            "activation": base.serialize_activation(self.activation),
            "use_bias": self.use_bias,
            "kernel_initializer": self.kernel_initializer.serialize(),
            "bias_initializer": self.bias_initializer.serialize(),
        })
        return config
