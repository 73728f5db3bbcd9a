"""FeatureCross on MI355X: drop-in for keras_rs.layers.FeatureCross.

Same constructor, call signature, weight order, config keys and errors as
keras_rs/src/layers/feature_interaction/feature_cross.py:93-222; the arithmetic
runs in krs_gemm (MFMA, fused cross epilogue) via autograd.CrossLayerFn.
"""

from __future__ import annotations

from typing import Any

import torch

from keras_rs_amd import _lib as L
from keras_rs_amd.autograd import CrossEpilogueFn, CrossLayerFn, Dx0Relay
from keras_rs_amd.layers import base

_FUSED_ACTS = {None: L.ACT_NONE, base.linear: L.ACT_NONE, base.relu: L.ACT_RELU,
               base.sigmoid: L.ACT_SIGMOID, base.tanh: L.ACT_TANH}


class FeatureCross(base.Layer):
    """x_{i+1} = x0 * (W x_i + bias + diag_scale * x_i) + x_i  (DCN-v2 cross layer).

    Args mirror the reference (feature_cross.py:93-108): projection_dim, diag_scale,
    use_bias, pre_activation, kernel_initializer, bias_initializer,
    kernel_regularizer, bias_regularizer, plus base kwargs (dtype, name).
    """

    def __init__(self, projection_dim: int | None = None, diag_scale: float | None = 0.0,
                 use_bias: bool = True, pre_activation=None, kernel_initializer="glorot_uniform",
                 bias_initializer="zeros", kernel_regularizer=None, bias_regularizer=None, **kwargs: Any):
        super().__init__(**kwargs)
        self.projection_dim = projection_dim
        self.diag_scale = diag_scale
        self.use_bias = use_bias
        self.pre_activation = base.get_activation(pre_activation)
        self.kernel_initializer = base.get_initializer(kernel_initializer)
        self.bias_initializer = base.get_initializer(bias_initializer)
        # applied as in the reference (its Dense sublayers, feature_cross.py:134-151): the penalties appear in
        # `layer.losses` (kernel_regularizer on BOTH kernels, bias_regularizer on the bias)
        self.kernel_regularizer = base.get_regularizer(kernel_regularizer)
        self.bias_regularizer = base.get_regularizer(bias_regularizer)
        self.supports_masking = True
        if self.diag_scale is not None and self.diag_scale < 0.0:  # feature_cross.py:124-128
            raise ValueError(f"`diag_scale` should be non-negative. Received: `diag_scale={self.diag_scale}`")
        for w in ("down_kernel", "kernel", "bias"):
            self.register_parameter(w, None)

    def build(self, input_shape, *_) -> None:
        d = int(input_shape[-1])
        # weight order of the reference (feature_cross_test.py:44-47):
        # [down_proj.kernel (d,p)], dense.kernel (p|d, d), dense.bias (d)
        if self.projection_dim is not None:
            self.down_kernel = self.add_weight((d, self.projection_dim),
                                               base.clone_initializer(self.kernel_initializer), "down_kernel",
                                               regularizer=self.kernel_regularizer)
        k_in = d if self.projection_dim is None else self.projection_dim
        self.kernel = self.add_weight((k_in, d), base.clone_initializer(self.kernel_initializer), "kernel",
                                      regularizer=self.kernel_regularizer)
        if self.use_bias:
            self.bias = self.add_weight((d,), base.clone_initializer(self.bias_initializer), "bias",
                                        dtype=torch.float32, regularizer=self.bias_regularizer)
        self.built = True

    def call(self, x0: torch.Tensor, x: torch.Tensor | None = None) -> torch.Tensor:
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
        act = self.pre_activation
        if act in _FUSED_ACTS:
            # dL/dx0 travels down a stack of layers on the same x0 through relays instead of autograd adds
            relay = Dx0Relay(x02)
            y = CrossLayerFn.apply(x02, x2, self.down_kernel, self.kernel, self.bias, diag, _FUSED_ACTS[act],
                                   self.compute_dtype, relay, None if same else getattr(x, "_krs_dx0_relay", None))
            out = y.reshape(*lead, d)
            out._krs_dx0_relay = relay
            return out
        else:
            # arbitrary callable: GEMM(+bias) on MFMA, the callable on the host framework, cross kernel fused
            cd = self.compute_dtype
            h = x2.to(cd)
            if self.down_kernel is not None:
                h = _Linear.apply(h, self.down_kernel, None, cd)
            z = _Linear.apply(h, self.kernel, self.bias, cd)
            u = act(z).to(cd)
            y = CrossEpilogueFn.apply(u, x02.to(cd), x2.to(cd), diag)
        return y.reshape(*lead, d)

    def get_config(self) -> dict:
        config = super().get_config()
        config.update({
            "projection_dim": self.projection_dim,
            "diag_scale": self.diag_scale,
            "use_bias": self.use_bias,
            "pre_activation": base.serialize_activation(self.pre_activation),
            "kernel_initializer": self.kernel_initializer.serialize(),
            "bias_initializer": self.bias_initializer.serialize(),
            "kernel_regularizer": base.serialize_regularizer(self.kernel_regularizer),
            "bias_regularizer": base.serialize_regularizer(self.bias_regularizer),
        })
        return config

    @classmethod
    def from_config(cls, config: dict):
        config = dict(config)
        if config.get("pre_activation") == "linear":
            config["pre_activation"] = None
        return cls(**config)


class _Linear(torch.autograd.Function):
    """z = h @ K (+ b) on krs_gemm, for the host-composed activation path."""

    @staticmethod
    def forward(ctx, h, kernel, bias, cd):
        from keras_rs_amd import dense_ops as D

        hc, kc = h.to(cd).contiguous(), kernel.to(cd)
        ctx.save_for_backward(hc, kc)
        ctx.meta = (kernel.dtype, bias is not None, h.dtype)
        z, _ = D.gemm(hc, kc, bias=bias)
        return z

    @staticmethod
    def backward(ctx, g):
        from keras_rs_amd import dense_ops as D

        hc, kc = ctx.saved_tensors
        k_dt, has_bias, h_dt = ctx.meta
        g = g.to(hc.dtype).contiguous()
        dk, _ = D.gemm(hc, g, a_is_km=True, out_dtype=torch.float32)
        dh, _ = D.gemm(g, kc, b_is_nk=True)
        return dh.to(h_dt), dk.to(k_dt), D.colsum(g) if has_bias else None, None
