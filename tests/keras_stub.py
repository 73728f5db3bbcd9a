"""Minimal stand-in for the Keras 3 symbols keras_rs_amd.keras_adapter touches (keras is not installed here):
layers.Layer (add_weight, build on first call, weights, dtype policy, get_config), initializers.get / serialize,
activations.get / serialize.  Semantics follow Keras 3's torch backend: a variable's `.value` is a torch
Parameter."""

import types

import numpy as np
import torch


class Variable:
    def __init__(self, value, trainable=True, name=None):
        self.value = value if isinstance(value, torch.nn.Parameter) else torch.nn.Parameter(value, requires_grad=trainable)
        self.trainable, self.name = trainable, name

    @property
    def shape(self):
        return tuple(self.value.shape)

    def numpy(self):
        return self.value.detach().cpu().numpy()


class _Init:
    def __init__(self, name):
        self.name = name

    def clone(self):
        return _Init(self.name)

    def __call__(self, shape, dtype=torch.float32, device="cpu"):
        if self.name == "zeros":
            return torch.zeros(shape, dtype=dtype, device=device)
        if self.name == "ones":
            return torch.ones(shape, dtype=dtype, device=device)
        fan_in, fan_out = (shape[0], shape[-1]) if len(shape) > 1 else (shape[0], shape[0])
        lim = float(np.sqrt(6.0 / (fan_in + fan_out)))
        return (torch.rand(shape, dtype=torch.float32, device=device) * 2 - 1).mul_(lim).to(dtype)


class _Policy:
    def __init__(self, name):
        self.name = name or "float32"
        self.compute_dtype = "bfloat16" if self.name in ("bfloat16", "mixed_bfloat16") else "float32"
        self.variable_dtype = "bfloat16" if self.name == "bfloat16" else "float32"


class Layer:
    DEVICE = "cpu"

    def __init__(self, dtype=None, name=None, trainable=True, **kwargs):
        if kwargs:
            raise TypeError(f"Unrecognized keyword arguments: {kwargs}")
        self.dtype_policy = _Policy(dtype)
        self.name = name or type(self).__name__.lower()
        self.trainable = trainable
        self.built = False
        self._weights = []

    def add_weight(self, shape=None, initializer="zeros", dtype=None, trainable=True, regularizer=None, name=None):
        init = initializers.get(initializer)
        v = Variable(init(tuple(shape), torch.float32, self.DEVICE), trainable=trainable, name=name)
        self._weights.append(v)
        return v

    def _track_variable(self, v):
        self._weights.append(v)

    @property
    def weights(self):
        return list(self._weights)

    def __call__(self, *args, **kwargs):
        if not self.built:
            first = args[0]
            if isinstance(first, dict):        # nested inputs (DistributedEmbedding): shapes are not inspected
                shape = None
            elif isinstance(first, (list, tuple)):
                shape = [tuple(t.shape) for t in first]
            else:
                shape = tuple(first.shape)
            self.build(shape)
            self.built = True
        return self.call(*args, **kwargs)

    def get_config(self):
        return {"name": self.name, "trainable": self.trainable, "dtype": self.dtype_policy.name}


def _get_init(x):
    return x if isinstance(x, _Init) else _Init(str(x))


_ACTS = {None: lambda t: t, "linear": lambda t: t, "relu": torch.relu, "sigmoid": torch.sigmoid, "tanh": torch.tanh}


def _get_act(x):
    return x if callable(x) else _ACTS[x]


def _ser_act(fn):
    for k, v in _ACTS.items():
        if v is fn and k is not None:
            return k
    return getattr(fn, "__name__", "custom")


layers = types.SimpleNamespace(Layer=Layer)
initializers = types.SimpleNamespace(get=_get_init, serialize=lambda i: i.name)
activations = types.SimpleNamespace(get=_get_act, serialize=_ser_act)
backend = types.SimpleNamespace(backend=lambda: "torch")
