"""Minimal Keras-style layer protocol on torch.nn.Module.

Keras is not installed where this runs (and must not be needed on the GPU box),
so the drop-in layers keep the *surface* of keras.layers.Layer that the
reference's hot path relies on -- constructor kwargs (`dtype`, `name`),
`__call__` -> lazy `build(input_shape)` -> `call(...)`, `.weights` in creation
order, `get_config()/from_config()`, `built`, `supports_masking` -- without
importing it.  INTEGRATION.md shows how the same classes register as real
keras layers when keras (torch backend) is present.
"""

from __future__ import annotations

import itertools
import math
from typing import Any, Callable

import torch

_name_counters: dict[str, itertools.count] = {}


def _auto_name(prefix: str) -> str:
    c = _name_counters.setdefault(prefix, itertools.count())
    i = next(c)
    return prefix if i == 0 else f"{prefix}_{i}"


def default_device() -> torch.device:
    return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")


# --------------------------------------------------------------------------- #
def stream_capturing() -> bool:
    """True while the current HIP stream records into a graph (torch.cuda.graph): host-side polling of events and of
    page-locked mirrors is then left out of the step -- a replay cannot run it -- and done by the caller between
    replays (`check_ids(wait=True)`, `poll_exchange_stats()`)."""
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


# dtype policy (keras.DTypePolicy subset: float32 | bfloat16 | mixed_bfloat16)
# --------------------------------------------------------------------------- #
class DTypePolicy:
    def __init__(self, name: str | None = None):
        name = name or "float32"
        if isinstance(name, DTypePolicy):
            name = name.name
        if isinstance(name, torch.dtype):
            name = {torch.float32: "float32", torch.bfloat16: "bfloat16"}[name]
        if name not in ("float32", "bfloat16", "mixed_bfloat16"):
            raise ValueError(f"Unsupported dtype policy '{name}' (float32, bfloat16, mixed_bfloat16)")
        self.name = name
        self.compute_dtype = torch.float32 if name == "float32" else torch.bfloat16
        self.variable_dtype = torch.bfloat16 if name == "bfloat16" else torch.float32


# --------------------------------------------------------------------------- #
# initializers (the ones the hot path's layers and examples use)
# --------------------------------------------------------------------------- #
class Initializer:
    def __call__(self, shape, dtype=torch.float32, device=None) -> torch.Tensor:
        raise NotImplementedError

    def get_config(self) -> dict:
        return {}

    def serialize(self) -> dict:
        return {"class_name": type(self).__name__, "config": self.get_config()}

    def clone(self) -> "Initializer":
        return type(self)(**self.get_config())


def _fans(shape):
    if len(shape) < 1:
        return 1, 1
    if len(shape) == 1:
        return shape[0], shape[0]
    return shape[-2], shape[-1]


class Zeros(Initializer):
    def __call__(self, shape, dtype=torch.float32, device=None):
        return torch.zeros(shape, dtype=dtype, device=device)


class Ones(Initializer):
    def __call__(self, shape, dtype=torch.float32, device=None):
        return torch.ones(shape, dtype=dtype, device=device)


class Constant(Initializer):
    def __init__(self, value=0.0):
        self.value = value

    def __call__(self, shape, dtype=torch.float32, device=None):
        return torch.full(shape, float(self.value), dtype=dtype, device=device)

    def get_config(self):
        return {"value": self.value}


class RandomUniform(Initializer):
    """`device_rng=True` (an extension) draws on the target device in the target dtype, for tables too
    large to stage through host memory (a 40 M x 128 table is 20 GB in fp32); the values then come from
    the device generator instead of the host one."""

    def __init__(self, minval=-0.05, maxval=0.05, seed=None, device_rng=False):
        self.minval, self.maxval, self.seed, self.device_rng = minval, maxval, seed, bool(device_rng)

    def __call__(self, shape, dtype=torch.float32, device=None):
        if self.device_rng and device is not None and torch.device(device).type != "cpu":
            g = None if self.seed is None else torch.Generator(device=device).manual_seed(int(self.seed))
            return torch.empty(tuple(shape), dtype=dtype, device=device).uniform_(self.minval, self.maxval, generator=g)
        g = None if self.seed is None else torch.Generator().manual_seed(int(self.seed))
        t = torch.rand(shape, generator=g, dtype=torch.float32) * (self.maxval - self.minval) + self.minval
        return t.to(dtype).to(device)

    def get_config(self):
        cfg = {"minval": self.minval, "maxval": self.maxval, "seed": self.seed}
        if self.device_rng:
            cfg["device_rng"] = True
        return cfg


class VarianceScaling(Initializer):
    """keras.initializers.VarianceScaling (default of TableConfig is mode="fan_out",
    distributed_embedding_config.py:54-56)."""

    def __init__(self, scale=1.0, mode="fan_in", distribution="truncated_normal", seed=None):
        self.scale, self.mode, self.distribution, self.seed = scale, mode, distribution, seed

    def __call__(self, shape, dtype=torch.float32, device=None):
        fan_in, fan_out = _fans(tuple(shape))
        n = {"fan_in": fan_in, "fan_out": fan_out, "fan_avg": (fan_in + fan_out) / 2.0}[self.mode]
        scale = self.scale / max(1.0, n)
        g = None if self.seed is None else torch.Generator().manual_seed(int(self.seed))
        if self.distribution == "uniform":
            lim = math.sqrt(3.0 * scale)
            t = (torch.rand(shape, generator=g) * 2.0 - 1.0) * lim
        elif self.distribution in ("truncated_normal", "normal"):
            std = math.sqrt(scale) / 0.87962566103423978
            t = torch.empty(shape)
            torch.nn.init.trunc_normal_(t, 0.0, std, -2 * std, 2 * std, generator=g)
        else:  # untruncated_normal
            t = torch.randn(shape, generator=g) * math.sqrt(scale)
        return t.to(dtype).to(device)

    def get_config(self):
        return {"scale": self.scale, "mode": self.mode, "distribution": self.distribution, "seed": self.seed}


class GlorotUniform(VarianceScaling):
    def __init__(self, seed=None):
        super().__init__(scale=1.0, mode="fan_avg", distribution="uniform", seed=seed)

    def get_config(self):
        return {"seed": self.seed}


class LecunNormal(VarianceScaling):
    def __init__(self, seed=None):
        super().__init__(scale=1.0, mode="fan_in", distribution="truncated_normal", seed=seed)

    def get_config(self):
        return {"seed": self.seed}


class CallableInitializer(Initializer):
    """Wraps a user callable `(shape, dtype) -> array-like`."""

    def __init__(self, fn: Callable):
        self.fn = fn

    def __call__(self, shape, dtype=torch.float32, device=None):
        return torch.as_tensor(self.fn(tuple(shape), dtype)).to(dtype).to(device)

    def get_config(self):
        return {"fn": getattr(self.fn, "__name__", repr(self.fn))}

    def clone(self):
        return CallableInitializer(self.fn)


_INITIALIZERS = {
    "zeros": Zeros, "ones": Ones, "uniform": RandomUniform, "random_uniform": RandomUniform,
    "glorot_uniform": GlorotUniform, "variance_scaling": VarianceScaling, "lecun_normal": LecunNormal,
    "Zeros": Zeros, "Ones": Ones, "RandomUniform": RandomUniform, "GlorotUniform": GlorotUniform,
    "VarianceScaling": VarianceScaling, "LecunNormal": LecunNormal, "Constant": Constant,
}


def get_initializer(identifier) -> Initializer:
    if isinstance(identifier, Initializer):
        return identifier
    if isinstance(identifier, str):
        if identifier not in _INITIALIZERS:
            raise ValueError(f"Unknown initializer '{identifier}'")
        return _INITIALIZERS[identifier]()
    if isinstance(identifier, dict):
        return _INITIALIZERS[identifier["class_name"]](**identifier.get("config", {}))
    if callable(identifier):
        return CallableInitializer(identifier)
    raise ValueError(f"Cannot interpret initializer {identifier!r}")


def clone_initializer(init: Initializer) -> Initializer:
    """keras_rs/src/utils/keras_utils.py:30-51: sublayers get their own initializer object."""
    return init.clone()


# --------------------------------------------------------------------------- #
# regularizers (keras.regularizers subset: L1, L2, L1L2, callables).  A regulariser on a weight is what it is in
# Keras: a penalty term the layer REPORTS in `layer.losses` (evaluated on the current weight at access time) and the
# training loop adds to its loss -- keras.layers.Dense(kernel_regularizer=...) inside the reference's FeatureCross
# (feature_cross.py:134-151) and keras.layers.Embedding(embeddings_regularizer=...) under EmbedReduce do exactly that.
# The penalties are O(weight) elementwise sums on small dense weights, evaluated with torch (not part of the hot path).
# --------------------------------------------------------------------------- #
class Regularizer:
    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError

    def get_config(self) -> dict:
        return {}

    def serialize(self) -> dict:
        return {"class_name": type(self).__name__, "config": self.get_config()}


class L1L2(Regularizer):
    """keras.regularizers.L1L2: l1 * sum(|x|) + l2 * sum(x^2), in fp32."""

    def __init__(self, l1: float = 0.0, l2: float = 0.0):
        for name, v in (("l1", l1), ("l2", l2)):
            if v is None or not math.isfinite(float(v)) or float(v) < 0:
                raise ValueError(f"Invalid value for argument {name}: expected a non-negative finite float. Received: {name}={v}")
        self.l1, self.l2 = float(l1), float(l2)

    def __call__(self, x):
        xf = x.float()
        out = xf.new_zeros(())
        if self.l1:
            out = out + self.l1 * xf.abs().sum()
        if self.l2:
            out = out + self.l2 * xf.square().sum()
        return out

    def get_config(self):
        return {"l1": self.l1, "l2": self.l2}


class L1(L1L2):
    def __init__(self, l1: float = 0.01):
        super().__init__(l1=l1, l2=0.0)

    def get_config(self):
        return {"l1": self.l1}


class L2(L1L2):
    def __init__(self, l2: float = 0.01):
        super().__init__(l1=0.0, l2=l2)

    def get_config(self):
        return {"l2": self.l2}


class CallableRegularizer(Regularizer):
    def __init__(self, fn: Callable):
        self.fn = fn

    def __call__(self, x):
        return self.fn(x)

    def get_config(self):
        return {"fn": getattr(self.fn, "__name__", repr(self.fn))}


_REGULARIZERS = {"l1": lambda: L1(), "l2": lambda: L2(), "l1_l2": lambda: L1L2(l1=0.01, l2=0.01),
                 "L1": L1, "L2": L2, "L1L2": L1L2}


def get_regularizer(identifier):
    """None | Regularizer | "l1" / "l2" / "l1_l2" | serialized dict | callable  (keras.regularizers.get)."""
    if identifier is None or isinstance(identifier, Regularizer):
        return identifier
    if isinstance(identifier, str):
        if identifier not in _REGULARIZERS:
            raise ValueError(f"Unknown regularizer '{identifier}'")
        return _REGULARIZERS[identifier]()
    if isinstance(identifier, dict):
        return _REGULARIZERS[identifier["class_name"]](**identifier.get("config", {}))
    if callable(identifier):
        return CallableRegularizer(identifier)
    raise ValueError(f"Cannot interpret regularizer {identifier!r}")


def serialize_regularizer(reg):
    return None if reg is None else reg.serialize()


# --------------------------------------------------------------------------- #
# weight constraints (keras.constraints): projections applied to a variable AFTER each optimizer update.  Keras optimizers
# do that themselves for variables created with `constraint=`; here `keras_rs_amd.optim.Adagrad.step()` does it for the
# parameters it updates, and `Layer.apply_constraints()` is the explicit call for training loops on another optimizer.
# (Tables on the fused `sparsecore` placement are updated inside K2 and take no constraint -- the reference's SparseCore
# path has none either; EmbedReduce's table is an ordinary weight.)
# --------------------------------------------------------------------------- #
class Constraint:
    def __call__(self, w: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError

    def get_config(self) -> dict:
        return {}

    def serialize(self) -> dict:
        return {"class_name": type(self).__name__, "config": self.get_config()}


def _axis_norm(w, axis):
    return w.float().square().sum(dim=axis, keepdim=True).sqrt()


class MaxNorm(Constraint):
    """keras.constraints.MaxNorm: w * clip(norm, 0, max_value) / (eps + norm) along `axis`."""

    def __init__(self, max_value: float = 2.0, axis=0):
        self.max_value, self.axis = float(max_value), axis

    def __call__(self, w):
        n = _axis_norm(w, self.axis)
        return (w.float() * (n.clamp(0.0, self.max_value) / (1e-7 + n))).to(w.dtype)

    def get_config(self):
        return {"max_value": self.max_value, "axis": self.axis}


class NonNeg(Constraint):
    def __call__(self, w):
        return w * (w >= 0).to(w.dtype)


class UnitNorm(Constraint):
    def __init__(self, axis=0):
        self.axis = axis

    def __call__(self, w):
        return (w.float() / (1e-7 + _axis_norm(w, self.axis))).to(w.dtype)

    def get_config(self):
        return {"axis": self.axis}


class MinMaxNorm(Constraint):
    def __init__(self, min_value: float = 0.0, max_value: float = 1.0, rate: float = 1.0, axis=0):
        self.min_value, self.max_value, self.rate, self.axis = float(min_value), float(max_value), float(rate), axis

    def __call__(self, w):
        n = _axis_norm(w, self.axis)
        desired = self.rate * n.clamp(self.min_value, self.max_value) + (1.0 - self.rate) * n
        return (w.float() * (desired / (1e-7 + n))).to(w.dtype)

    def get_config(self):
        return {"min_value": self.min_value, "max_value": self.max_value, "rate": self.rate, "axis": self.axis}


class CallableConstraint(Constraint):
    def __init__(self, fn: Callable):
        self.fn = fn

    def __call__(self, w):
        return self.fn(w)

    def get_config(self):
        return {"fn": getattr(self.fn, "__name__", repr(self.fn))}


_CONSTRAINTS = {"max_norm": MaxNorm, "non_neg": NonNeg, "unit_norm": UnitNorm, "min_max_norm": MinMaxNorm,
                "MaxNorm": MaxNorm, "NonNeg": NonNeg, "UnitNorm": UnitNorm, "MinMaxNorm": MinMaxNorm}


def get_constraint(identifier):
    """None | Constraint | "max_norm" / "non_neg" / "unit_norm" / "min_max_norm" | serialized dict | callable
    (keras.constraints.get)."""
    if identifier is None or isinstance(identifier, Constraint):
        return identifier
    if isinstance(identifier, str):
        if identifier not in _CONSTRAINTS:
            raise ValueError(f"Unknown constraint '{identifier}'")
        return _CONSTRAINTS[identifier]()
    if isinstance(identifier, dict):
        if identifier.get("class_name") not in _CONSTRAINTS:
            raise ValueError(f"Unknown constraint {identifier!r}")
        return _CONSTRAINTS[identifier["class_name"]](**identifier.get("config", {}))
    if callable(identifier):
        return CallableConstraint(identifier)
    raise ValueError(f"Cannot interpret constraint {identifier!r}")


def serialize_constraint(c):
    return None if c is None else c.serialize()


def apply_constraint(p: torch.Tensor) -> bool:
    """Projects a parameter through the constraint attached by `Layer.add_weight(constraint=...)`, in place (the
    version counter moves, so cached casts of the weight are rebuilt).  True when there was one."""
    c = getattr(p, "_krs_constraint", None)
    if c is None:
        return False
    with torch.no_grad():
        p.copy_(c(p.detach()))
    return True


# --------------------------------------------------------------------------- #
# activations fused in the GEMM epilogue
# --------------------------------------------------------------------------- #
def relu(x):
    return torch.relu(x)


def sigmoid(x):
    return torch.sigmoid(x)


def tanh(x):
    return torch.tanh(x)


def linear(x):
    return x


_ACTIVATIONS = {None: None, "linear": linear, "relu": relu, "sigmoid": sigmoid, "tanh": tanh}


def get_activation(identifier):
    if identifier is None or callable(identifier):
        return identifier
    if identifier not in _ACTIVATIONS:
        raise ValueError(f"Unknown activation '{identifier}'")
    return _ACTIVATIONS[identifier]


def serialize_activation(fn):
    if fn is None:
        return "linear"
    return getattr(fn, "__name__", repr(fn))


# --------------------------------------------------------------------------- #
# nested-structure helpers (the keras.tree subset DistributedEmbedding needs)
# --------------------------------------------------------------------------- #
def _is_nest(x) -> bool:
    return isinstance(x, (dict, list, tuple)) and not hasattr(x, "_fields_krs_leaf")


def flatten_with_path(structure, is_leaf=None, _path=()):
    """[(path tuple, leaf)], dict keys in sorted order (keras.tree / optree convention)."""
    if is_leaf is not None and is_leaf(structure):
        return [(_path, structure)]
    if isinstance(structure, dict):
        out = []
        for k in sorted(structure.keys(), key=str):
            out += flatten_with_path(structure[k], is_leaf, _path + (k,))
        return out
    if isinstance(structure, (list, tuple)):
        out = []
        for i, v in enumerate(structure):
            out += flatten_with_path(v, is_leaf, _path + (i,))
        return out
    return [(_path, structure)]


def flatten(structure, is_leaf=None):
    return [v for _, v in flatten_with_path(structure, is_leaf)]


# (The walkers below are module-level functions with explicit arguments, NOT recursive closures: a nested `def rec`
#  that calls itself sits in its own closure cell, a reference cycle that keeps everything the closure captured -- the
#  iterator over a step's output tensors, and through their `_krs_slab` tags the whole lookup slab -- alive until the
#  cyclic collector runs.  With the collector switched off (bench.py's timed region) that leaked the slab of every step.)
def _pack_rec(s, it, is_leaf):
    if is_leaf is not None and is_leaf(s):
        return next(it)
    if isinstance(s, dict):
        vals = {k: _pack_rec(s[k], it, is_leaf) for k in sorted(s.keys(), key=str)}
        return {k: vals[k] for k in s.keys()}  # keep the caller's insertion order
    if isinstance(s, tuple):
        return tuple(_pack_rec(v, it, is_leaf) for v in s)
    if isinstance(s, list):
        return [_pack_rec(v, it, is_leaf) for v in s]
    return next(it)


def pack_sequence_as(structure, flat, is_leaf=None):
    return _pack_rec(structure, iter(flat), is_leaf)


def _map_rec(s, others, fn, is_leaf):
    if (is_leaf is not None and is_leaf(s)) or not isinstance(s, (dict, list, tuple)):
        return fn(*others)
    if isinstance(s, dict):
        for o in others:
            if not isinstance(o, dict) or set(o.keys()) != set(s.keys()):
                raise ValueError("The two structures don't have the same nested structure.")
        return {k: _map_rec(s[k], [o[k] for o in others], fn, is_leaf) for k in s.keys()}
    for o in others:
        if not isinstance(o, (list, tuple)) or len(o) != len(s):
            raise ValueError("The two structures don't have the same nested structure.")
    out = [_map_rec(v, [o[i] for o in others], fn, is_leaf) for i, v in enumerate(s)]
    return tuple(out) if isinstance(s, tuple) else out


def map_structure_up_to(shallow, fn, *structures, is_leaf=None):
    """Walks `shallow`; at each of its leaves calls fn(*matching sub-structures of `structures`)
    (keras.tree.map_structure_up_to)."""
    return _map_rec(shallow, list(structures), fn, is_leaf)


def _assert_rec(x, y, is_leaf):
    if (is_leaf is not None and is_leaf(x)) or not isinstance(x, (dict, list, tuple)):
        if isinstance(y, (dict, list, tuple)) and not (is_leaf is not None and is_leaf(y)):
            # a leaf on one side may face an array-like on the other; nests must match nests
            if isinstance(y, dict) or (isinstance(y, (list, tuple)) and any(isinstance(e, (dict, list, tuple)) for e in y)):
                raise ValueError("The two structures don't have the same nested structure.")
        return
    if isinstance(x, dict):
        if not isinstance(y, dict) or set(x.keys()) != set(y.keys()):
            raise ValueError("The two structures don't have the same nested structure.")
        for k in x:
            _assert_rec(x[k], y[k], is_leaf)
    else:
        if not isinstance(y, (list, tuple)) or len(x) != len(y):
            raise ValueError("The two structures don't have the same nested structure.")
        for u, v in zip(x, y):
            _assert_rec(u, v, is_leaf)


def assert_same_structure(a, b, is_leaf=None):
    _assert_rec(a, b, is_leaf)


# --------------------------------------------------------------------------- #
# Layer
# --------------------------------------------------------------------------- #
class Layer(torch.nn.Module):
    """keras.layers.Layer look-alike (see module docstring)."""

    def __init__(self, *, dtype=None, name: str | None = None, trainable: bool = True, device=None,
                 **kwargs: Any):
        super().__init__()
        if kwargs:
            raise TypeError(f"Unrecognized keyword arguments passed to {type(self).__name__}: {kwargs}")
        self.dtype_policy = dtype if isinstance(dtype, DTypePolicy) else DTypePolicy(dtype)
        self.name = name or _auto_name(_snake(type(self).__name__))
        self.trainable = trainable
        self.built = False
        self.supports_masking = False
        self._device = torch.device(device) if device is not None else default_device()
        self._weight_order: list[torch.nn.Parameter] = []
        self._regularized: list[tuple[torch.nn.Parameter, Regularizer]] = []

    # keras spelling
    @property
    def compute_dtype(self):
        return self.dtype_policy.compute_dtype

    @property
    def variable_dtype(self):
        return self.dtype_policy.variable_dtype

    @property
    def dtype(self):
        return self.dtype_policy.variable_dtype

    def add_weight(self, shape, initializer, name: str, dtype=None, trainable=True, regularizer=None,
                   constraint=None) -> torch.nn.Parameter:
        init = get_initializer(initializer)
        value = init(tuple(shape), dtype or self.variable_dtype, self._device)
        p = torch.nn.Parameter(value, requires_grad=trainable and self.trainable)
        con = get_constraint(constraint)
        if con is not None:
            p._krs_constraint = con     # applied after an update: optim.Adagrad.step() / Layer.apply_constraints()
        reg = get_regularizer(regularizer)
        if reg is not None:
            self._regularized.append((p, reg))
            # the penalty is a second gradient contribution to this weight (autograd.CrossLayerFn keeps such a weight's
            # gradient on the main stream)
            p._krs_has_regularizer = True
        pname = name.replace(".", "_")
        if pname in self._parameters:
            self._parameters[pname] = p  # declared as None in __init__
        else:
            self.register_parameter(pname, p)
        self._weight_order.append(p)
        return p

    def _sublayers(self) -> list:
        """Direct sublayers: registered module children AND Layers kept in plain Python lists / tuples / dicts of this
        object (keras tracks those too; nn.Module.children() does not see them)."""
        out, seen = [], set()

        def add(m):
            if isinstance(m, Layer) and id(m) not in seen and m is not self:
                seen.add(id(m))
                out.append(m)

        for m in self.children():
            add(m)
        for v in self.__dict__.values():
            if isinstance(v, (list, tuple)):
                for m in v:
                    add(m)
            elif isinstance(v, dict) and v is not self._modules and v is not self._parameters and v is not self._buffers:
                for m in v.values():
                    add(m)
        return out

    @property
    def weights(self) -> list[torch.nn.Parameter]:
        out = list(self._weight_order)
        for m in self._sublayers():
            out += m.weights
        return out

    def apply_constraints(self) -> int:
        """Projects every weight of this layer and its sublayers that was created with a constraint (what a Keras
        optimizer does after each update).  Returns the number of weights projected."""
        return sum(int(apply_constraint(w)) for w in self.weights)

    @property
    def trainable_weights(self):
        return [w for w in self.weights if w.requires_grad]

    @property
    def variables(self):
        return self.weights

    @property
    def losses(self) -> list[torch.Tensor]:
        """Weight-regularisation penalties of this layer and its sublayers, evaluated on the current weights
        (keras.layers.Layer.losses): add `sum(layer.losses)` to the training loss, as `Model.fit` does."""
        out = [reg(w) for w, reg in self._regularized if w.requires_grad]
        for m in self._sublayers():
            out += m.losses
        return out

    def build(self, *input_shapes) -> None:
        self.built = True

    def call(self, *args, **kwargs):
        raise NotImplementedError

    def _input_shapes(self, args):
        return tuple(tuple(a.shape) if hasattr(a, "shape") else a for a in args)

    def forward(self, *args, **kwargs):
        if not self.built:
            self.build(*self._input_shapes(args))
            self.built = True
        return self.call(*args, **kwargs)

    def get_config(self) -> dict:
        return {"name": self.name, "trainable": self.trainable, "dtype": self.dtype_policy.name}

    @classmethod
    def from_config(cls, config: dict):
        return cls(**config)


def _snake(name: str) -> str:
    out = []
    for i, ch in enumerate(name):
        if ch.isupper() and i and not name[i - 1].isupper():
            out.append("_")
        out.append(ch.lower())
    return "".join(out)
