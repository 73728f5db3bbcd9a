"""Stand-in for the Keras 3 symbols keras_rs_amd.keras_adapter touches (keras is not installed here), written as a
CONTRACT: it enforces the parts of the Keras-3 `Layer` protocol the adapter depends on instead of merely accepting
calls, so that a violation fails here and not on first contact with the real package.  What it enforces, with the
Keras 3 behaviour it mirrors (keras/src/layers/layer.py, keras/src/backend/torch/core.py, keras/src/saving/saving_lib.py):

  * `super().__init__()` first: attribute assignment on a layer whose base initialiser has not run raises
    (Keras: "It looks like you are subclassing `Layer` and you forgot to call `super().__init__()`");
  * `add_weight(shape=None, initializer=None, dtype=None, trainable=True, autocast=True, regularizer=None,
    constraint=None, aggregation="none", overwrite_with_gradient=False, name=None)`: that signature and no other
    (everything but `shape` by keyword, unknown keywords rejected); the variable's dtype is `dtype` or the policy's
    VARIABLE dtype; a regulariser becomes an entry of `layer.losses`;
  * torch backend: a `Variable` built from a `torch.nn.Parameter` REUSES that Parameter (as TorchModuleWrapper relies
    on), anything else becomes a new Parameter; names must not contain "/";
  * `__call__`: `build` runs once, on the first call, with the shape(s) of the first argument (`build` takes exactly one
    positional argument here, as the adapter's layers do: Keras then passes the first argument's shape structure),
    `built` is set by the base class afterwards; floating-point tensor arguments are AUTOCAST to the policy's compute
    dtype before `call` (Keras `autocast=True`);
  * `dtype_policy` object with `name` / `compute_dtype` / `variable_dtype`, and the layer properties of the same names;
  * `save_own_variables(store)` / `load_own_variables(store)`: the store behaves like saving_lib's H5Entry -- it offers
    `__setitem__`, `__getitem__`, `keys()`, `items()`, `values()`, `__len__` and NOTHING else (no `in`, no iteration over
    the object itself), refuses keys with "/" (nested groups are the saving library's business), stores numpy arrays and
    hands back dataset-like objects that must be read with `[...]`.
`Model`-shaped composition (layers assigned as attributes are tracked, `weights` / `trainable_weights` /
`non_trainable_weights` walk them in creation order) is what tests/test_keras_adapter.py drives in the order of
examples/ml_perf/model.py:105-212."""

import inspect
import types

import numpy as np
import torch

_TORCH_DT = {"float32": torch.float32, "bfloat16": torch.bfloat16, "float16": torch.float16}


class Variable:
    def __init__(self, initializer, shape=None, dtype=None, trainable=True, autocast=True, aggregation="none", name=None):
        if name is not None and "/" in name:
            raise ValueError(f"Argument `name` must be a string and cannot contain character `/`. Received: name={name}")
        if isinstance(initializer, torch.nn.Parameter):
            self._value = initializer                    # keras/src/backend/torch/core.py: "Reuse same parameter"
            self._value.requires_grad_(bool(trainable) and initializer.is_floating_point())
        elif callable(initializer):
            self._value = torch.nn.Parameter(initializer(tuple(shape), _TORCH_DT.get(str(dtype), torch.float32)),
                                             requires_grad=trainable)
        else:
            t = torch.as_tensor(initializer)
            self._value = torch.nn.Parameter(t.to(_TORCH_DT[str(dtype)]) if dtype else t, requires_grad=trainable)
        self.trainable, self.name, self.autocast = bool(trainable), name, autocast

    @property
    def value(self):
        return self._value

    @property
    def shape(self):
        return tuple(self._value.shape)

    @property
    def dtype(self):
        return str(self._value.dtype).replace("torch.", "")

    def numpy(self):
        v = self._value.detach()
        return (v.float() if v.dtype == torch.bfloat16 else v).cpu().numpy()

    def assign(self, value):
        with torch.no_grad():
            self._value.copy_(torch.as_tensor(value))


class _Init:
    def __init__(self, name):
        self.name = name

    def clone(self):
        return _Init(self.name)

    def __call__(self, shape, dtype=torch.float32, device=None):
        device = device or Layer.DEVICE
        if self.name == "zeros":
            return torch.zeros(shape, dtype=dtype, device=device)
        if self.name == "ones":
            return torch.ones(shape, dtype=dtype, device=device)
        fan_in, fan_out = (shape[0], shape[-1]) if len(shape) > 1 else (shape[0], shape[0])
        lim = float(np.sqrt(6.0 / (fan_in + fan_out)))
        return (torch.rand(shape, dtype=torch.float32, device=device) * 2 - 1).mul_(lim).to(dtype)


class DTypePolicy:
    def __init__(self, name=None):
        self.name = name or "float32"
        if self.name not in ("float32", "bfloat16", "mixed_bfloat16", "float16", "mixed_float16"):
            raise ValueError(f"Cannot convert '{name}' to a mixed precision DTypePolicy.")
        self.compute_dtype = {"mixed_bfloat16": "bfloat16", "mixed_float16": "float16"}.get(self.name, self.name)
        self.variable_dtype = "float32" if self.name.startswith("mixed_") else self.name


_ADD_WEIGHT_PARAMS = ("shape", "initializer", "dtype", "trainable", "autocast", "regularizer", "constraint", "aggregation",
                      "overwrite_with_gradient", "name")


class Layer:
    DEVICE = "cpu"
    _ready = False

    def __init__(self, *, activity_regularizer=None, trainable=True, dtype=None, autocast=True, name=None, **kwargs):
        if kwargs:
            raise ValueError(f"Unrecognized keyword arguments passed to {type(self).__name__}: {kwargs}")
        object.__setattr__(self, "_ready", True)
        self.dtype_policy = dtype if isinstance(dtype, DTypePolicy) else DTypePolicy(dtype)
        self.name = name or type(self).__name__.lower()
        if "/" in self.name:
            raise ValueError("layer names cannot contain '/'")
        self.trainable = trainable
        self.autocast = autocast
        self.built = False
        self._variables, self._layers, self._reg = [], [], []
        self.build_calls = 0

    def __setattr__(self, key, value):
        if not self._ready:
            raise RuntimeError("It looks like you are subclassing `Layer` and you forgot to call `super().__init__()` as the "
                               "first statement in the `__init__()` method.")
        if isinstance(value, Layer) and value not in self._layers:
            self._layers.append(value)
        elif isinstance(value, (list, tuple)):
            for v in value:
                if isinstance(v, Layer) and v not in self._layers:
                    self._layers.append(v)
        object.__setattr__(self, key, value)

    # ---- dtype policy
    @property
    def compute_dtype(self):
        return self.dtype_policy.compute_dtype

    @property
    def variable_dtype(self):
        return self.dtype_policy.variable_dtype

    @property
    def dtype(self):
        return self.variable_dtype

    # ---- variables
    def add_weight(self, shape=None, *args, **kwargs):
        if args:
            raise TypeError("Layer.add_weight: everything but `shape` is passed by keyword in Keras 3 "
                            "(add_weight(shape=None, initializer=None, dtype=None, trainable=True, ...))")
        bad = [k for k in kwargs if k not in _ADD_WEIGHT_PARAMS]
        if bad:
            raise TypeError(f"Layer.add_weight() got unexpected keyword arguments {bad}")
        init = initializers.get(kwargs.get("initializer") or "glorot_uniform")
        dtype = kwargs.get("dtype") or self.variable_dtype
        trainable = bool(kwargs.get("trainable", True)) and self.trainable
        v = Variable(lambda s, d: init(s, d, self.DEVICE), shape=tuple(shape or ()), dtype=dtype, trainable=trainable,
                     autocast=kwargs.get("autocast", True), name=kwargs.get("name"))
        reg = regularizers.get(kwargs.get("regularizer"))     # (Layer.add_weight resolves identifiers itself)
        if reg is not None:
            self._reg.append((v, reg))
        v.constraint = kwargs.get("constraint")
        self._variables.append(v)
        return v

    def _track_variable(self, v):
        if not isinstance(v, Variable):
            raise TypeError("_track_variable takes a keras Variable")
        self._variables.append(v)

    @property
    def weights(self):
        out = list(self._variables)
        for sub in self._layers:
            out += sub.weights
        return out

    @property
    def trainable_weights(self):
        return [v for v in self.weights if v.trainable]

    @property
    def non_trainable_weights(self):
        return [v for v in self.weights if not v.trainable]

    @property
    def losses(self):
        out = [reg(v.value) for v, reg in self._reg]
        for sub in self._layers:
            out += sub.losses
        return out

    # ---- call protocol
    def build(self, input_shape):
        pass

    def _autocast(self, x):
        cd = _TORCH_DT[self.compute_dtype]
        if isinstance(x, torch.Tensor):
            return x.to(cd) if (x.is_floating_point() and x.dtype != cd) else x
        if isinstance(x, (list, tuple)):
            return type(x)(self._autocast(v) for v in x)
        return x          # dicts of ids, numpy integer arrays, None: untouched

    @staticmethod
    def _shape_of(x):
        if isinstance(x, dict):
            return {k: Layer._shape_of(v) for k, v in x.items()}
        if isinstance(x, (list, tuple)):
            return [Layer._shape_of(v) for v in x]
        return tuple(x.shape) if hasattr(x, "shape") else None

    def __call__(self, *args, **kwargs):
        if not self._ready:
            raise RuntimeError("the layer was never initialised (super().__init__() missing)")
        if not self.built:
            params = [p for p in inspect.signature(self.build).parameters.values()
                      if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD)]
            if len(params) != 1:
                raise TypeError("this contract covers build(self, <one shape argument>): Keras passes the first call argument's "
                                "shape structure to it")
            self.build(self._shape_of(args[0]) if args else None)
            self.build_calls += 1
            self.built = True
        if self.autocast:
            args = tuple(self._autocast(a) for a in args)
            kwargs = {k: self._autocast(v) for k, v in kwargs.items()}
        return self.call(*args, **kwargs)

    # ---- saving
    def save_own_variables(self, store):
        for i, v in enumerate(self._variables):
            store[str(i)] = v.numpy()

    def load_own_variables(self, store):
        if len(store.keys()) != len(self._variables):
            raise ValueError(f"Layer '{self.name}' expected {len(self._variables)} variables, but received {len(store.keys())}")
        for i, v in enumerate(self._variables):
            v.assign(store[str(i)][...])

    def get_config(self):
        return {"name": self.name, "trainable": self.trainable, "dtype": self.dtype_policy.name}

    @classmethod
    def from_config(cls, config):
        return cls(**config)


class _Dataset:
    """What a store hands back: read it with `[...]` (h5py.Dataset), or through numpy's array protocol."""

    def __init__(self, arr):
        self._arr = arr
        self.shape, self.dtype = arr.shape, arr.dtype

    def __getitem__(self, idx):
        if idx is not Ellipsis and idx != ():
            raise TypeError("the contract reads whole entries: store[key][...]")
        return self._arr.copy()

    def __array__(self, dtype=None, copy=None):
        return self._arr.astype(dtype) if dtype is not None else self._arr.copy()


class Store:
    """saving_lib.H5Entry's surface: __setitem__ / __getitem__ / keys / items / values / __len__, nothing else."""

    __slots__ = ("_d",)

    def __init__(self):
        self._d = {}

    def __setitem__(self, key, value):
        if not isinstance(key, str) or "/" in key or not key:
            raise ValueError(f"store keys are flat, non-empty strings without '/' (got {key!r}): nested groups belong to the "
                             "saving library")
        arr = np.asarray(value)
        if arr.dtype == object:
            raise TypeError(f"store['{key}']: only numeric arrays can be saved")
        self._d[key] = arr.copy()

    def __getitem__(self, key):
        return _Dataset(self._d[key])

    def __len__(self):
        return len(self._d)

    def keys(self):
        return self._d.keys()

    def items(self):
        return [(k, _Dataset(v)) for k, v in self._d.items()]

    def values(self):
        return [_Dataset(v) for v in self._d.values()]

    __contains__ = None       # `key in store` raises TypeError, as on an object without __contains__ / __iter__
    __iter__ = None


class Model(Layer):
    """keras.Model as far as examples/ml_perf/model.py:105-163 uses it: a Layer that tracks the layers assigned to it."""


def _get_init(x):
    return x if isinstance(x, _Init) else _Init(str(x))


def linear(t):
    return t


_ACTS = {None: linear, "linear": linear, "relu": torch.relu, "sigmoid": torch.sigmoid, "tanh": torch.tanh}   # get(None) is linear


def _get_act(x):
    return x if callable(x) else _ACTS[x]


def _ser_act(fn):
    for k, v in _ACTS.items():
        if v is fn and k is not None:
            return k
    return getattr(fn, "__name__", "custom")


def _get_reg(x):
    if x is None or callable(x):
        return x
    table = {"l1": lambda w: 0.01 * w.abs().sum(), "l2": lambda w: 0.01 * w.square().sum(),
             "l1_l2": lambda w: 0.01 * w.abs().sum() + 0.01 * w.square().sum()}
    if x not in table:
        raise ValueError(f"Could not interpret regularizer identifier: {x}")
    return table[x]


regularizers = types.SimpleNamespace(get=_get_reg)
layers = types.SimpleNamespace(Layer=Layer)
initializers = types.SimpleNamespace(get=_get_init, serialize=lambda i: i.name)
activations = types.SimpleNamespace(get=_get_act, serialize=_ser_act)
backend = types.SimpleNamespace(backend=lambda: "torch")
