"""DistributedEmbedding on MI355X: drop-in for keras_rs.layers.DistributedEmbedding.

Keeps the public surface of keras_rs/src/layers/embedding/base_distributed_embedding.py
(constructor :468-477, preprocess :630-729, call :740-808, get_embedding_tables
:810-825, has_sparsecores :940-988, get_config/from_config :1053-1139) and fills
the backend hook set `_sparsecore_{init,build,preprocess,call,get_embedding_tables}`
(:990-1042) with the MI355X path -- the third sibling of the reference's jax/
and tensorflow/ backends:

  * all features of a placement that share an embedding width are looked up by
    ONE launch of the fused gather+pool kernel (K1) instead of the per-feature
    python loop of :910-928;
  * "default_device" tables keep the reference's autodiff semantics (dense
    [V, D] gradients on ordinary trainable weights, computed by K2);
  * "sparsecore" tables take the accelerated path: their per-table optimizer
    (SGD / Adagrad of TableConfig.optimizer) runs inside the backward on the
    touched rows only (the design of jax/embedding_lookup.py:174-273);
  * "auto" resolves to "sparsecore" when an MI355X is present and the table's
    optimizer can be fused, else "default_device".

`preprocess` may be called concurrently from loader threads
(examples/ml_perf/main.py:56-87): it only reads immutable configuration.
"""

from __future__ import annotations

import dataclasses
import math
import threading
import weakref
from typing import Any, Sequence

import numpy as np
import torch

from keras_rs_amd import _lib as L
from keras_rs_amd.autograd import EmbedBagFn, EmbedBagFusedFn
from keras_rs_amd.embedding_ops import FusedBags
from keras_rs_amd.layers import base
from keras_rs_amd.layers.distributed_embedding_config import FeatureConfig
from keras_rs_amd.layers.embed_reduce import Ragged, check_shapes_compatible

SUPPORTED_PLACEMENTS = ("auto", "default_device", "sparsecore")


# ---- per-table optimizers that K2 can fuse -------------------------------------------
# The reference accepts keras SGD / Adagrad / Adam / Ftrl objects (or their names) on 'sparsecore'
# tables and maps them to the SparseCore library's optimizer specs, rejecting the options that
# library lacks (jax/config_conversion.py:211-288).  Keras is not installed here, so these small
# classes carry the same constructor arguments and defaults; any object with the same attribute
# names (e.g. a real keras optimizer) resolves the same way.
@dataclasses.dataclass
class SGD:
    learning_rate: float = 0.01

    def get_config(self):
        return {"learning_rate": self.learning_rate}


@dataclasses.dataclass
class Adagrad:
    learning_rate: float = 0.001
    initial_accumulator_value: float = 0.1

    def get_config(self):
        return {"learning_rate": self.learning_rate, "initial_accumulator_value": self.initial_accumulator_value}


@dataclasses.dataclass
class RowwiseAdagrad:
    """NOT a reference optimizer: Adagrad with ONE accumulator per table row (the mean of the squared gradient
    over the row's columns; the FBGEMM / TorchRec "rowwise_adagrad" rule, no epsilon).  Opt-in for 'sparsecore'
    tables: the fused update then moves 8 accumulator bytes per touched row instead of 2 x dim x 4 -- at C3 K2's
    traffic drops to a third.  Trajectories differ from exact Adagrad (equal only where a row's gradient has the
    same magnitude in every column)."""

    learning_rate: float = 0.001
    initial_accumulator_value: float = 0.1

    def get_config(self):
        return {"learning_rate": self.learning_rate, "initial_accumulator_value": self.initial_accumulator_value}


@dataclasses.dataclass
class Adam:
    learning_rate: float = 0.001
    beta_1: float = 0.9
    beta_2: float = 0.999
    epsilon: float = 1e-7

    def get_config(self):
        return dataclasses.asdict(self)


@dataclasses.dataclass
class Ftrl:
    learning_rate: float = 0.001
    learning_rate_power: float = -0.5
    initial_accumulator_value: float = 0.1
    l1_regularization_strength: float = 0.0
    l2_regularization_strength: float = 0.0
    beta: float = 0.0

    def get_config(self):
        return dataclasses.asdict(self)


def optimizer_from_config(cfg: dict):
    return {"SGD": SGD, "Adagrad": Adagrad, "Adam": Adam, "Ftrl": Ftrl,
            "RowwiseAdagrad": RowwiseAdagrad}[cfg["class_name"]](**cfg.get("config", {}))


@dataclasses.dataclass(frozen=True)
class FusedOptimizer:
    """What K2 needs to run a table's optimizer inside the backward."""

    kind: str                 # "sgd" | "adagrad" | "adam" | "ftrl" | "adagrad_rowwise"
    lr: Any                   # float, or a schedule: callable(step) / callable() (jax/config_conversion.py:136-176)
    acc0: float = 0.0         # initial value of the (first) slot plane: Adagrad / FTRL accumulator
    consts: tuple = ()        # adam: (beta_1, beta_2, epsilon); ftrl: (lr_power, l1, l2, beta)

    @property
    def n_slot_planes(self) -> int:
        return {"sgd": 0, "adagrad": 1, "adam": 2, "ftrl": 2, "adagrad_rowwise": 1}[self.kind]

    def lr_at(self, step: int) -> float:
        """Learning rate of the update with 0-based index `step` (keras `iterations`)."""
        if not callable(self.lr):
            return float(self.lr)
        return float(self.lr(step)) if _n_positional(self.lr) == 1 else float(self.lr())

    def hyper(self, step: int):
        """The four floats krs_embed_bag_bwd_fused_{adam,ftrl} take; `step` is the 1-based update count."""
        if self.kind == "adam":
            b1, b2, eps = self.consts
            return (b1, b2, eps, math.sqrt(1.0 - b2 ** step) / (1.0 - b1 ** step))
        return self.consts if self.kind == "ftrl" else None

    def new_slot(self, shape, device):
        if self.kind == "adagrad_rowwise":
            return torch.full((shape[0],), self.acc0, dtype=torch.float32, device=device)
        if self.kind == "adagrad":
            return torch.full(shape, self.acc0, dtype=torch.float32, device=device)
        if self.kind in ("adam", "ftrl"):
            slot = torch.zeros((2,) + tuple(shape), dtype=torch.float32, device=device)
            if self.kind == "ftrl":
                slot[0].fill_(self.acc0)
            return slot
        return None


def _n_positional(fn) -> int:
    import inspect

    try:
        args = inspect.getfullargspec(fn).args
    except TypeError:
        return 99
    return len(args) if inspect.isfunction(fn) else len(args) - 1


def resolve_fused_optimizer(opt) -> FusedOptimizer | None:
    """The FusedOptimizer of a TableConfig.optimizer, or None when it cannot be fused (an option
    the reference rejects as well, or an unknown optimizer).  Learning rates: a number, or a schedule
    called with the step count or with nothing (jax/config_conversion.py:136-176)."""
    if isinstance(opt, str):
        cls = {"sgd": SGD, "adagrad": Adagrad, "adam": Adam, "ftrl": Ftrl}.get(opt.lower())
        if cls is None:
            return None
        opt = cls()
    name = type(opt).__name__.lower()
    # keras optimizers evaluate `learning_rate` at the current step; the schedule itself is `_learning_rate`
    lr = getattr(opt, "_learning_rate", None)
    if lr is None:
        lr = getattr(opt, "learning_rate", None)
    if lr is None or (callable(lr) and _n_positional(lr) > 1):
        return None
    # options without a fused counterpart (jax/config_conversion.py:232-283)
    if any(getattr(opt, k, None) is not None for k in ("clipnorm", "global_clipnorm", "loss_scale_factor")) or \
            getattr(opt, "use_ema", False):
        return None
    lr = lr if callable(lr) else float(lr)
    if name == "sgd":
        if getattr(opt, "nesterov", False) or float(getattr(opt, "momentum", 0.0) or 0.0) != 0.0:
            return None
        return FusedOptimizer("sgd", lr)
    if name == "adagrad":
        if float(getattr(opt, "epsilon", 1e-7)) != 1e-7:
            return None
        return FusedOptimizer("adagrad", lr, float(getattr(opt, "initial_accumulator_value", 0.1)))
    if name == "rowwiseadagrad":
        return FusedOptimizer("adagrad_rowwise", lr, float(getattr(opt, "initial_accumulator_value", 0.1)))
    if name == "adam":
        if getattr(opt, "amsgrad", False):
            return None
        return FusedOptimizer("adam", lr, 0.0, (float(getattr(opt, "beta_1", 0.9)), float(getattr(opt, "beta_2", 0.999)),
                                                float(getattr(opt, "epsilon", 1e-7))))
    if name == "ftrl":
        if float(getattr(opt, "l2_shrinkage_regularization_strength", 0.0) or 0.0) != 0.0:
            return None
        return FusedOptimizer("ftrl", lr, float(getattr(opt, "initial_accumulator_value", 0.1)),
                              (float(getattr(opt, "learning_rate_power", -0.5)),
                               float(getattr(opt, "l1_regularization_strength", 0.0)),
                               float(getattr(opt, "l2_regularization_strength", 0.0)),
                               float(getattr(opt, "beta", 0.0))))
    return None


@dataclasses.dataclass(eq=True, unsafe_hash=True, order=True)
class PlacementAndPath:
    placement: str
    path: str


def _is_feature_config(x) -> bool:
    return isinstance(x, FeatureConfig)


def _is_placement_leaf(x) -> bool:
    return isinstance(x, PlacementAndPath)


MAX_GROUP_DESCRIPTORS = 512      # = kMaxLdsDesc of csrc/embed_bag_bwd.hip


@dataclasses.dataclass
class _Group:
    """Features of one placement sharing (embedding_dim): one FusedBags, one launch."""

    placement: str
    dim: int
    paths: list
    table_index: list          # per feature: index into `tables`
    table_configs: list        # unique TableConfig objects
    bags: FusedBags | None = None
    fused: FusedOptimizer | None = None   # shared by the group's tables (learning rates may differ)
    table_opts: list | None = None        # per table: its FusedOptimizer (learning rate / schedule)
    step: int = 0                         # fused updates applied so far (Adam bias correction, schedules)

    @property
    def fused_kind(self) -> str | None:
        return None if self.fused is None else self.fused.kind

    _constants: Any = None                # embedding_ops.StepConstants when a constant of the update depends on `step`

    def next_hyper(self):
        """Called once per fused update, from the backward pass: counts the update, refreshes the constants that depend on
        the count -- scheduled learning rates (in the kernel descriptors) and Adam's bias correction, both kept in DEVICE
        memory (embedding_ops.StepConstants) -- and returns the Adam / FTRL constants (None for SGD / Adagrad).  While a
        stream is capturing nothing is counted or written: GraphedStep does that before every replay."""
        scheduled = bool(self.table_opts) and any(callable(o.lr) for o in self.table_opts)
        adam = self.fused.kind == "adam"
        if not scheduled and not adam:
            from keras_rs_amd import graphs

            graphs.count_update(self)      # (per replay under GraphedStep)
            return self.fused.hyper(self.step)
        if self._constants is None:
            from keras_rs_amd.embedding_ops import StepConstants

            self._constants = StepConstants(
                self, lambda: self.bags, (lambda step: [o.lr_at(step) for o in self.table_opts]) if scheduled else None,
                self.fused.consts[:2] if adam else None)
        self._constants.on_backward()
        if adam:
            b1, b2, eps = self.fused.consts
            return (b1, b2, eps, self._constants.bias_correction)
        return self.fused.hyper(self.step)


class DistributedEmbedding(base.Layer):
    """Args (base_distributed_embedding.py:468-477): feature_configs (nested structure of
    FeatureConfig), table_stacking ("auto" | names), update_stats, **kwargs.

    Not in the reference: slab_lead_cols.  The fused-optimizer ("sparsecore") lookups of one
    embedding width land in ONE [B, n*dim] buffer whose column blocks are the returned features;
    slab_lead_cols reserves that many extra leading columns in it, so that
    layers.concat_features([dense, *embeddings]) (the DLRM / DCN interaction input,
    examples/ml_perf/model.py:204-207) can place `dense` there and return the buffer itself
    instead of copying every embedding."""

    def __init__(self, feature_configs, *, table_stacking="auto", update_stats: bool = False,
                 slab_lead_cols: int = 0, **kwargs: Any):
        super().__init__(**kwargs)
        self.slab_lead_cols = int(slab_lead_cols)
        self._table_stacking = table_stacking
        self.update_stats = update_stats
        self._lock = threading.Lock()
        self._anchor = None
        self._err_dev = None      # device int32[1]: the lookup kernels OR their KRS_FLAG_* bits into it
        self._err_host = None     # its page-locked mirror, refreshed asynchronously after every call
        self._err_event = None
        self._init_feature_configs_structures(feature_configs)
        self._groups: dict[str, list[_Group]] = {}
        self._table_params: dict[int, torch.nn.Parameter] = {}
        self._table_slots: dict[int, torch.Tensor | None] = {}
        self._slot_buffer_names: dict[int, str] = {}
        self._stacks: list = []
        self._stack_of: dict[int, tuple] = {}      # table -> (stacked parameter, first row) for named stacks
        if "sparsecore" in self._placement_to_path_to_feature_config:
            self._sparsecore_init(self._placement_to_path_to_feature_config["sparsecore"], table_stacking)
        if "default_device" in self._placement_to_path_to_feature_config:
            self._default_device_init(self._placement_to_path_to_feature_config["default_device"], table_stacking)

    # ------------------------------------------------------------------ structure
    def _init_feature_configs_structures(self, feature_configs) -> None:
        self._feature_configs = feature_configs
        placement_and_paths = []
        self._placement_to_path_to_feature_config: dict[str, dict[str, FeatureConfig]] = {}
        has_sc = None
        for path, fc in base.flatten_with_path(feature_configs, is_leaf=_is_feature_config):
            if not isinstance(fc, FeatureConfig):
                raise ValueError(f"feature_configs leaves must be FeatureConfig, got {type(fc).__name__}")
            placement = fc.table.placement
            if placement == "auto":
                if has_sc is None:
                    has_sc = self.has_sparsecores()
                fusable = resolve_fused_optimizer(fc.table.optimizer) is not None
                placement = "sparsecore" if (has_sc and fusable) else "default_device"
            spath = ".".join(str(e) for e in path)
            if placement not in SUPPORTED_PLACEMENTS:  # base:573-577
                raise ValueError(f"Feature '{spath}' with name '{fc.name}' has unsupported placement '{placement}'.")
            placement_and_paths.append(PlacementAndPath(placement, spath))
            self._placement_to_path_to_feature_config.setdefault(placement, {})[spath] = fc
        self._feature_deeply_nested_placement_and_paths = base.pack_sequence_as(
            feature_configs, placement_and_paths, is_leaf=_is_feature_config)

    @classmethod
    def has_sparsecores(cls) -> bool:
        """True when the accelerated embedding path is available: on this backend, an
        AMD GPU visible to torch (the role SparseCores play at base:940-988)."""
        return bool(torch.cuda.is_available() and getattr(torch.version, "hip", None))

    # ------------------------------------------------------------------ backend hooks
    def _make_groups(self, placement: str, feature_configs: dict[str, FeatureConfig]) -> None:
        by_dim: dict[int, _Group] = {}
        for path, fc in feature_configs.items():
            g = by_dim.get(fc.table.embedding_dim)
            if g is None:
                g = by_dim[fc.table.embedding_dim] = _Group(placement, fc.table.embedding_dim, [], [], [])
            # one table per distinct TableConfig object (base:836-852)
            idx = next((i for i, t in enumerate(g.table_configs) if t is fc.table), None)
            if idx is None:
                idx = len(g.table_configs)
                g.table_configs.append(fc.table)
            g.paths.append(path)
            g.table_index.append(idx)
        # K2's vector apply kernel keeps a launch's feature / table descriptors in LDS, MAX_GROUP_DESCRIPTORS of each; a
        # wider group would fall to its any-shape kernel (one column per lane, no hot-row path: correct, slow under skew --
        # ADVICE r5).  Groups are independent launches anyway, so a wider one is split between tables (the features of
        # one table stay together); each part gets its own slab.
        groups = []
        for g in by_dim.values():
            if len(g.paths) <= MAX_GROUP_DESCRIPTORS and len(g.table_configs) <= MAX_GROUP_DESCRIPTORS:
                groups.append(g)
                continue
            feats_of = [[i for i, t in enumerate(g.table_index) if t == ti] for ti in range(len(g.table_configs))]
            part = _Group(placement, g.dim, [], [], [])
            for ti, feats in enumerate(feats_of):
                if len(feats) > MAX_GROUP_DESCRIPTORS:
                    raise ValueError(f"Table '{g.table_configs[ti].name}' is shared by {len(feats)} features; a fused launch "
                                     f"takes at most {MAX_GROUP_DESCRIPTORS} features of one table")
                if part.paths and (len(part.paths) + len(feats) > MAX_GROUP_DESCRIPTORS
                                   or len(part.table_configs) + 1 > MAX_GROUP_DESCRIPTORS):
                    groups.append(part)
                    part = _Group(placement, g.dim, [], [], [])
                part.table_configs.append(g.table_configs[ti])
                for i in feats:
                    part.paths.append(g.paths[i])
                    part.table_index.append(len(part.table_configs) - 1)
            if part.paths:
                groups.append(part)
        self._groups[placement] = groups

    def _parse_table_stacking(self, table_stacking, feature_configs) -> list:
        """`table_stacking` (base:455-466, jax/distributed_embedding.py:413-453): None / "auto" / a list of table
        names / a list of such lists.  Here tables of one embedding width are ALWAYS looked up by one launch (the
        purpose of stacking on SparseCore), so "auto" and None need no physical change; named stacks additionally
        get ONE contiguous [sum V, D] buffer per stack (one allocation, one checkpoint entry), the per-table
        tensors being row windows of it.  Returns the stacks as lists of TableConfig."""
        if table_stacking is None or (isinstance(table_stacking, str) and table_stacking == "auto"):
            return []
        bad = ValueError(f"Unsupported table stacking {table_stacking}, must be None, 'auto', or sequences of table "
                         "names to stack.")
        if isinstance(table_stacking, str) or not isinstance(table_stacking, (list, tuple)) or not table_stacking:
            raise bad
        if isinstance(table_stacking[0], str):
            groups = [list(table_stacking)]
        elif all(isinstance(g, (list, tuple)) for g in table_stacking):
            groups = [list(g) for g in table_stacking]
        else:
            raise bad
        by_name = {fc.table.name: fc.table for fc in feature_configs.values()}
        stacks, seen = [], set()
        for names in groups:
            if not all(isinstance(n, str) for n in names):
                raise bad
            tcs = []
            for n in names:
                if n not in by_name:
                    raise ValueError(f"table_stacking names table '{n}', which no 'sparsecore' feature uses")
                if n in seen:
                    raise ValueError(f"table '{n}' appears in more than one stack")
                seen.add(n)
                tcs.append(by_name[n])
            if len({tc.embedding_dim for tc in tcs}) > 1:
                raise ValueError(f"tables {names} cannot be stacked: their embedding_dim differ")
            if len(tcs) > 1:
                stacks.append(tcs)
        return stacks

    def _sparsecore_init(self, feature_configs, table_stacking) -> None:
        self._stacks = self._parse_table_stacking(table_stacking, feature_configs)
        if not self.has_sparsecores():
            raise self._unsupported_placement_error("sparsecore")
        for path, fc in feature_configs.items():
            if resolve_fused_optimizer(fc.table.optimizer) is None:
                raise NotImplementedError(
                    f"Table '{fc.table.name}': the 'sparsecore' placement fuses the table optimizer into the "
                    f"backward and supports SGD, Adagrad, Adam and Ftrl (constant learning rate, the option "
                    f"set of the reference's SparseCore path); got {fc.table.optimizer!r}. Use "
                    "placement='default_device' for other optimizers.")
        self._make_groups("sparsecore", feature_configs)

    def _default_device_init(self, feature_configs, table_stacking) -> None:
        del table_stacking
        self._make_groups("default_device", feature_configs)

    def _build_groups(self, placement: str) -> None:
        for gi, g in enumerate(self._groups.get(placement, [])):
            if g.bags is not None:
                continue
            tables, slots, lrs, topts = [], [], [], []
            kinds: set = set()
            if placement == "sparsecore":
                # named stacks: ONE parameter per stack, the tables are row windows of it
                for tcs in getattr(self, "_stacks", []):
                    if id(tcs[0]) in self._table_params or not all(tc in g.table_configs for tc in tcs):
                        continue
                    rows = sum(tc.vocabulary_size for tc in tcs)
                    stack = self.add_weight((rows, g.dim), "zeros", "sparsecore_stack_" + "_".join(tc.name for tc in tcs),
                                            trainable=False)
                    r0 = 0
                    with torch.no_grad():
                        for tc in tcs:
                            init = base.get_initializer(tc.initializer)
                            stack[r0:r0 + tc.vocabulary_size] = init((tc.vocabulary_size, g.dim), stack.dtype, stack.device)
                            self._table_params[id(tc)] = stack.data[r0:r0 + tc.vocabulary_size]
                            self._table_slots[id(tc)] = None
                            self._stack_of[id(tc)] = (stack, r0)
                            r0 += tc.vocabulary_size
            for tc in g.table_configs:
                key = id(tc)
                if key not in self._table_params:
                    p = self.add_weight((tc.vocabulary_size, tc.embedding_dim), tc.initializer,
                                        f"{placement}_{tc.name}_embeddings",
                                        trainable=placement == "default_device")
                    self._table_params[key] = p
                    self._table_slots[key] = None
                p = self._table_params[key]
                tables.append(p)
                lr = 0.0
                if placement == "sparsecore":
                    fo = resolve_fused_optimizer(tc.optimizer)
                    lr = fo.lr_at(0)
                    topts.append(fo)
                    kinds.add(dataclasses.replace(fo, lr=0.0))   # everything but the learning rate is per group
                    if self._table_slots[key] is None:
                        slot = fo.new_slot(p.shape, p.device)
                        if slot is not None:
                            # optimizer slot variables are part of the layer's state, as in the reference
                            # (add_weight'ed slot variables, jax/distributed_embedding.py:316-345): a persistent
                            # buffer travels with state_dict() / load_state_dict() / .to()
                            bname = f"{placement}_{tc.name}_slot".replace(".", "_")
                            self.register_buffer(bname, slot, persistent=True)
                            self._slot_buffer_names[key] = bname
                        self._table_slots[key] = slot
                slots.append(self._table_slots[key])
                lrs.append(lr)
            if len(kinds) > 1:
                raise NotImplementedError("Tables of one embedding width on 'sparsecore' must share the optimizer "
                                          "type and its constants (only the learning rate may differ per table)")
            g.fused = kinds.pop() if kinds else None
            g.table_opts = topts or None
            fcs = self._placement_to_path_to_feature_config[placement]
            feats = [(g.table_index[i], fcs[p].table.combiner, i * g.dim) for i, p in enumerate(g.paths)]
            g.bags = FusedBags(tables, feats, slots=slots, lrs=lrs)

    def _sparsecore_build(self, input_shapes) -> None:
        del input_shapes
        self._build_groups("sparsecore")
        if self._anchor is None:
            dev = self._groups["sparsecore"][0].bags.tables[0].device
            self._anchor = torch.zeros((), device=dev, requires_grad=True)

    def _default_device_build(self, input_shapes) -> None:
        del input_shapes
        self._build_groups("default_device")

    # preprocess: per group, concatenate the features' ids feature-major into one buffer
    def _fuse_inputs(self, placement: str, inputs: dict, weights: dict | None):
        fcs = self._placement_to_path_to_feature_config[placement]
        out_inputs, out_weights = {}, {}
        for gi, g in enumerate(self._groups[placement]):
            dev = g.bags.tables[0].device
            id_parts, w_parts, hots, lens = [], [], [], []
            ragged = False
            batch = None
            use_w = weights is not None
            for path in g.paths:
                x = inputs[path]
                w = None if weights is None else weights[path]
                x, w = _ragged_numpy_to_csr(x, w)
                if isinstance(x, Ragged):
                    ragged = True
                    vals = _to_tensor(x.values)
                    offs = np.asarray(_to_numpy(x.row_offsets), dtype=np.int64)
                    b = len(offs) - 1
                    id_parts.append(vals.reshape(-1))
                    lens.append(np.diff(offs))
                    hots.append(None)
                    if w is not None:
                        w_parts.append(_to_tensor(w.values if isinstance(w, Ragged) else w).reshape(-1))
                else:
                    t = _to_tensor(x)
                    if t.dim() == 1:
                        # rank-1: no reduction; weights only survive for "sum" (embed_reduce.py:224)
                        if fcs[path].table.combiner != "sum":
                            w = None if w is None else torch.ones(t.shape)
                        t = t.reshape(-1, 1)
                    elif t.dim() != 2:
                        raise ValueError(f"Feature '{path}': inputs must be rank 1 or 2, got {tuple(t.shape)}")
                    b = t.shape[0]
                    id_parts.append(t.reshape(-1))
                    hots.append(int(t.shape[1]))
                    lens.append(np.full(b, t.shape[1], dtype=np.int64))
                    if w is not None:
                        wt = _to_tensor(w).float()
                        if wt.numel() != t.numel():
                            raise ValueError(f"Feature '{path}': weights shape {tuple(wt.shape)} does not match "
                                             f"inputs shape {tuple(t.shape)}")
                        w_parts.append(wt.reshape(-1))
                if batch is None:
                    batch = b
                elif batch != b:
                    raise ValueError("All features of a DistributedEmbedding call must share the batch size")
            ids = _cat_index(id_parts).to(dev, non_blocking=True)
            offsets = None
            if ragged:
                offsets = torch.from_numpy(
                    np.concatenate([[0], np.cumsum(np.concatenate(lens))]).astype(np.int32)).to(dev, non_blocking=True)
                hots_t = None
            else:
                hots_t = tuple(hots)
            key = f"group{gi}"
            out_inputs[key] = {"ids": ids, "offsets": offsets, "hots": hots_t, "batch": batch}
            if use_w:
                if len(w_parts) != len(g.paths):
                    raise ValueError("weights must be given for every feature or for none")
                out_weights[key] = torch.cat([p.reshape(-1) for p in w_parts]).float().to(dev, non_blocking=True)
        res = {"inputs": out_inputs}
        if weights is not None:
            res["weights"] = out_weights
        return res

    def _sparsecore_preprocess(self, inputs, weights, training=False):
        del training
        return self._fuse_inputs("sparsecore", inputs, weights)

    def _default_device_preprocess(self, inputs, weights, training=False):
        del training
        return self._fuse_inputs("default_device", inputs, weights)

    # ---- module state: optimizer slots are buffers, update counts travel as extra state ---------------
    def _apply(self, fn, recurse=True):
        """.to() / .cuda() / .float(): torch replaces buffer tensors; re-point the kernel descriptors at them."""
        out = super()._apply(fn, recurse)
        for key, bname in getattr(self, "_slot_buffer_names", {}).items():
            self._table_slots[key] = self._buffers[bname]
        for key, (stack, r0) in getattr(self, "_stack_of", {}).items():
            self._table_params[key] = stack.data[r0:r0 + self._table_params[key].shape[0]]
        for groups in getattr(self, "_groups", {}).values():
            for g in groups:
                if g.bags is not None:
                    g.bags.slots = [self._table_slots[id(tc)] for tc in g.table_configs]
                    g.bags.tables = [self._table_params[id(tc)] for tc in g.table_configs]
                    g.bags._tab_key = None
        self._err_dev = self._err_host = self._err_event = None
        return out

    def get_extra_state(self):
        """Fused-update counts per group (Adam bias correction, learning-rate schedules): the reference keeps
        `_iterations` as a layer variable (jax/distributed_embedding.py:340-345)."""
        return {"iterations": {f"{pl}/{gi}": int(g.step) for pl, groups in self._groups.items()
                               for gi, g in enumerate(groups)}}

    def set_extra_state(self, state) -> None:
        its = (state or {}).get("iterations", {})
        for pl, groups in self._groups.items():
            for gi, g in enumerate(groups):
                if f"{pl}/{gi}" in its:
                    g.step = int(its[f"{pl}/{gi}"])

    # ---- out-of-range ids: flagged by the kernels, raised lazily (no per-step host sync) ---------------
    def _err_flag(self, device) -> torch.Tensor | None:
        if device.type != "cuda":
            return None
        if self._err_dev is None:
            self._err_dev = torch.zeros(1, dtype=torch.int32, device=device)
            self._err_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        return self._err_dev

    def _err_snapshot(self) -> None:
        """Queues a copy of the error word into page-locked memory behind the lookups just launched."""
        if self._err_dev is not None:
            self._err_host.copy_(self._err_dev, non_blocking=True)
            if base.stream_capturing():
                # inside a graph the copy is a node of every replay; there is no event to poll: check_ids(wait=True)
                # between replays waits for the device instead
                self._err_event, self._err_in_graph = None, True
                return
            self._err_event = torch.cuda.Event()
            self._err_event.record()

    def check_ids(self, wait: bool = False) -> None:
        """Raises IndexError if a lookup launched by an earlier call met an id outside [0, vocabulary_size)
        (such ids contribute nothing; they are never clamped).  Called at the start of every `call` without
        waiting for the GPU (only snapshots that have already arrived are looked at), with wait=True from
        get_embedding_tables() and by callers that want the verdict on the last step now."""
        if base.stream_capturing():
            return
        ev = self._err_event
        if ev is None:
            if not (wait and getattr(self, "_err_in_graph", False)):
                return
            torch.cuda.current_stream(self._err_dev.device).synchronize()   # replays of a captured step
        elif wait:
            ev.synchronize()
        elif not ev.query():
            return
        self._err_event = None
        if int(self._err_host.item()) & L.FLAG_ID_OUT_OF_RANGE:
            self._err_dev.zero_()
            self._err_host.zero_()
            raise IndexError("DistributedEmbedding: an embedding id was out of range for its table "
                             "(ids are never clamped; the lookup contributed nothing)")

    def _call_groups(self, placement: str, inputs: dict, weights: dict | None):
        outputs = {}
        for gi, g in enumerate(self._groups[placement]):
            key = f"group{gi}"
            fi = inputs[key]
            w = None if weights is None else weights[key]
            out_dtype = self.compute_dtype
            err = self._err_flag(fi["ids"].device)
            if placement == "sparsecore":
                lead = self.slab_lead_cols if len(self._groups[placement]) == 1 else 0
                slab, *out = EmbedBagFusedFn.apply(g.bags, fi["ids"], fi["batch"], fi["hots"], fi["offsets"], w,
                                                   out_dtype, g, self._anchor, lead, err)
                for i, o in enumerate(out):  # lets layers.concat_features find the slab (zero-copy concat)
                    o._krs_slab = (slab, lead + i * g.dim, len(out), lead)
            else:
                out = EmbedBagFn.apply(g.bags, fi["ids"], fi["batch"], fi["hots"], fi["offsets"], w, out_dtype,
                                       err if err is not None else False, *g.bags.tables)
            for path, o in zip(g.paths, out):
                outputs[path] = o
        return outputs

    def _sparsecore_call(self, inputs, weights=None, training=False):
        del training
        return self._call_groups("sparsecore", inputs, weights)

    def _default_device_call(self, inputs, weights=None, training=False):
        del training
        return self._call_groups("default_device", inputs, weights)

    def _get_tables(self, placement: str) -> dict:
        tables = {}
        for g in self._groups.get(placement, []):
            for tc in g.table_configs:
                tables[tc.name] = self._table_params[id(tc)].detach()
        return tables

    def _sparsecore_get_embedding_tables(self):
        return self._get_tables("sparsecore")

    def _default_device_get_embedding_tables(self):
        return self._get_tables("default_device")

    # ------------------------------------------------------------------ public API
    def _input_shapes(self, args):
        inputs = args[0]
        if self._is_preprocessed(inputs):
            return (inputs,)
        return (base.map_structure_up_to(self._feature_configs, lambda fc, x: _shape_of(x), self._feature_configs,
                                         inputs, is_leaf=_is_feature_config),)

    def build(self, input_shapes=None, *_) -> None:
        if self.built:
            return
        with self._lock:
            if self.built:
                return
            if input_shapes is not None:
                self._verify_input_shapes(input_shapes)
            if "sparsecore" in self._placement_to_path_to_feature_config:
                self._sparsecore_build(None)
            if "default_device" in self._placement_to_path_to_feature_config:
                self._default_device_build(None)
            self.built = True

    def preprocess(self, inputs, weights=None, training: bool = False):
        """Bundles the (possibly ragged, possibly host-resident) inputs of every placement
        into the fused feature-major form the kernels consume (base:630-729)."""
        base.assert_same_structure(self._feature_configs, inputs, is_leaf=_is_feature_config)
        if weights is not None:
            base.assert_same_structure(self._feature_configs, weights, is_leaf=_is_feature_config)
        if not self.built:
            self.build(self._input_shapes((inputs,))[0])

        def to_placement_to_path(tensors):
            result = {p: {} for p in self._placement_to_path_to_feature_config}
            base.map_structure_up_to(
                self._feature_deeply_nested_placement_and_paths,
                lambda pp, x: result[pp.placement].__setitem__(pp.path, x),
                self._feature_deeply_nested_placement_and_paths, tensors, is_leaf=_is_placement_leaf)
            return result

        p_inputs = to_placement_to_path(inputs)
        p_weights = to_placement_to_path(weights) if weights is not None else None
        pre = {}
        if "sparsecore" in p_inputs:
            pre["sparsecore"] = self._sparsecore_preprocess(
                p_inputs["sparsecore"], p_weights["sparsecore"] if p_weights is not None else None, training)
        if "default_device" in p_inputs:
            pre["default_device"] = self._default_device_preprocess(
                p_inputs["default_device"], p_weights["default_device"] if p_weights is not None else None, training)
        return {"preprocessed_inputs_per_placement": pre}

    def _is_preprocessed(self, inputs) -> bool:
        return isinstance(inputs, dict) and "preprocessed_inputs_per_placement" in inputs

    def call(self, inputs, weights=None, training: bool = False):
        self.check_ids()   # verdict on earlier steps, if it has arrived (no wait)
        pre = inputs if self._is_preprocessed(inputs) else self.preprocess(inputs, weights, training)
        pre = pre["preprocessed_inputs_per_placement"]
        outs = {}
        if "sparsecore" in pre:
            outs["sparsecore"] = self._sparsecore_call(**pre["sparsecore"], training=training)
        if "default_device" in pre:
            outs["default_device"] = self._default_device_call(**pre["default_device"], training=training)
        self._err_snapshot()
        return base.map_structure_up_to(
            self._feature_deeply_nested_placement_and_paths, lambda pp: outs[pp.placement][pp.path],
            self._feature_deeply_nested_placement_and_paths, is_leaf=_is_placement_leaf)

    def get_embedding_tables(self) -> dict[str, torch.Tensor]:
        """{TableConfig.name: [vocabulary_size, embedding_dim]} (base:810-825)."""
        if not self.built:
            self.build(None)
        self.check_ids(wait=True)
        tables = {}
        if "sparsecore" in self._placement_to_path_to_feature_config:
            tables.update(self._sparsecore_get_embedding_tables())
        if "default_device" in self._placement_to_path_to_feature_config:
            tables.update(self._default_device_get_embedding_tables())
        return tables

    def set_embedding_tables(self, tables: dict) -> None:
        """Overwrites tables by name (jax/distributed_embedding.py:746-756)."""
        if not self.built:
            self.build(None)
        with torch.no_grad():
            for groups in self._groups.values():
                for g in groups:
                    for tc in g.table_configs:
                        if tc.name in tables:
                            p = self._table_params[id(tc)]
                            p.copy_(torch.as_tensor(np.asarray(tables[tc.name].detach().cpu()
                                                               if isinstance(tables[tc.name], torch.Tensor)
                                                               else tables[tc.name])).to(p.dtype))

    def compute_output_shape(self, input_shapes):
        self._verify_input_shapes(input_shapes)
        return base.map_structure_up_to(self._feature_configs, lambda fc: fc.output_shape, self._feature_configs,
                                        is_leaf=_is_feature_config)

    def get_config(self) -> dict:
        # shared tables are serialised once, features refer to them by index (base:1053-1093)
        table_dicts: list = []
        table_index: dict[int, int] = {}

        def ser(fc: FeatureConfig):
            d = fc.get_config()
            if id(fc.table) not in table_index:
                table_index[id(fc.table)] = len(table_dicts)
                table_dicts.append(d["table"])
            d["table"] = table_index[id(fc.table)]
            return d

        config = super().get_config()
        config["feature_configs"] = base.map_structure_up_to(self._feature_configs, ser, self._feature_configs,
                                                             is_leaf=_is_feature_config)
        config["tables"] = table_dicts
        config["table_stacking"] = self._table_stacking
        if self.slab_lead_cols:
            config["slab_lead_cols"] = self.slab_lead_cols
        return config

    @classmethod
    def from_config(cls, config: dict):
        config = dict(config)
        table_dicts = config.pop("tables")
        tables: list = [None] * len(table_dicts)

        def is_fc_dict(d):
            return isinstance(d, dict) and isinstance(d.get("name"), str) and "table" in d

        def de(d):
            d = dict(d)
            idx = d["table"]
            d["table"] = table_dicts[idx]
            fc = FeatureConfig.from_config(d)
            if tables[idx] is None:
                tables[idx] = fc.table
            else:
                fc.table = tables[idx]
            return fc

        flat = base.flatten(config["feature_configs"], is_leaf=is_fc_dict)
        config["feature_configs"] = base.pack_sequence_as(config["feature_configs"], [de(d) for d in flat],
                                                          is_leaf=is_fc_dict)
        return cls(**config)

    def _verify_input_shapes(self, input_shapes) -> None:
        if self._is_preprocessed(input_shapes):
            return

        def verify(fc: FeatureConfig, shape):
            if shape is None:
                return
            if not isinstance(shape, (tuple, list)) or not all(isinstance(d, (int, type(None))) for d in shape):
                raise ValueError(f"Received invalid input shape {shape}.")
            if len(shape) < 1:
                raise ValueError(f"Received input shape {shape}. Rank must be 1 or above.")
            # The reference discards this result (base:1179-1181): a mismatch is accepted
            # (examples/ml_perf/main.py:175-177 relies on it).
            check_shapes_compatible(tuple(fc.input_shape), tuple(shape))

        base.map_structure_up_to(self._feature_configs, verify, self._feature_configs, input_shapes,
                                 is_leaf=_is_feature_config)

    def _unsupported_placement_error(self, placement: str) -> Exception:
        return NotImplementedError(f"No AMD GPU visible to torch: the '{placement}' placement is not available.")


# ---------------------------------------------------------------------- helpers
def slab_grad_relay(slab: torch.Tensor):
    """The SlabGradRelay of a lookup slab (created on first use; see autograd.SlabGradRelay)."""
    from keras_rs_amd.autograd import SlabGradRelay

    relay = getattr(slab, "_krs_grad_relay", None)
    if relay is None:
        relay = slab._krs_grad_relay = SlabGradRelay()
    return relay


def slab_views_run(tensors: Sequence[torch.Tensor]):
    """(slab, n_heads) when the trailing tensors are ALL the feature views of one DistributedEmbedding slab, in
    order, and the leading ones have exactly the total width of its reserved columns; else (None, 0)."""
    tensors = list(tensors)
    info = [getattr(t, "_krs_slab", None) for t in tensors]
    if not tensors or info[-1] is None:
        return None, 0
    start = len(tensors)
    while start > 0 and info[start - 1] is not None and info[start - 1][0] is info[-1][0]:
        start -= 1
    run = info[start:]
    slab, _, n_views, lead = run[0]
    dim = (slab.shape[1] - lead) // n_views
    ordered = len(run) == n_views and all(r[1] == lead + i * dim for i, r in enumerate(run))
    heads = tensors[:start]
    if not ordered or any(h.dim() != 2 or h.shape[0] != slab.shape[0] for h in heads):
        return None, 0
    if sum(h.shape[1] for h in heads) != lead:
        return None, 0
    return slab, start


def concat_features(tensors: Sequence[torch.Tensor]) -> torch.Tensor:
    """torch.cat(tensors, dim=-1) for the interaction input of a DLRM / DCN model
    (examples/ml_perf/model.py:204-207 concatenates the bottom-MLP output and every embedding).

    When the trailing tensors are ALL the features of one DistributedEmbedding slab, in order, the
    slab is used in place: with `slab_lead_cols` equal to the total width of the tensors in front of
    them, those are written into the reserved columns and the slab itself is returned (no copy of
    the embeddings, and their gradient arrives as one matrix); otherwise the result is a two-piece
    concat of [heads..., slab].  Anything else falls back to torch.cat."""
    from keras_rs_amd.autograd import SlabFillFn

    tensors = list(tensors)
    info = [getattr(t, "_krs_slab", None) for t in tensors]
    start = len(tensors)
    while start > 0 and info[start - 1] is not None and info[start - 1][0] is info[-1][0]:
        start -= 1
    run = info[start:]
    if run:
        slab, _, n_views, lead = run[0]
        dim = (slab.shape[1] - lead) // n_views
        ordered = len(run) == n_views and all(r[1] == lead + i * dim for i, r in enumerate(run))
        heads = tensors[:start]
        if ordered and all(h.dim() == 2 and h.shape[0] == slab.shape[0] and h.dtype == slab.dtype for h in heads):
            # the reserved columns can be handed out once: a second concat with other heads gets a copy
            if sum(h.shape[1] for h in heads) == lead and not getattr(slab, "_krs_lead_taken", False):
                if not heads:
                    return slab
                slab._krs_lead_taken = True
                relay = slab_grad_relay(slab)
                out = SlabFillFn.apply(slab, relay, *heads)
                relay.out_ref = weakref.ref(out)
                return out
            slab_grad_relay(slab).disabled = True     # a second consumer of the slab: no in-place gradient join
            return torch.cat(heads + [slab[:, lead:]], dim=-1)
    return torch.cat(tensors, dim=-1)


def _to_numpy(x):
    if isinstance(x, torch.Tensor):
        return x.detach().cpu().numpy()
    return np.asarray(x)


def _to_tensor(x) -> torch.Tensor:
    if isinstance(x, torch.Tensor):
        return x
    if hasattr(x, "numpy") and callable(x.numpy):
        x = x.numpy()
    return torch.from_numpy(np.ascontiguousarray(x))


def _cat_index(parts: Sequence[torch.Tensor]) -> torch.Tensor:
    dt = torch.int64 if any(p.dtype == torch.int64 for p in parts) else torch.int32
    parts = [p.to(dt) for p in parts]
    if all(p.device.type == "cpu" for p in parts) and torch.cuda.is_available():
        # host ids: concatenate straight into page-locked memory, so that the upload that follows is a
        # true asynchronous DMA (the loader threads of data.ThreadedDataLoader overlap it with compute)
        buf = torch.empty(sum(p.numel() for p in parts), dtype=dt, pin_memory=True)
        return torch.cat(parts, out=buf) if len(parts) > 1 else buf.copy_(parts[0].reshape(-1))
    return torch.cat(parts) if len(parts) > 1 else parts[0].contiguous()


def _shape_of(x):
    if isinstance(x, Ragged):
        return (len(x.row_offsets) - 1, None)
    if isinstance(x, np.ndarray) and x.dtype == object:
        return (len(x), None)
    return tuple(x.shape) if hasattr(x, "shape") else None


def _ragged_numpy_to_csr(x, w):
    """numpy object arrays of rows (the ragged form of base:31-92) -> Ragged CSR.
    Results equal the reference's pad-to-dense form (padding carries weight 0)."""
    if isinstance(x, np.ndarray) and x.dtype == object and len(x) > 0:
        rx = Ragged.from_rows(list(x), dtype=np.asarray(x[0]).dtype if np.asarray(x[0]).dtype.kind == "i" else np.int32)
        rw = None
        if w is not None:
            rw = Ragged(Ragged.from_rows(list(w), dtype=np.float32).values, rx.row_offsets)
        return rx, rw
    return x, w
