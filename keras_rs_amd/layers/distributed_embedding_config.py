"""TableConfig / FeatureConfig: the configuration dataclasses of
keras_rs.layers.DistributedEmbedding (reference: distributed_embedding_config.py:12-132)."""

from __future__ import annotations

import dataclasses
from typing import Any

from keras_rs_amd.layers import base


def _default_table_initializer():
    return base.VarianceScaling(mode="fan_out")


def _serialize(obj):
    if isinstance(obj, base.Initializer):
        return obj.serialize()
    if hasattr(obj, "get_config") and not isinstance(obj, (str, dict)):
        return {"class_name": type(obj).__name__, "config": obj.get_config()}
    return obj


@dataclasses.dataclass(order=True)
class TableConfig:
    """One embedding table: name, vocabulary_size, embedding_dim, initializer, optimizer
    ("sgd" | "adagrad" | "adam" | object), combiner (mean | sum | sqrtn), placement
    (auto | default_device | sparsecore), max_ids_per_partition, max_unique_ids_per_partition."""

    name: str
    vocabulary_size: int
    embedding_dim: int
    initializer: Any = dataclasses.field(default_factory=_default_table_initializer)
    optimizer: Any = "adam"
    combiner: str = "mean"
    placement: str = "auto"
    max_ids_per_partition: int = 256
    max_unique_ids_per_partition: int = 256

    def get_config(self) -> dict[str, Any]:
        return {
            "name": self.name,
            "vocabulary_size": self.vocabulary_size,
            "embedding_dim": self.embedding_dim,
            "initializer": _serialize(self.initializer),
            "optimizer": _serialize(self.optimizer),
            "combiner": self.combiner,
            "placement": self.placement,
            "max_ids_per_partition": self.max_ids_per_partition,
            "max_unique_ids_per_partition": self.max_unique_ids_per_partition,
        }

    @classmethod
    def from_config(cls, config: dict[str, Any]) -> "TableConfig":
        config = dict(config)
        if isinstance(config.get("initializer"), (dict, str)):
            config["initializer"] = base.get_initializer(config["initializer"])
        opt = config.get("optimizer")
        if isinstance(opt, dict):
            from keras_rs_amd.layers.distributed_embedding import optimizer_from_config

            config["optimizer"] = optimizer_from_config(opt)
        return cls(**config)


@dataclasses.dataclass(order=True)
class FeatureConfig:
    """One feature: name, table (TableConfig; several features may share one),
    input_shape (batch, [valence]), output_shape (batch, embedding_dim)."""

    name: str
    table: TableConfig
    input_shape: tuple
    output_shape: tuple

    def get_config(self) -> dict[str, Any]:
        return {
            "name": self.name,
            "table": self.table.get_config(),
            "input_shape": self.input_shape,
            "output_shape": self.output_shape,
        }

    @classmethod
    def from_config(cls, config: dict[str, Any]) -> "FeatureConfig":
        config = dict(config)
        config["table"] = TableConfig.from_config(config["table"])
        config["input_shape"] = tuple(config["input_shape"])
        config["output_shape"] = tuple(config["output_shape"])
        return cls(**config)
