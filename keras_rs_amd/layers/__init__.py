"""keras_rs.layers surface of the hot path."""

from keras_rs_amd.layers.distributed_embedding import (Adagrad, Adam, DistributedEmbedding, Ftrl, RowwiseAdagrad,
                                                       SGD, concat_features)
from keras_rs_amd.layers.dense import Dense
from keras_rs_amd.layers.distributed_embedding_config import FeatureConfig, TableConfig
from keras_rs_amd.layers.dot_interaction import DotInteraction
from keras_rs_amd.layers.embed_reduce import EmbedReduce, Embedding, Ragged
from keras_rs_amd.layers.feature_cross import FeatureCross
from keras_rs_amd.layers.losses import BinaryCrossentropy, binary_crossentropy

__all__ = ["Adagrad", "Adam", "BinaryCrossentropy", "binary_crossentropy", "Dense", "DistributedEmbedding", "DotInteraction", "EmbedReduce", "Embedding", "FeatureConfig",
           "FeatureCross", "Ftrl", "Ragged", "RowwiseAdagrad", "SGD", "TableConfig", "concat_features"]
