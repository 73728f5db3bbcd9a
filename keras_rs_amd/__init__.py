"""keras_rs_amd: the embedding-lookup + DotInteraction + FeatureCross (DCN-v2) hot path of
keras-rs, rebuilt for AMD MI355X (gfx950): hand-written HIP kernels behind a C ABI
(include/krs.h, libkrs_hip.so), Python layers with the keras_rs.layers signatures on top.

There is no CPU fallback: computing without the HIP library or without a GPU raises.
"""

from keras_rs_amd import layers  # noqa: F401
from keras_rs_amd._lib import KrsError  # noqa: F401

__version__ = "0.1.0"
