"""DotInteraction on MI355X: drop-in for keras_rs.layers.DotInteraction
(keras_rs/src/layers/feature_interaction/dot_interaction.py:84-234)."""

from __future__ import annotations

from typing import Any, Sequence

import torch

from keras_rs_amd import _lib as L
from keras_rs_amd.autograd import DotInteractionFn
from keras_rs_amd.layers import base


class DotInteraction(base.Layer):
    """Pairwise dot products of feature vectors (DLRM interaction).

    Args: self_interaction (keep the diagonal), skip_gather (masked [B, F*F] output
    instead of the packed lower triangle) -- dot_interaction.py:84-94.
    """

    def __init__(self, self_interaction: bool = False, skip_gather: bool = False, **kwargs: Any):
        super().__init__(**kwargs)
        self.self_interaction = self_interaction
        self.skip_gather = skip_gather

    def _input_shapes(self, args):
        return (tuple(tuple(t.shape) for t in args[0]),)

    def call(self, inputs: Sequence[torch.Tensor]) -> torch.Tensor:
        shape = tuple(inputs[0].shape)
        for idx, t in enumerate(inputs):
            if len(shape) != 2:  # dot_interaction.py:156-160
                raise ValueError("All feature tensors inside `inputs` should have rank 2. "
                                 f"Received rank {len(shape)} at index {idx}.")
            if tuple(t.shape) != shape:  # :162-167
                raise ValueError("All feature tensors in `inputs` should have the same shape. Found at least "
                                 f"one conflict: shape = {shape} at index 0 and shape = {tuple(t.shape)} at "
                                 f"index {idx}.")
        for t in inputs:
            L.require_device(t, "DotInteraction input")
        cd = self.compute_dtype
        feats = [t if t.dtype == cd else t.to(cd) for t in inputs]
        # the features of one lookup slab behind a head of its reserved width (the DLRM input list
        # [bottom_mlp_output, *embeddings] with equal widths): their gradient can join the slab's in the kernel
        relay, n_heads = None, 0
        if all(f is t for f, t in zip(feats, inputs)):
            from keras_rs_amd.layers.distributed_embedding import slab_grad_relay, slab_views_run

            slab, k = slab_views_run(inputs)
            if slab is not None and slab.dtype == cd and len(inputs) <= 64 and \
                    all(tuple(h.shape) == shape for h in inputs[:k]):   # (the kernel's accumulate mask has 64 bits)
                relay, n_heads = slab_grad_relay(slab), k
        return DotInteractionFn.apply(self.self_interaction, self.skip_gather, relay, n_heads, *feats)

    def compute_output_shape(self, input_shape):
        n = len(input_shape)
        batch = input_shape[0][0]
        if self.skip_gather:
            return (batch, n * n)
        return (batch, n * (n + 1) // 2 if self.self_interaction else n * (n - 1) // 2)

    def get_config(self) -> dict:
        config = super().get_config()
        config.update({"self_interaction": self.self_interaction, "skip_gather": self.skip_gather})
        return config
