"""The loss of the DLRM step on MI355X: `keras.losses.BinaryCrossentropy()` as the reference's ml_perf example compiles
it (examples/ml_perf/main.py:201-210: probabilities in -- the top MLP ends in a sigmoid, model.py:105-163 --, mean over
the batch), forward and backward in ONE pass over the predictions (krs_bce_fwd_bwd, csrc/loss.hip)."""

from __future__ import annotations

import torch

from keras_rs_amd.autograd import BinaryCrossentropyFn


class BinaryCrossentropy:
    """Callable like keras.losses.BinaryCrossentropy(from_logits=False, reduction="sum_over_batch_size"):
    loss(y_true, y_pred) -> scalar.  epsilon = keras.backend.epsilon() = 1e-7 (the clip of the probabilities)."""

    def __init__(self, from_logits: bool = False, epsilon: float = 1e-7, name: str = "binary_crossentropy"):
        if from_logits:
            raise NotImplementedError("BinaryCrossentropy(from_logits=True): the reference's model ends in a sigmoid "
                                      "(examples/ml_perf/model.py:105-163) and compiles the loss with probabilities")
        self.epsilon, self.name = float(epsilon), name

    def __call__(self, y_true: torch.Tensor, y_pred: torch.Tensor) -> torch.Tensor:
        if y_true.numel() != y_pred.numel():
            raise ValueError(f"BinaryCrossentropy: y_true {tuple(y_true.shape)} and y_pred {tuple(y_pred.shape)} differ")
        return BinaryCrossentropyFn.apply(y_pred, y_true, self.epsilon)

    def get_config(self) -> dict:
        return {"name": self.name, "from_logits": False, "epsilon": self.epsilon}


def binary_crossentropy(y_true: torch.Tensor, y_pred: torch.Tensor, epsilon: float = 1e-7) -> torch.Tensor:
    return BinaryCrossentropyFn.apply(y_pred, y_true, epsilon)
