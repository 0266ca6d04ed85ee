"""Oracle: pooling + MLP heads (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Restates the reference's own model wrappers:
  * ``Signal_vs_Noise/src/model.py:4-29``   two_channel_ligo_binary_classifier
        head 2d -> 1024 -> 512 -> 256 -> num_classes, ReLU between, pooling
        ``last_hidden_state[:, -1, :]`` per detector then ``cat`` (:25-28)
  * ``Signal_vs_Noise/src/model.py:31-52``  one_channel_ligo_binary_classifier
        head d -> 512 -> 256 -> 128 -> 64 -> num_classes
  * ``Glitch_classification/src/model.py:10-21,34-38``  d -> 512 -> 256 -> 128 -> C
        with Dropout(0.3) (identity in eval)

Head parameters are dicts keyed like the ``nn.Sequential`` ``state_dict()``
("0.weight", "0.bias", "2.weight", ...), matching the shipped ``.pth`` layouts
(SURVEY.md appendix A).
"""

from __future__ import annotations

import numpy as np


def mlp(x: np.ndarray, head: dict) -> np.ndarray:
    """Linear/ReLU stack; the last Linear has no activation."""
    idx = sorted({int(k.split(".")[0]) for k in head})
    for j, i in enumerate(idx):
        x = x @ np.asarray(head[f"{i}.weight"]).astype(x.dtype).T + np.asarray(head[f"{i}.bias"]).astype(x.dtype)
        if j != len(idx) - 1:
            x = np.maximum(x, 0)
    return x


def two_channel_logits(last_h1: np.ndarray, last_l1: np.ndarray, head: dict) -> np.ndarray:
    """``Signal_vs_Noise/src/model.py:25-28``."""
    return mlp(np.concatenate([last_h1, last_l1], axis=1), head)


def one_channel_logits(last: np.ndarray, head: dict) -> np.ndarray:
    """``Signal_vs_Noise/src/model.py:50-52`` / ``Glitch_classification/src/model.py:34-38``."""
    return mlp(last, head)


def binary_labels(logits: np.ndarray) -> np.ndarray:
    """``sigmoid(logits).round()`` as in ``Signal_vs_Noise/src/train.py:94-100``."""
    return (1.0 / (1.0 + np.exp(-logits.astype(np.float64)))).round().astype(np.int64)
