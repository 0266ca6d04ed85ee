"""Counterparts of the reference's model wrappers (the callers of the hot path).

Same names, constructor arguments, ``nn.Sequential`` layouts (so the shipped ``.pth`` heads
load unchanged -- SURVEY.md appendix A) and forward semantics as

  * ``Signal_vs_Noise/src/model.py:4-29``   two_channel_ligo_binary_classifier
  * ``Signal_vs_Noise/src/model.py:31-52``  one_channel_ligo_binary_classifier
  * ``Glitch_classification/src/model.py:4-39``  (multi-class head with Dropout(0.3))

The reference's own classes also work unchanged on a ``gw_whisper_amd`` encoder (they only
call ``encoder(mel).last_hidden_state[:, -1, :]`` and read ``encoder.config.d_model``);
these copies exist because the reference tree does not travel to the GPU box, and they use
the encoder's ``last_token`` fast path when it is available (inference: only token 1499 goes
through the final LayerNorm; training: the last layer's row-wise ops and their backward run on
the pooled rows only).  The MLP heads are plain ``torch.nn`` on the GPU.
"""

from __future__ import annotations

import torch
import torch.nn as nn


def _pooled(encoder, mel):
    fast = getattr(encoder, "last_token", None)
    if fast is not None:
        return fast(mel)
    return encoder(mel).last_hidden_state[:, -1, :]


class two_channel_ligo_binary_classifier(nn.Module):
    def __init__(self, encoder, num_classes=1):
        super().__init__()
        self.encoder = encoder
        self.classifier = nn.Sequential(
            nn.Linear(self.encoder.config.d_model * 2, 1024), nn.ReLU(),
            nn.Linear(1024, 512), nn.ReLU(),
            nn.Linear(512, 256), nn.ReLU(),
            nn.Linear(256, num_classes))

    def forward(self, mel_tensor_0, mel_tensor_1):
        # both detectors share the encoder weights: one pass over the stacked batch (same arithmetic
        # as the reference's two sequential calls, segments are independent)
        n = mel_tensor_0.shape[0]
        both = _pooled(self.encoder, torch.cat((mel_tensor_0, mel_tensor_1), dim=0))
        return self.classifier(torch.cat((both[:n], both[n:]), dim=1))


class one_channel_ligo_binary_classifier(nn.Module):
    def __init__(self, encoder, num_classes=1):
        super().__init__()
        self.encoder = encoder
        self.classifier = nn.Sequential(
            nn.Linear(self.encoder.config.d_model, 512), nn.ReLU(),
            nn.Linear(512, 256), nn.ReLU(),
            nn.Linear(256, 128), nn.ReLU(),
            nn.Linear(128, 64), nn.ReLU(),
            nn.Linear(64, num_classes))

    def forward(self, mel_tensor_0):
        return self.classifier(_pooled(self.encoder, mel_tensor_0))


class glitch_classifier(nn.Module):
    """``Glitch_classification/src/model.py:4-39`` (there also named
    ``one_channel_ligo_binary_classifier``): d -> 512 -> 256 -> 128 -> C with Dropout(0.3)."""

    def __init__(self, encoder, num_classes):
        super().__init__()
        self.encoder = encoder
        self.classifier = nn.Sequential(
            nn.Linear(self.encoder.config.d_model, 512), nn.ReLU(), nn.Dropout(0.3),
            nn.Linear(512, 256), nn.ReLU(), nn.Dropout(0.3),
            nn.Linear(256, 128), nn.ReLU(), nn.Dropout(0.3),
            nn.Linear(128, num_classes))

    def forward(self, mel_tensor_0):
        return self.classifier(_pooled(self.encoder, mel_tensor_0))
