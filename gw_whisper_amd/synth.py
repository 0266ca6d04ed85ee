"""Deterministic synthetic inputs and weights (numpy only, no compute path).

Pretrained ``openai/whisper-*`` weights are not available offline, so the
tests, the golden-vector script and ``bench.py`` all use seeded random weights
laid out exactly like the HF ``WhisperEncoder.state_dict()`` (SURVEY.md
appendix A).  The generator is numpy's PCG64 so the same arrays come out in the
build container and on the GPU box.
"""

from __future__ import annotations

import math

import numpy as np

ENCODER_SIZES = {
    # name: (d_model, layers, heads, ffn)
    "tiny": (384, 4, 6, 1536),
    "base": (512, 6, 8, 2048),
    "small": (768, 12, 12, 3072),
    "micro": (128, 2, 2, 512),      # not a Whisper size: the reduced geometry the parity tests use
}


def sinusoid_table(length: int = 1500, channels: int = 384) -> np.ndarray:
    """Whisper's frozen positional table (float32)."""
    inc = math.log(10000.0) / (channels // 2 - 1)
    inv = np.exp(-inc * np.arange(channels // 2, dtype=np.float32)).astype(np.float32)
    st = np.arange(length, dtype=np.float32)[:, None] * inv[None, :]
    return np.concatenate([np.sin(st), np.cos(st)], axis=1).astype(np.float32)


def encoder_state_dict(d_model: int, layers: int, heads: int, ffn: int, seed: int = 0,
                       n_mels: int = 80) -> dict:
    """Seeded weights with HF key names; every tensor float32.

    Weights ~ N(0, 1/fan_in), biases ~ N(0, 0.02^2), LayerNorm gain 1 + 0.1 N,
    LayerNorm bias 0.1 N -- non-trivial everywhere so a dropped bias or a
    swapped gain shows up in the parity tests.
    """
    del heads
    rng = np.random.default_rng(seed)

    def w(*shape, fan_in):
        return (rng.standard_normal(shape) / math.sqrt(fan_in)).astype(np.float32)

    def b(n, s=0.02):
        return (rng.standard_normal(n) * s).astype(np.float32)

    sd = {}
    sd["conv1.weight"] = w(d_model, n_mels, 3, fan_in=n_mels * 3)
    sd["conv1.bias"] = b(d_model)
    sd["conv2.weight"] = w(d_model, d_model, 3, fan_in=d_model * 3)
    sd["conv2.bias"] = b(d_model)
    sd["embed_positions.weight"] = sinusoid_table(1500, d_model)
    for i in range(layers):
        p = f"layers.{i}."
        sd[p + "self_attn.k_proj.weight"] = w(d_model, d_model, fan_in=d_model)
        sd[p + "self_attn.v_proj.weight"] = w(d_model, d_model, fan_in=d_model)
        sd[p + "self_attn.v_proj.bias"] = b(d_model)
        sd[p + "self_attn.q_proj.weight"] = w(d_model, d_model, fan_in=d_model)
        sd[p + "self_attn.q_proj.bias"] = b(d_model)
        sd[p + "self_attn.out_proj.weight"] = w(d_model, d_model, fan_in=d_model)
        sd[p + "self_attn.out_proj.bias"] = b(d_model)
        sd[p + "self_attn_layer_norm.weight"] = (1.0 + 0.1 * rng.standard_normal(d_model)).astype(np.float32)
        sd[p + "self_attn_layer_norm.bias"] = b(d_model, 0.1)
        sd[p + "fc1.weight"] = w(ffn, d_model, fan_in=d_model)
        sd[p + "fc1.bias"] = b(ffn)
        sd[p + "fc2.weight"] = w(d_model, ffn, fan_in=ffn)
        sd[p + "fc2.bias"] = b(d_model)
        sd[p + "final_layer_norm.weight"] = (1.0 + 0.1 * rng.standard_normal(d_model)).astype(np.float32)
        sd[p + "final_layer_norm.bias"] = b(d_model, 0.1)
    sd["layer_norm.weight"] = (1.0 + 0.1 * rng.standard_normal(d_model)).astype(np.float32)
    sd["layer_norm.bias"] = b(d_model, 0.1)
    return sd


def named_encoder_state_dict(name: str, seed: int = 0) -> dict:
    return encoder_state_dict(*ENCODER_SIZES[name], seed=seed)


def head_state_dict(sizes, seed: int = 0, sequential_stride: int = 2) -> dict:
    """``nn.Sequential`` head weights: keys "0.weight", "2.weight", ... (stride 3 for
    the Glitch head whose Sequential also holds Dropout slots)."""
    rng = np.random.default_rng(seed + 7919)
    sd = {}
    for j in range(len(sizes) - 1):
        fan_in, fan_out = sizes[j], sizes[j + 1]
        bound = 1.0 / math.sqrt(fan_in)
        sd[f"{j * sequential_stride}.weight"] = rng.uniform(-bound, bound, (fan_out, fan_in)).astype(np.float32)
        sd[f"{j * sequential_stride}.bias"] = rng.uniform(-bound, bound, fan_out).astype(np.float32)
    return sd


def dora_adapter(d_out: int, d_in: int, r: int, W0: np.ndarray, seed: int, trained: bool = True):
    """(A [r,in], B [out,r], m [out]).  ``trained=False`` is peft's init (B = 0,
    m = ||W0|| rows => identity); ``trained=True`` perturbs B and m so the
    adapter actually changes the output."""
    rng = np.random.default_rng(seed)
    bound = 1.0 / math.sqrt(d_in)
    A = rng.uniform(-bound, bound, (r, d_in)).astype(np.float32)
    if trained:
        B = (rng.standard_normal((d_out, r)) * 0.05).astype(np.float32)
        m = (np.linalg.norm(W0, axis=1) * (1.0 + 0.1 * rng.standard_normal(d_out))).astype(np.float32)
    else:
        B = np.zeros((d_out, r), np.float32)
        m = np.linalg.norm(W0, axis=1).astype(np.float32)
    return A, B, m


def strain_segments(n: int, seed: int = 0, n_samples: int = 16000) -> np.ndarray:
    """Whitened-like unit-variance 1 s segments at 16 kHz, [n, n_samples] float32
    (BASELINE.md section 3)."""
    rng = np.random.default_rng(seed)
    return rng.standard_normal((n, n_samples)).astype(np.float32)
