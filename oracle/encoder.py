"""Oracle: Whisper encoder forward (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Numpy restatement of ``WhisperEncoder.forward`` as the reference uses it
(``Signal_vs_Noise/src/train.py:227-228`` builds it, ``src/model.py:25-26``
calls it).  The arithmetic is in the third-party ``transformers`` package:

  * ``HF:models/whisper/modeling_whisper.py:55-64``    sinusoids (pos table)
  * ``HF:models/whisper/modeling_whisper.py:592-646``  encoder forward
        conv1+gelu :618, conv2(stride 2)+gelu :619, permute + pos :621-624,
        layers :627-640, final layer_norm :642
  * ``HF:models/whisper/modeling_whisper.py:379-413``  pre-LN encoder layer
  * ``HF:models/whisper/modeling_whisper.py:284-356``  attention: q scaled by
        head_dim**-0.5 BEFORE q k^T (:309), k_proj has no bias (:279),
        softmax(q k^T) v with no mask (:215-238), out_proj (:354)

Parameters are a ``dict[str, np.ndarray]`` keyed exactly like the HF
``state_dict()``.  ``emulate_bf16=True`` rounds every
matrix-multiply operand to bfloat16 (round-to-nearest-even) and accumulates in
the working dtype -- that is what the MI355X throughput path does, so kernel
tests can use a tight tolerance; ``False`` is the plain fp32/fp64 math the
golden vectors from HF pin.
"""

from __future__ import annotations

from dataclasses import dataclass

import numpy as np
from scipy.special import erf

LN_EPS = 1e-5
T_IN = 3000
T_OUT = 1500
N_MELS = 80
HEAD_DIM = 64


@dataclass(frozen=True)
class EncCfg:
    d_model: int = 384
    layers: int = 4
    heads: int = 6
    ffn: int = 1536

    @staticmethod
    def named(name: str) -> "EncCfg":
        return {
            "tiny": EncCfg(384, 4, 6, 1536),
            "base": EncCfg(512, 6, 8, 2048),
            "small": EncCfg(768, 12, 12, 3072),
        }[name]


def bf16_round(x: np.ndarray) -> np.ndarray:
    """Round float32 to the nearest bfloat16 (ties to even), returned as float32."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32)
    r = ((u >> 16) & 1) + np.uint32(0x7FFF)
    out = ((u + r) & np.uint32(0xFFFF0000)).astype(np.uint32)
    nan = np.isnan(x)
    out = np.where(nan, u | np.uint32(0x00400000), out).astype(np.uint32)
    return out.view(np.float32).reshape(x.shape)


def sinusoids(length: int, channels: int, max_timescale: float = 10000.0) -> np.ndarray:
    """HF:modeling_whisper.py:55-64 (float32 like torch's default)."""
    inc = float(np.log(max_timescale)) / (channels // 2 - 1)   # python float: stays f32 like torch
    inv = np.exp(-inc * np.arange(channels // 2, dtype=np.float32)).astype(np.float32)
    st = np.arange(length, dtype=np.float32)[:, None] * inv[None, :]
    return np.concatenate([np.sin(st), np.cos(st)], axis=1).astype(np.float32)


def gelu(x):
    """Exact (erf) GELU -- ``nn.functional.gelu`` default."""
    return 0.5 * x * (1.0 + erf(x * 0.7071067811865476))


def layer_norm(x, w, b, eps=LN_EPS):
    mu = x.mean(axis=-1, keepdims=True)
    xc = x - mu
    var = (xc * xc).mean(axis=-1, keepdims=True)
    return xc / np.sqrt(var + eps) * w + b


def _mm(a, b, emulate_bf16, dtype):
    """a @ b with optional bf16 operand rounding, accumulate in ``dtype``."""
    if emulate_bf16:
        a = bf16_round(a.astype(np.float32)).astype(dtype)
        b = bf16_round(b.astype(np.float32)).astype(dtype)
    return np.matmul(a, b)


def conv1d_k3(x, w, b, stride, emulate_bf16, dtype):
    """Conv1d(kernel 3, padding 1) on token-major input.

    x [B, T, Cin] ; w [Cout, Cin, 3] ; returns [B, T_out, Cout].
    out[t] = sum_k x[stride t + k - 1] @ w[:, :, k].T + b
    """
    B, T, Cin = x.shape
    xp = np.zeros((B, T + 2, Cin), dtype=dtype)
    xp[:, 1:T + 1] = x
    t_out = (T + 2 - 3) // stride + 1
    acc = np.zeros((B, t_out, w.shape[0]), dtype=dtype)
    for k in range(3):
        xs = xp[:, k:k + stride * (t_out - 1) + 1:stride]
        acc += _mm(xs, w[:, :, k].T.astype(dtype), emulate_bf16, dtype)
    return acc + b.astype(dtype)


def attention(q, k, v, heads, emulate_bf16, dtype):
    """softmax(q k^T) v per head; q already scaled.  q,k,v [B, T, d]."""
    B, T, d = q.shape
    dh = d // heads
    qh = q.reshape(B, T, heads, dh).transpose(0, 2, 1, 3)
    kh = k.reshape(B, T, heads, dh).transpose(0, 2, 1, 3)
    vh = v.reshape(B, T, heads, dh).transpose(0, 2, 1, 3)
    if emulate_bf16:
        qh = bf16_round(qh.astype(np.float32)).astype(dtype)
        kh = bf16_round(kh.astype(np.float32)).astype(dtype)
        vh = bf16_round(vh.astype(np.float32)).astype(dtype)
    s = np.matmul(qh, kh.transpose(0, 1, 3, 2))
    s = s - s.max(axis=-1, keepdims=True)
    p = np.exp(s)
    l = p.sum(axis=-1, keepdims=True)
    if emulate_bf16:
        # the HIP kernel feeds un-normalised bf16 P to the PV product and
        # divides by the fp32 row sum afterwards
        o = np.matmul(bf16_round(p.astype(np.float32)).astype(dtype), vh) / l
    else:
        o = np.matmul(p / l, vh)
    return o.transpose(0, 2, 1, 3).reshape(B, T, d)


def encoder_forward(params: dict, mel: np.ndarray, cfg: EncCfg, dtype=np.float32,
                    emulate_bf16: bool = False, return_stages: bool = False):
    """mel [B, 80, 3000] -> last_hidden_state [B, 1500, d] (and per-stage dict).

    HF:modeling_whisper.py:592-646.
    """
    mel = np.asarray(mel)
    if mel.shape[-1] != T_IN or mel.shape[-2] != N_MELS:
        raise ValueError(f"Whisper expects the mel input features to be of length {T_IN}, "
                         f"but found {mel.shape[-1]}.")
    P = {k: np.asarray(v).astype(dtype) for k, v in params.items()}
    d, H = cfg.d_model, cfg.heads
    stages = {}
    x = mel.astype(dtype).transpose(0, 2, 1)                       # [B, 3000, 80]
    x = gelu(conv1d_k3(x, P["conv1.weight"], P["conv1.bias"], 1, emulate_bf16, dtype))
    stages["conv1"] = x
    x = gelu(conv1d_k3(x, P["conv2.weight"], P["conv2.bias"], 2, emulate_bf16, dtype))
    x = x + P["embed_positions.weight"][None]
    stages["embed"] = x
    scale = dtype(HEAD_DIM ** -0.5)
    for i in range(cfg.layers):
        p = f"layers.{i}."
        h = layer_norm(x, P[p + "self_attn_layer_norm.weight"], P[p + "self_attn_layer_norm.bias"])
        q = (_mm(h, P[p + "self_attn.q_proj.weight"].T, emulate_bf16, dtype)
             + P[p + "self_attn.q_proj.bias"]) * scale
        k = _mm(h, P[p + "self_attn.k_proj.weight"].T, emulate_bf16, dtype)
        v = _mm(h, P[p + "self_attn.v_proj.weight"].T, emulate_bf16, dtype) + P[p + "self_attn.v_proj.bias"]
        a = attention(q, k, v, H, emulate_bf16, dtype)
        if i == 0:
            stages["l0.q"], stages["l0.k"], stages["l0.v"], stages["l0.attn"] = q, k, v, a
        x = x + _mm(a, P[p + "self_attn.out_proj.weight"].T, emulate_bf16, dtype) + P[p + "self_attn.out_proj.bias"]
        if i == 0:
            stages["l0.post_attn"] = x
        h = layer_norm(x, P[p + "final_layer_norm.weight"], P[p + "final_layer_norm.bias"])
        f = gelu(_mm(h, P[p + "fc1.weight"].T, emulate_bf16, dtype) + P[p + "fc1.bias"])
        x = x + _mm(f, P[p + "fc2.weight"].T, emulate_bf16, dtype) + P[p + "fc2.bias"]
        stages[f"l{i}.out"] = x
    out = layer_norm(x, P["layer_norm.weight"], P["layer_norm.bias"])
    stages["final"] = out
    return (out, stages) if return_stages else out


def last_token(params: dict, mel: np.ndarray, cfg: EncCfg, **kw) -> np.ndarray:
    """``encoder(mel).last_hidden_state[:, -1, :]`` (reference ``src/model.py:25``)."""
    return encoder_forward(params, mel, cfg, **kw)[:, -1, :]
