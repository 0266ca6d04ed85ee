"""Oracle: Whisper encoder forward on torch-CPU (TEST INFRASTRUCTURE -- see oracle/__init__.py).

The same restatement as ``oracle/encoder.py`` (``HF:models/whisper/modeling_whisper.py:592-646, 379-413, 284-356``)
written with torch's CPU operators (oneDNN convolutions, threaded fp32 GEMMs, the fused CPU SDPA kernel) -- the
operators the reference's HF encoder itself runs on when it is put on a CPU.  It exists for ``bench.py``'s
``cpu_baseline`` leg: the numpy oracle is a correctness checker and a poor timing baseline (2.3 segments/s on 128
threads); this one is what "the reference on the host cores" costs (SURVEY.md section 8d measured HF/torch-CPU itself at
6.4 segments/s on 8 cores).  Pinned by the same HF goldens as the numpy oracle (tests/test_oracle_encoder.py).
Never imported by the product.
"""

from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from .encoder import EncCfg, HEAD_DIM


@torch.no_grad()
def encoder_forward(params: dict, mel, cfg: EncCfg, chunk: int = 8) -> np.ndarray:
    """mel [B, 80, 3000] (numpy or CPU tensor) -> last_hidden_state [B, 1500, d] float32 numpy; ``chunk`` segments at
    a time."""
    P = {k: torch.as_tensor(np.asarray(v), dtype=torch.float32) for k, v in params.items()}
    x_all = torch.as_tensor(np.asarray(mel), dtype=torch.float32)
    d, H = cfg.d_model, cfg.heads
    outs = []
    for b0 in range(0, x_all.shape[0], chunk):
        x = x_all[b0:b0 + chunk]
        x = F.gelu(F.conv1d(x, P["conv1.weight"], P["conv1.bias"], padding=1))
        x = F.gelu(F.conv1d(x, P["conv2.weight"], P["conv2.bias"], stride=2, padding=1))
        x = x.permute(0, 2, 1) + P["embed_positions.weight"]
        B, T, _ = x.shape
        for i in range(cfg.layers):
            p = f"layers.{i}."
            h = F.layer_norm(x, (d,), P[p + "self_attn_layer_norm.weight"], P[p + "self_attn_layer_norm.bias"])
            q = F.linear(h, P[p + "self_attn.q_proj.weight"], P[p + "self_attn.q_proj.bias"]) * HEAD_DIM ** -0.5
            k = F.linear(h, P[p + "self_attn.k_proj.weight"])
            v = F.linear(h, P[p + "self_attn.v_proj.weight"], P[p + "self_attn.v_proj.bias"])
            q, k, v = (t.view(B, T, H, HEAD_DIM).transpose(1, 2) for t in (q, k, v))
            a = F.scaled_dot_product_attention(q, k, v, scale=1.0)     # HF's sdpa path (q already scaled, no mask)
            a = a.transpose(1, 2).reshape(B, T, d)
            x = x + F.linear(a, P[p + "self_attn.out_proj.weight"], P[p + "self_attn.out_proj.bias"])
            h = F.layer_norm(x, (d,), P[p + "final_layer_norm.weight"], P[p + "final_layer_norm.bias"])
            x = x + F.linear(F.gelu(F.linear(h, P[p + "fc1.weight"], P[p + "fc1.bias"])), P[p + "fc2.weight"], P[p + "fc2.bias"])
        outs.append(F.layer_norm(x, (d,), P["layer_norm.weight"], P["layer_norm.bias"]))
    return torch.cat(outs).numpy()
