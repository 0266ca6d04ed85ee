"""Oracle: peft DoRA ``Linear`` (TEST INFRASTRUCTURE -- see oracle/__init__.py).

The reference creates DoRA adapters with
``LoraConfig(use_dora=True, r, lora_alpha, target_modules)`` +
``get_peft_model`` (``Signal_vs_Noise/src/train.py:263-264``,
``MLGWSC-1/train.py:695-696``) and reloads them with
``PeftModel.from_pretrained`` (``Signal_vs_Noise/src/train.py:50``).  The math is
in the third-party ``peft`` package, pinned 0.12.0 by the reference
(``requirements.txt:177``) and NOT installed in the build container, so this
file restates the published algorithm of peft 0.12.0
``tuners/lora/layer.py`` (``Linear.forward``) and ``tuners/lora/dora.py``
(``DoraLinearLayer.forward`` / ``get_weight_norm``):

    s        = lora_alpha / r
    W'       = W0 + s * B @ A                      # [out, in]
    n        = || W' ||_2 per OUTPUT ROW           # detached from autograd
    y        = W0 x + b + (m/n - 1) * (W0 x) + (m/n) * s * B (A x)
             = (m/n) * (W' x) + b

Init: A ~ kaiming-uniform(a=sqrt 5), B = 0, m = ||W0|| rows => identity.

PARITY UNPINNED beyond: identity at init, merged == unmerged, and the on-disk
key schema of the adapters the reference ships (tests/golden/adapter_schema.json).
"""

from __future__ import annotations

import numpy as np


def dora_weight_norm(W0, A, B, scaling):
    """Row-wise L2 norm of W0 + s B A  (peft dora.py ``get_weight_norm``, dim=1)."""
    Wp = W0 + scaling * (B @ A)
    return np.sqrt((Wp * Wp).sum(axis=1))


def dora_merge(W0, A, B, m, scaling):
    """Effective dense weight: (m / ||W'||)[:, None] * W'."""
    Wp = W0 + scaling * (B @ A)
    n = np.sqrt((Wp * Wp).sum(axis=1))
    return (m / n)[:, None] * Wp


def dora_linear_unmerged(x, W0, bias, A, B, m, scaling):
    """peft's literal forward (layer.py Linear.forward + dora.py forward)."""
    base = x @ W0.T
    result = base + (0.0 if bias is None else bias)
    lora = (x @ A.T) @ B.T
    n = dora_weight_norm(W0, A, B, scaling)
    g = (m / n)[None, :]
    return result + (g - 1.0) * base + g * lora * scaling


def dora_linear_merged(x, W0, bias, A, B, m, scaling):
    y = x @ dora_merge(W0, A, B, m, scaling).T
    return y if bias is None else y + bias


def dora_grads(x, dy, W0, A, B, m, scaling):
    """Gradients of sum(y * dy) wrt A, B, m and x with the norm detached.

    x [N, in], dy [N, out].  With g = m/n (n constant):
        y   = g * (x W0^T + s (x A^T) B^T) + b
        dm  = sum_N dy * (W' x) / n
        dB  = s * (g*dy)^T (x A^T)            [out, r]
        dA  = s * B^T (g*dy)^T x              [r, in]
        dx  = (g*dy) W'                       [N, in]
    """
    Wp = W0 + scaling * (B @ A)
    n = np.sqrt((Wp * Wp).sum(axis=1))
    g = m / n
    gdy = dy * g[None, :]
    dm = ((x @ Wp.T) * dy).sum(axis=0) / n
    dB = scaling * gdy.T @ (x @ A.T)
    dA = scaling * (B.T @ gdy.T) @ x
    dx = gdy @ Wp
    return dA, dB, dm, dx
