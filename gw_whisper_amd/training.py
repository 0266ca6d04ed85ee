"""Autograd bridge of the DoRA training step.

``encoder_train_forward(encoder, mel)`` runs the HIP training forward (activations kept in an
arena) and returns ``last_hidden_state`` as a tensor that participates in torch autograd; its
backward calls ``gww_encoder_train_backward`` and hands the A / B / magnitude gradients of every
DoRA-wrapped q / k / v projection to autograd, so the reference's step

    loss = criterion(model(h1, l1), labels); loss.backward(); optimizer.step()
    (Signal_vs_Noise/src/train.py:163-168)

works with the MLP head, the loss and AdamW in plain torch on the GPU and everything inside the
encoder in libgww.  The DoRA weight norm is detached exactly like peft 0.12.0 ``dora.py``.
"""

from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from . import encoder as _encoder
from ._lib import check, lib
from .peft import DoraLinear

_PROJ = {"q_proj": 0, "k_proj": 1, "v_proj": 2, "out_proj": 3}


def dora_targets(encoder):
    """[(layer index, proj id, DoraLinear)] of the adapted attention projections."""
    out = []
    for li, layer in enumerate(encoder.layers):
        for name, pid in _PROJ.items():
            mod = getattr(layer.self_attn, name)
            if isinstance(mod, DoraLinear):
                out.append((li, pid, mod))   # use_dora=False (plain LoRA, the reference's --method LoRA) included
    return out


class _EncoderTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, encoder, mel, pooled, *params):
        enc = encoder
        c = enc.config
        x = mel.to(torch.float32).contiguous()
        B = x.shape[0]
        dev = x.device
        with torch.cuda.device(dev):
            enc._sync_weights()
            _encoder._note_training(enc)   # from now on the packed weights follow every optimizer step at once
            h = enc._ensure_handle()
            ws = torch.empty((lib().gww_train_workspace_bytes(h, B),), dtype=torch.uint8, device=dev)
            saved = torch.empty((lib().gww_train_saved_bytes(h, B),), dtype=torch.uint8, device=dev)
            shape = (B, c.d_model) if pooled else (B, c.max_source_positions, c.d_model)
            hidden = torch.empty(shape, dtype=torch.float32, device=dev)
            check(lib().gww_encoder_train_forward(h, x.data_ptr(), B, ws.data_ptr(), ws.numel(), saved.data_ptr(),
                                                  saved.numel(), hidden.data_ptr(), int(pooled),
                                                  torch.cuda.current_stream().cuda_stream),
                  "gww_encoder_train_forward")
        ctx.enc, ctx.B, ctx.ws, ctx.saved, ctx.pooled = enc, B, ws, saved, bool(pooled)
        ctx.n_params = len(params)
        ctx.mel_shape = tuple(x.shape)
        return hidden

    @staticmethod
    def backward(ctx, d_hidden):
        enc, B = ctx.enc, ctx.B
        dev = d_hidden.device
        d_hidden = d_hidden.to(torch.float32).contiguous()
        targets = dora_targets(enc)
        arr = (_lib.DoraTarget * max(len(targets), 1))()
        grads, keep = [], []
        def grad_buffer(param, like):
            # The HIP backward ACCUMULATES into the buffers it is given.  When the parameter already owns a dense fp32
            # .grad (optimizer.zero_grad(set_to_none=False), dist.FlatGradBucket views) it accumulates straight into
            # that and autograd gets None for this input -- no zeros_like + add kernel per tensor (96 tiny launches
            # per whisper-tiny step).  Otherwise: a fresh zero buffer handed back to autograd as usual.
            g = param.grad
            if (g is not None and g.dtype == torch.float32 and g.is_contiguous() and g.device == like.device
                    and g.shape == like.shape and not g.requires_grad):
                return g, None
            z = torch.zeros_like(like)
            return z, z
        for i, (li, pid, mod) in enumerate(targets):
            pA, pB = mod.lora_A[mod.adapter].weight, mod.lora_B[mod.adapter].weight
            A = pA.detach().float().contiguous()
            Bm = pB.detach().float().contiguous()
            (dA, rA), (dB, rB) = grad_buffer(pA, A), grad_buffer(pB, Bm)
            if mod.use_dora:
                pm = mod.lora_magnitude_vector[mod.adapter].weight
                mag = pm.detach().float().contiguous()
                nrm = mod._last_norm
                dm, rm = grad_buffer(pm, mag)
                grads.append((rA, rB, rm))
            else:
                # plain LoRA (peft tuners/lora/layer.py, use_dora=False: W' = W0 + s B A) is the DoRA gradient with the
                # row gain g = m / ||W'|| == 1: magnitude and norm are both ones, the magnitude gradient is discarded
                mag = nrm = torch.ones(mod.out_features, dtype=torch.float32, device=dev)
                dm = torch.zeros_like(mag)
                grads.append((rA, rB))
            keep += [A, Bm, mag, nrm, dA, dB, dm]
            arr[i] = _lib.DoraTarget(li, pid, mod.r, float(mod.scaling), A.data_ptr(), Bm.data_ptr(), mag.data_ptr(),
                                     nrm.data_ptr(), dA.data_ptr(), dB.data_ptr(), dm.data_ptr())
        # gradient w.r.t. the input features (conv stem backward) only when autograd asks for it
        d_mel = torch.empty(ctx.mel_shape, dtype=torch.float32, device=dev) if ctx.needs_input_grad[1] else None
        with torch.cuda.device(dev):
            check(lib().gww_encoder_train_backward(enc._ensure_handle(), B, ctx.ws.data_ptr(), ctx.ws.numel(),
                                                   ctx.saved.data_ptr(), ctx.saved.numel(), d_hidden.data_ptr(), arr,
                                                   len(targets), None, d_mel.data_ptr() if d_mel is not None else None,
                                                   int(ctx.pooled), torch.cuda.current_stream().cuda_stream),
                  "gww_encoder_train_backward")
        flat = []
        for g in grads:
            flat += list(g)
        assert len(flat) == ctx.n_params
        ctx.ws = ctx.saved = None
        return (None, d_mel, None, *flat)


def encoder_train_forward(encoder, mel: torch.Tensor, pooled: bool = False) -> torch.Tensor:
    """last_hidden_state [B, 1500, d] -- or, with ``pooled``, its last token [B, d] (what every classifier of the
    reference reads, ``Signal_vs_Noise/src/model.py:25-26``; the last layer's row-wise ops and their backward then run
    on B rows instead of B * 1500) -- with autograd through the DoRA parameters and, when ``mel.requires_grad``,
    through the conv stem to the input features."""
    if encoder.precision != "bf16":
        raise _lib.GwwError("the training step is implemented for precision='bf16'")
    params = []
    for _, _, mod in dora_targets(encoder):
        params += [mod.lora_A[mod.adapter].weight, mod.lora_B[mod.adapter].weight]
        if mod.use_dora:
            params.append(mod.lora_magnitude_vector[mod.adapter].weight)
    return _EncoderTrain.apply(encoder, mel, bool(pooled), *params)
