"""Tensor-level wrappers around the C ABI (device pointers + the current HIP stream).

PyTorch is used here only for device memory and streams.  Every function
requires CUDA(HIP) tensors and calls into libgww.so; none has a torch fallback.
"""

from __future__ import annotations

import torch

from . import _lib
from ._lib import check, lib


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _dev(t: torch.Tensor, dtype=None, name="tensor") -> torch.Tensor:
    if not t.is_cuda:
        raise _lib.GwwError(f"{name} must live on the GPU (got {t.device}); gw_whisper_amd has no CPU path")
    if dtype is not None and t.dtype != dtype:
        raise _lib.GwwError(f"{name} must be {dtype}, got {t.dtype}")
    return t.contiguous()


_frontend = {}


def _fe(device) -> int:
    idx = torch.device(device).index or 0
    if idx not in _frontend:
        import ctypes as C
        h = C.c_void_p()
        with torch.cuda.device(idx):
            check(lib().gww_frontend_create(C.byref(h)), "gww_frontend_create")
        _frontend[idx] = h
    return _frontend[idx]


def logmel(wave: torch.Tensor, n_samples: int | None = None) -> torch.Tensor:
    """[n, L] fp32 GPU waveform (16 kHz) -> [n, 80, 3000] fp32 ``input_features``.

    Same arithmetic as ``WhisperFeatureExtractor(...)`` (reference call site
    Signal_vs_Noise/src/dataset.py:20-21): zero-pad/truncate to 30 s, STFT, mel, log10,
    per-segment dynamic-range clamp, affine.
    """
    if wave.dim() == 1:
        wave = wave[None]
    wave = _dev(wave, torch.float32, "wave")
    n, L = wave.shape
    n_samples = L if n_samples is None else n_samples
    out = torch.empty((n, 80, 3000), dtype=torch.float32, device=wave.device)
    seg_max = torch.empty((max(n, 1),), dtype=torch.float32, device=wave.device)
    with torch.cuda.device(wave.device):
        step = 32768   # gridDim.y limit
        for i in range(0, n, step):
            m = min(step, n - i)
            check(lib().gww_logmel_f32(_fe(wave.device), wave[i:].data_ptr(), m, n_samples, wave.stride(0),
                                       out[i:].data_ptr(), seg_max[i:].data_ptr(), _stream()), "gww_logmel_f32")
    return out


def logmel_host(wave, n_samples: int | None = None) -> torch.Tensor:
    """CPU twin of :func:`logmel`: [n, L] fp32 host waveform -> [n, 80, 3000] fp32 CPU tensor.

    Runs ``gww_logmel_host_f32`` (plain C++ inside libgww.so, no HIP call), so it works inside forked DataLoader
    workers, where the reference calls the extractor (Signal_vs_Noise/src/dataset.py:20-21 under
    src/train.py:224-225).  Not a fallback of the GPU path: callers pick it by handing over host data.
    """
    import numpy as np
    w = np.ascontiguousarray(wave.numpy() if isinstance(wave, torch.Tensor) else wave, dtype=np.float32)
    if w.ndim == 1:
        w = w[None]
    n, L = w.shape
    n_samples = L if n_samples is None else n_samples
    out = torch.empty((n, 80, 3000), dtype=torch.float32)
    check(lib().gww_logmel_host_f32(w.ctypes.data, n, n_samples, max(L, 1) if n else 1, out.data_ptr()),
          "gww_logmel_host_f32")
    return out


def layernorm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, out_bf16: bool = False) -> torch.Tensor:
    x = _dev(x, torch.float32, "x")
    M, d = x.shape
    y = torch.empty((M, d), dtype=torch.bfloat16 if out_bf16 else torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        check(lib().gww_layernorm(x.data_ptr(), _dev(w, torch.float32).data_ptr(), _dev(b, torch.float32).data_ptr(),
                                  y.data_ptr(), int(out_bf16), M, d, _stream()), "gww_layernorm")
    return y


def cast_bf16(x: torch.Tensor) -> torch.Tensor:
    x = _dev(x, torch.float32, "x")
    y = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    with torch.cuda.device(x.device):
        check(lib().gww_cast_f32_bf16(x.data_ptr(), y.data_ptr(), x.numel(), _stream()), "gww_cast_f32_bf16")
    return y


def gemm(a: torch.Tensor, w: torch.Tensor, bias: torch.Tensor | None = None, epilogue: int = 0,
         resid: torch.Tensor | None = None) -> torch.Tensor:
    """C = A[M,K] @ W[N,K]^T + bias with epilogue 0 none / 1 GELU / 2 += resid.

    bf16 A/W -> bf16 C (fp32 C for the residual epilogue); fp32 A/W -> fp32 C.
    """
    bf = a.dtype == torch.bfloat16
    a = _dev(a, torch.bfloat16 if bf else torch.float32, "A")
    w = _dev(w, a.dtype, "W")
    M, K = a.shape
    N = w.shape[0]
    if w.shape[1] != K:
        raise _lib.GwwError(f"gemm: A is [{M},{K}] but W is {tuple(w.shape)}")
    out_dtype = torch.float32 if (not bf or epilogue == _lib.EPI_RESID) else torch.bfloat16
    c = torch.empty((M, N), dtype=out_dtype, device=a.device)
    bptr = _dev(bias, torch.float32, "bias").data_ptr() if bias is not None else None
    rptr = _dev(resid, torch.float32, "resid").data_ptr() if resid is not None else None
    fn = lib().gww_gemm_bf16 if bf else lib().gww_gemm_f32
    with torch.cuda.device(a.device):
        check(fn(a.data_ptr(), w.data_ptr(), bptr, rptr, c.data_ptr(), M, N, K, epilogue, _stream()), "gww_gemm")
    return c


def gemm_v4_split(a: torch.Tensor, w: torch.Tensor, bias: torch.Tensor | None = None, epilogue: int = 0,
                  resid: torch.Tensor | None = None, n_split: int = 0) -> torch.Tensor:
    """``gemm`` on the 256 x 256 x 64 kernel of the wide encoders with an explicit column split of its work items
    (``gww_gemm_bf16_v4_split``; 0 = the automatic choice).  bf16 A [M % 256 == 0, K] / W [N, K]."""
    a = _dev(a, torch.bfloat16, "A")
    w = _dev(w, torch.bfloat16, "W")
    M, K = a.shape
    N = w.shape[0]
    if w.shape[1] != K:
        raise _lib.GwwError(f"gemm_v4_split: A is [{M},{K}] but W is {tuple(w.shape)}")
    c = torch.empty((M, N), dtype=torch.float32 if epilogue == _lib.EPI_RESID else torch.bfloat16, device=a.device)
    bptr = _dev(bias, torch.float32, "bias").data_ptr() if bias is not None else None
    rptr = _dev(resid, torch.float32, "resid").data_ptr() if resid is not None else None
    with torch.cuda.device(a.device):
        check(lib().gww_gemm_bf16_v4_split(a.data_ptr(), w.data_ptr(), bptr, rptr, c.data_ptr(), M, N, K, epilogue,
                                           int(n_split), _stream()), "gww_gemm_bf16_v4_split")
    return c


def ln_fold_weights(w: torch.Tensor, ln_w: torch.Tensor, ln_b: torch.Tensor, bias=None, scale: float = 1.0):
    """Fold LayerNorm(gain ln_w, shift ln_b) into the Linear (w fp32 [N,K], bias) that follows it.
    Returns (w_folded bf16 [N,K], u fp32 [N], cb fp32 [N]) for ``gemm_astat(..., ln=(u, cb))``."""
    w = _dev(w, torch.float32, "w")
    N, K = w.shape
    wf = torch.empty((N, K), dtype=torch.bfloat16, device=w.device)
    u = torch.empty((N,), dtype=torch.float32, device=w.device)
    cb = torch.empty((N,), dtype=torch.float32, device=w.device)
    with torch.cuda.device(w.device):
        check(lib().gww_ln_fold_weights(w.data_ptr(), _dev(ln_w, torch.float32).data_ptr(),
                                        _dev(ln_b, torch.float32).data_ptr(),
                                        _dev(bias, torch.float32).data_ptr() if bias is not None else None,
                                        float(scale), N, K, wf.data_ptr(), u.data_ptr(), cb.data_ptr(), _stream()),
              "gww_ln_fold_weights")
    return wf, u, cb


def gemm_astat(a: torch.Tensor, w: torch.Tensor, bias=None, epilogue: int = 0, ln=None, delta=None,
               return_x: bool = False):
    """A-stationary bf16 GEMM (K in {256,384,512}) -> bf16 [M, N].

    ``ln=(u, cb)`` (from ``ln_fold_weights``, with ``w`` the folded panel): ``a`` is the fp32
    residual stream, ``x_new = a + delta`` (bf16 ``delta`` optional) and LayerNorm are fused
    into the GEMM in one pass; ``return_x`` also returns ``x_new``.  Output rows are padded
    to a multiple of 256 internally."""
    fused = ln is not None
    a = _dev(a, torch.float32 if fused else torch.bfloat16, "A")
    w = _dev(w, torch.bfloat16, "W")
    M, K = a.shape
    N = w.shape[0]
    Mp = (M + 255) // 256 * 256
    c = torch.empty((Mp, N), dtype=torch.bfloat16, device=a.device)
    dl = _dev(delta, torch.bfloat16, "delta") if delta is not None else None
    # x_new is only materialised when there is a delta to add; otherwise x_new IS a
    x_out = torch.empty_like(a) if (fused and return_x and dl is not None) else None
    with torch.cuda.device(a.device):
        check(lib().gww_gemm_astat_bf16(
            a.data_ptr(), dl.data_ptr() if (dl is not None and x_out is not None) else None,
            x_out.data_ptr() if x_out is not None else None,
            _dev(ln[0], torch.float32).data_ptr() if fused else None,
            _dev(ln[1], torch.float32).data_ptr() if fused else None, w.data_ptr(),
            _dev(bias, torch.float32).data_ptr() if bias is not None else None, c.data_ptr(), M, N, K,
            epilogue, _stream()), "gww_gemm_astat_bf16")
    return (c[:M], x_out if x_out is not None else a) if return_x else c[:M]


def gemm_fulln(a: torch.Tensor, w: torch.Tensor, bias=None, epilogue: int = 0) -> torch.Tensor:
    """Full-N bf16 GEMM (N in {384, 512}, K % 32 == 0): complete output rows per workgroup."""
    a = _dev(a, torch.bfloat16, "A")
    w = _dev(w, torch.bfloat16, "W")
    M, K = a.shape
    N = w.shape[0]
    Mp = (M + 127) // 128 * 128
    c = torch.empty((Mp, N), dtype=torch.bfloat16, device=a.device)
    with torch.cuda.device(a.device):
        check(lib().gww_gemm_fulln_bf16(a.data_ptr(), w.data_ptr(),
                                        _dev(bias, torch.float32).data_ptr() if bias is not None else None,
                                        c.data_ptr(), M, N, K, epilogue, _stream()), "gww_gemm_fulln_bf16")
    return c[:M]


def mlp_pack(w1_folded: torch.Tensor, w2: torch.Tensor, wqkv_folded: torch.Tensor | None = None) -> torch.Tensor:
    """Weight stream of mlp_fused: folded fc1 panel [F, 384] + fc2 panel [384, F] (+ the next layer's folded
    q / k / v panel [NQ, 384]) -> bf16 [2 * 384 * F (+ NQ * 384)]."""
    wq = _dev(wqkv_folded, torch.bfloat16, "Wqkv") if wqkv_folded is not None else None
    if w1_folded is None:            # only the q / k / v panel: the stream of lnqkv_fused
        if wq is None:
            raise _lib.GwwError("mlp_pack: nothing to pack")
        w1 = w2 = None
        F, d = 0, wq.shape[1]
    else:
        w1 = _dev(w1_folded, torch.bfloat16, "W1")
        w2 = _dev(w2, torch.bfloat16, "W2")
        F, d = w1.shape
    NQ = wq.shape[0] if wq is not None else 0
    dev = wq.device if w1 is None else w1.device
    out = torch.empty((2 * d * F + NQ * d,), dtype=torch.bfloat16, device=dev)
    with torch.cuda.device(dev):
        check(lib().gww_mlp_pack_bf16(w1.data_ptr() if w1 is not None else None, w2.data_ptr() if w2 is not None else None,
                                      wq.data_ptr() if wq is not None else None, out.data_ptr(), d, F, NQ, _stream()),
              "gww_mlp_pack_bf16")
    return out


def attn_out_mlp_fused(x, ctx, wo, bo, w1_folded, w2, ln_u, ln_cb, b2, qkv=None):
    """The attention output projection fused in front of ``mlp_fused``: x_new = x + bf16(ctx Wo^T + bo), then the block.
    ``qkv=(wqkv_folded, u, cb)`` appends the next layer's LN1 + q / k / v.  Returns (C or qkv, x_new or x_next)."""
    x = _dev(x, torch.float32, "x")
    ctx = _dev(ctx, torch.bfloat16, "ctx")
    wo = _dev(wo, torch.bfloat16, "Wo")
    w1 = _dev(w1_folded, torch.bfloat16, "W1")
    w2 = _dev(w2, torch.bfloat16, "W2")
    wq = _dev(qkv[0], torch.bfloat16, "Wqkv") if qkv is not None else None
    F, d = w1.shape
    NQ = wq.shape[0] if wq is not None else 0
    M = x.shape[0]
    Mp = (M + 127) // 128 * 128
    f = lambda t: _dev(t, torch.float32)
    bo, u, cb, b2 = f(bo), f(ln_u), f(ln_cb), f(b2)
    wt = torch.empty((d * d + 2 * d * F + NQ * d,), dtype=torch.bfloat16, device=x.device)
    if qkv is not None:
        x = x.clone()      # the library writes x_next back over x (include/gww.h): the caller's tensor stays untouched
    x_out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        check(lib().gww_mlp_pack_op_bf16(wo.data_ptr(), w1.data_ptr(), w2.data_ptr(), wq.data_ptr() if wq is not None else None,
                                         wt.data_ptr(), d, F, NQ, _stream()), "gww_mlp_pack_op_bf16")
        if qkv is None:
            out = torch.empty((Mp, d), dtype=torch.bfloat16, device=x.device)
            qu = qc = None
        else:
            qu, qc = f(qkv[1]), f(qkv[2])
            out = torch.empty((Mp, NQ), dtype=torch.bfloat16, device=x.device)
        check(lib().gww_attn_out_mlp_fused_bf16(x.data_ptr(), ctx.data_ptr(), bo.data_ptr(), x_out.data_ptr(), u.data_ptr(),
                                                cb.data_ptr(), wt.data_ptr(), b2.data_ptr(),
                                                out.data_ptr() if qkv is None else None, M, d, F,
                                                qu.data_ptr() if qu is not None else None,
                                                qc.data_ptr() if qc is not None else None,
                                                out.data_ptr() if qkv is not None else None, NQ, _stream()),
              "gww_attn_out_mlp_fused_bf16")
    return out[:M], (x_out if qkv is None else x)


def attn_out_mlp_final(x, ctx, wo, bo, w1_folded, w2, ln_u, ln_cb, b2, lnf_w, lnf_b):
    """The LAST block with the encoder's final LayerNorm as its epilogue (``gww_attn_out_mlp_final_bf16``): returns
    (y fp32 [M, 384] = LayerNorm_final(x_mid + bf16(mlp(LayerNorm2(x_mid)) + b2)), x_mid = x + bf16(ctx Wo^T + bo))."""
    x = _dev(x, torch.float32, "x")
    ctx = _dev(ctx, torch.bfloat16, "ctx")
    wo, w1, w2 = _dev(wo, torch.bfloat16, "Wo"), _dev(w1_folded, torch.bfloat16, "W1"), _dev(w2, torch.bfloat16, "W2")
    F, d = w1.shape
    M = x.shape[0]
    f = lambda t: _dev(t, torch.float32)
    bo, u, cb, b2, lw, lb = f(bo), f(ln_u), f(ln_cb), f(b2), f(lnf_w), f(lnf_b)
    wt = torch.empty((d * d + 2 * d * F,), dtype=torch.bfloat16, device=x.device)
    x_mid, y = torch.empty_like(x), torch.empty_like(x)
    with torch.cuda.device(x.device):
        check(lib().gww_mlp_pack_op_bf16(wo.data_ptr(), w1.data_ptr(), w2.data_ptr(), None, wt.data_ptr(), d, F, 0, _stream()),
              "gww_mlp_pack_op_bf16")
        check(lib().gww_attn_out_mlp_final_bf16(x.data_ptr(), ctx.data_ptr(), bo.data_ptr(), x_mid.data_ptr(), u.data_ptr(),
                                                cb.data_ptr(), wt.data_ptr(), b2.data_ptr(), lw.data_ptr(), lb.data_ptr(),
                                                y.data_ptr(), M, d, F, _stream()), "gww_attn_out_mlp_final_bf16")
    return y, x_mid


def lnqkv_fused(x, wt, qkv_u, qkv_cb):
    """qkv bf16 [M, NQ] = LayerNorm(x) Wqkv'^T + cb for a residual stream without a pending delta (layer 0): the panel
    prologue and the q / k / v tail of the fused MLP kernel; ``wt = mlp_pack(None, None, wqkv_folded)``."""
    x = _dev(x, torch.float32, "x")
    wt = _dev(wt, torch.bfloat16, "Wt")
    M, d = x.shape
    qu, qc = _dev(qkv_u, torch.float32), _dev(qkv_cb, torch.float32)
    NQ = qu.numel()
    qo = torch.empty(((M + 127) // 128 * 128, NQ), dtype=torch.bfloat16, device=x.device)
    with torch.cuda.device(x.device):
        check(lib().gww_lnqkv_fused_bf16(x.data_ptr(), qu.data_ptr(), qc.data_ptr(), wt.data_ptr(), qo.data_ptr(), M, d, NQ,
                                         _stream()), "gww_lnqkv_fused_bf16")
    return qo[:M]


def mlp_fused(x, delta, wt, ln_u, ln_cb, b2, qkv=None):
    """(C bf16 [M, 384], x_new fp32 [M, 384]) = fused LayerNorm -> fc1 -> GELU -> fc2 of x + delta.
    ``qkv=(u, cb)`` of the next layer's folded q / k / v projection (panel appended to ``wt``): returns
    (qkv bf16 [M, NQ], x_next fp32 [M, 384]) instead, x_next = x + delta + bf16(mlp output)."""
    x = _dev(x, torch.float32, "x")
    delta = _dev(delta, torch.bfloat16, "delta")
    wt = _dev(wt, torch.bfloat16, "Wt")
    M, d = x.shape
    F = ln_u.numel()
    Mp = (M + 127) // 128 * 128
    if qkv is not None:
        x = x.clone()      # the library writes x_next back over x (include/gww.h): the caller's tensor stays untouched
    x_out = torch.empty_like(x)
    f = lambda t: _dev(t, torch.float32)
    u, cb, b2 = f(ln_u), f(ln_cb), f(b2)
    if qkv is None:
        c = torch.empty((Mp, d), dtype=torch.bfloat16, device=x.device)
        qu = qc = qo = None
        NQ = 0
    else:
        qu, qc = f(qkv[0]), f(qkv[1])
        NQ = qu.numel()
        qo = torch.empty((Mp, NQ), dtype=torch.bfloat16, device=x.device)
        c = None
    with torch.cuda.device(x.device):
        check(lib().gww_mlp_fused_bf16(x.data_ptr(), delta.data_ptr(), x_out.data_ptr(), u.data_ptr(), cb.data_ptr(),
                                       wt.data_ptr(), b2.data_ptr(), c.data_ptr() if c is not None else None, M, d, F,
                                       qu.data_ptr() if qu is not None else None,
                                       qc.data_ptr() if qc is not None else None,
                                       qo.data_ptr() if qo is not None else None, NQ, _stream()),
              "gww_mlp_fused_bf16")
    return ((c if qkv is None else qo)[:M], x_out if qkv is None else x)


def attention(qkv: torch.Tensor, n_heads: int) -> torch.Tensor:
    """qkv [B, T, 3 d] (q pre-scaled) -> ctx [B, T, d]; bf16 or fp32."""
    bf = qkv.dtype == torch.bfloat16
    qkv = _dev(qkv, torch.bfloat16 if bf else torch.float32, "qkv")
    B, T, d3 = qkv.shape
    d = d3 // 3
    if d != n_heads * 64:
        raise _lib.GwwError(f"attention: d={d} != n_heads*64")
    ctx = torch.empty((B, T, d), dtype=qkv.dtype, device=qkv.device)
    fn = lib().gww_attention_bf16 if bf else lib().gww_attention_f32
    with torch.cuda.device(qkv.device):
        check(fn(qkv.data_ptr(), ctx.data_ptr(), B, T, n_heads, _stream()), "gww_attention")
    return ctx


def attention_log2q(qkv: torch.Tensor, n_heads: int, want_lse: bool = False):
    """The attention kernel of the encoder's bf16 paths: like :func:`attention`, but the q section is in log2 units
    (projected with log2(e) / 8 instead of 1 / 8).  Returns ctx, or (ctx, lse [B, H, T] natural log)."""
    qkv = _dev(qkv, torch.bfloat16, "qkv")
    B, T, d3 = qkv.shape
    if d3 != 3 * n_heads * 64:
        raise _lib.GwwError(f"attention: d={d3 // 3} != n_heads*64")
    ctx = torch.empty((B, T, d3 // 3), dtype=torch.bfloat16, device=qkv.device)
    lse = torch.empty((B, n_heads, T), dtype=torch.float32, device=qkv.device) if want_lse else None
    with torch.cuda.device(qkv.device):
        check(lib().gww_attention_log2q_bf16(qkv.data_ptr(), ctx.data_ptr(), lse.data_ptr() if want_lse else None, B, T,
                                             n_heads, _stream()), "gww_attention_log2q_bf16")
    return (ctx, lse) if want_lse else ctx


def dora_merge(w0: torch.Tensor, a: torch.Tensor, b: torch.Tensor, m: torch.Tensor, scaling: float,
               return_norm: bool = False):
    """W_eff = (m / ||W0 + s B A||_row) * (W0 + s B A)  (peft 0.12.0 dora.py), fp32."""
    w0 = _dev(w0, torch.float32, "w0")
    d_out, d_in = w0.shape
    r = a.shape[0]
    out = torch.empty_like(w0)
    nrm = torch.empty((d_out,), dtype=torch.float32, device=w0.device)
    with torch.cuda.device(w0.device):
        check(lib().gww_dora_merge_f32(w0.data_ptr(), _dev(a, torch.float32).data_ptr(),
                                       _dev(b, torch.float32).data_ptr(), _dev(m, torch.float32).data_ptr(),
                                       float(scaling), d_out, d_in, r, out.data_ptr(), nrm.data_ptr(), _stream()),
              "gww_dora_merge_f32")
    return (out, nrm) if return_norm else out


def conv1_gelu(mel: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """``gelu(conv1(input_features))`` of the Whisper stem (HF:modeling_whisper.py:619-620) read straight from the
    ``[B, 80, T]`` feature layout -> token-major bf16 ``[B, T + 2, d]`` with zero rows 0 and T + 1 (conv2's padding)."""
    mel = _dev(mel, torch.float32, "mel")
    weight, bias = _dev(weight, torch.float32, "weight"), _dev(bias, torch.float32, "bias")
    B, C_, T = mel.shape
    d = weight.shape[0]
    if C_ != 80 or tuple(weight.shape) != (d, 80, 3) or tuple(bias.shape) != (d,):
        raise _lib.GwwError(f"conv1_gelu: mel {tuple(mel.shape)}, weight {tuple(weight.shape)}, bias {tuple(bias.shape)}")
    scratch = torch.empty((d, 256), dtype=torch.bfloat16, device=mel.device)
    out = torch.empty((B, T + 2, d), dtype=torch.bfloat16, device=mel.device)
    with torch.cuda.device(mel.device):
        check(lib().gww_conv1_gelu_bf16(mel.data_ptr(), weight.data_ptr(), bias.data_ptr(), scratch.data_ptr(), out.data_ptr(),
                                        B, T, d, _stream()), "gww_conv1_gelu_bf16")
    return out


def dora_merge_batch(items, return_norm: bool = True):
    """``dora_merge`` for a list of ``(w0, a, b, m, scaling)`` tuples in ONE launch (an optimizer step changes every
    adapted projection at once; a launch per 384 x 384 matrix is launch-bound).  Returns a list of ``(w_eff, norm)``."""
    if not items:
        return []
    dev = items[0][0].device
    arr = (_lib.DoraMergeItem * len(items))()
    out, keep = [], []
    for i, (w0, a, b, m, scaling) in enumerate(items):
        w0, a, b, m = (_dev(t, torch.float32) for t in (w0, a, b, m))
        d_out, d_in = w0.shape
        r = a.shape[0]
        if a.shape != (r, d_in) or b.shape != (d_out, r) or m.shape != (d_out,) or w0.device != dev:
            raise _lib.GwwError(f"dora_merge_batch: item {i}: shapes {tuple(w0.shape)} {tuple(a.shape)} {tuple(b.shape)} "
                                f"{tuple(m.shape)} or device do not fit")
        w = torch.empty_like(w0)
        nrm = torch.empty((d_out,), dtype=torch.float32, device=dev)
        keep += [w0, a, b, m]
        arr[i] = _lib.DoraMergeItem(w0.data_ptr(), a.data_ptr(), b.data_ptr(), m.data_ptr(), w.data_ptr(), nrm.data_ptr(),
                                    float(scaling), d_out, d_in, r)
        out.append((w, nrm))
    with torch.cuda.device(dev):
        check(lib().gww_dora_merge_batch_f32(arr, len(items), _stream()), "gww_dora_merge_batch_f32")
    return out if return_norm else [w for w, _ in out]


# ------------------------------------------------------------------ training-step kernels
def attention_lse(qkv: torch.Tensor, n_heads: int):
    """bf16 attention forward that also returns the row log-sum-exp [B, H, T] (fp32)."""
    qkv = _dev(qkv, torch.bfloat16, "qkv")
    B, T, d3 = qkv.shape
    ctx = torch.empty((B, T, d3 // 3), dtype=torch.bfloat16, device=qkv.device)
    lse = torch.empty((B, n_heads, T), dtype=torch.float32, device=qkv.device)
    with torch.cuda.device(qkv.device):
        check(lib().gww_attention_lse_bf16(qkv.data_ptr(), ctx.data_ptr(), lse.data_ptr(), B, T, n_heads, _stream()),
              "gww_attention_lse_bf16")
    return ctx, lse


def attention_bwd(qkv, ctx, dctx, lse, n_heads: int, q_log2: bool = False) -> torch.Tensor:
    """dqkv [B, T, 3 d] (bf16) of softmax(q k^T) v given dctx; ``q_log2``: the q section is in log2 units and its
    gradient is taken with respect to that stored q."""
    qkv, ctx, dctx = (_dev(t, torch.bfloat16) for t in (qkv, ctx, dctx))
    lse = _dev(lse, torch.float32, "lse")
    B, T, d3 = qkv.shape
    dqkv = torch.empty_like(qkv)
    scratch = torch.empty(B * n_heads * (T + (T + 63) // 64), dtype=torch.float32, device=qkv.device)
    with torch.cuda.device(qkv.device):
        fn = lib().gww_attention_bwd_log2q_bf16 if q_log2 else lib().gww_attention_bwd_bf16
        check(fn(qkv.data_ptr(), ctx.data_ptr(), dctx.data_ptr(), lse.data_ptr(), scratch.data_ptr(), dqkv.data_ptr(),
                 B, T, n_heads, _stream()), "gww_attention_bwd_bf16")
    return dqkv


def layernorm_bwd(x, gamma, dy, dx=None, want_bf16: bool = False):
    """dx (+)= LayerNorm'(x)^T dy; dy fp32 or bf16.  Returns (dx fp32, dx bf16 | None)."""
    x = _dev(x, torch.float32, "x")
    M, d = x.shape
    dy = _dev(dy)
    acc = dx is not None
    if dx is None:
        dx = torch.empty_like(x)
    dxb = torch.empty((M, d), dtype=torch.bfloat16, device=x.device) if want_bf16 else None
    with torch.cuda.device(x.device):
        check(lib().gww_layernorm_bwd(x.data_ptr(), _dev(gamma, torch.float32).data_ptr(), dy.data_ptr(),
                                      int(dy.dtype == torch.float32), dx.data_ptr(), int(acc),
                                      dxb.data_ptr() if dxb is not None else None, M, d, _stream()), "gww_layernorm_bwd")
    return dx, dxb


def gelu_bf16(z, dgelu=None):
    z = _dev(z, torch.bfloat16, "z")
    out = torch.empty_like(z)
    with torch.cuda.device(z.device):
        check(lib().gww_gelu_bf16(z.data_ptr(), _dev(dgelu, torch.bfloat16).data_ptr() if dgelu is not None else None,
                                  out.data_ptr(), z.numel(), _stream()), "gww_gelu_bf16")
    return out


def dora_grads_multi(x, dy, y, col_off, bias_st, yscale, scaling, A, B, mag, nrm):
    """[(dA, dB, dm)] of up to three DoRA projections that read the same x (q, k, v of a layer) in one pass:
    x bf16 [M, d]; dy, y bf16 [M, W] with projection p in columns col_off[p] .. col_off[p] + d; the other arguments
    are per-projection lists.  d in {384, 512}, r = 8."""
    import ctypes as C
    x, dy, y = (_dev(t, torch.bfloat16) for t in (x, dy, y))
    M, d = x.shape
    n = len(col_off)
    keep = [[_dev(t, torch.float32) for t in lst] for lst in (bias_st, A, B, mag, nrm)]
    out = [(torch.zeros((8, d), dtype=torch.float32, device=x.device),
            torch.zeros((d, 8), dtype=torch.float32, device=x.device),
            torch.zeros((d,), dtype=torch.float32, device=x.device)) for _ in range(n)]
    ptrs = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
    with torch.cuda.device(x.device):
        check(lib().gww_dora_grads_multi(x.data_ptr(), x.stride(0), dy.data_ptr(), y.data_ptr(), dy.stride(0), n,
                                         (C.c_long * n)(*col_off), ptrs(keep[0]), (C.c_float * n)(*yscale),
                                         (C.c_float * n)(*scaling), ptrs(keep[1]), ptrs(keep[2]), ptrs(keep[3]),
                                         ptrs(keep[4]), ptrs([o[0] for o in out]), ptrs([o[1] for o in out]),
                                         ptrs([o[2] for o in out]), M, d, _stream()), "gww_dora_grads_multi")
    return out


def dora_grads(x, dy, y, bias_st, yscale, scaling, A, B, mag, nrm):
    """(dA [r,d], dB [d,r], dm [d]) of one DoRA projection; x, dy, y bf16 [M, d]."""
    x, dy, y = (_dev(t, torch.bfloat16) for t in (x, dy, y))
    M, d = x.shape
    r = A.shape[0]
    dA = torch.zeros((r, d), dtype=torch.float32, device=x.device)
    dB = torch.zeros((d, r), dtype=torch.float32, device=x.device)
    dm = torch.zeros((d,), dtype=torch.float32, device=x.device)
    f = lambda t: _dev(t, torch.float32).data_ptr()
    with torch.cuda.device(x.device):
        check(lib().gww_dora_grads(x.data_ptr(), d, dy.data_ptr(), y.data_ptr(), d, f(bias_st), float(yscale),
                                   float(scaling), f(A), f(B), f(mag), f(nrm), dA.data_ptr(), dB.data_ptr(),
                                   dm.data_ptr(), M, d, r, _stream()), "gww_dora_grads")
    return dA, dB, dm
