"""Whitening of the search pipeline's strain on the GPU.

Counterpart of ``whiten`` in the reference (``MLGWSC-1/inference.py:56-137``: PyCBC 2.4.0 ``TimeSeries.psd`` [Welch,
median] -> ``interpolate`` -> ``inverse_spectrum_truncation`` -> frequency-domain division -> drop the corrupted
edges), same signature and defaults, for ``[samples]`` or ``[detectors, samples]`` strain.

**Parity unpinned**: PyCBC is not installed in any environment this build can run in and the reference holds no
vector for this step; the arithmetic follows ``oracle/whiten.py`` (the restatement of the PyCBC routines with their
``delta_t`` / ``delta_f`` conventions) and is checked against it only.

Device work:

* Welch spectra: the half-overlapping segments times the hann-windowed real-DFT matrix -- one fp32-MFMA GEMM
  (``gww_gemm_f32``) -- then ``gww_welch_power_f32`` and the per-frequency median by radix select
  (``gww_column_median_f32``).  Whitening is invariant under a rescaling of the input, so the strain is first brought
  to unit scale by a power of two (raw 1e-21 strain squares to 1e-42, below fp32).
* Filter design (a 513-point PSD in, a short FIR out): PyCBC's own sequence -- interpolate to the segment's frequency
  resolution, ``q = ifft(1 / sqrt(psd))``, hann-tapered truncation to ``max_filter_duration``, ``|fft(q)|`` -- in fp64
  with the N-point transforms of ``torch.fft`` (rocFFT); the result is read off as the impulse response
  ``h = irfft(|fft(q)|)``, which is ``max_filter_len`` taps long up to the tail the modulus adds (kept until it is below
  1e-6 of the response; with a low-frequency cutoff the rectified stop-band ripple can make it too long for the FIR form
  (> 8191 taps): the response is then applied with the N-point transform, as the reference does).
* Filter application: ``gww_fir_f32`` -- the strain streams once through LDS against those taps (time domain, O(1)
  extra memory; circular padding reproduces the reference's circular convolution exactly) and the
  ``max_filter_len // 2`` corrupted samples on each side are never computed.
"""

from __future__ import annotations

import math

import numpy as np
import torch

from . import _lib
from ._lib import check, lib

_DFT = {}


def _welch_dft_matrix(seg_len: int, device) -> torch.Tensor:
    """[>= 2 (seg_len / 2 + 1) rounded up to 4, seg_len] fp32: rows (2 j, 2 j + 1) = hann[k] cos / -hann[k] sin(2 pi j k / L)."""
    key = (seg_len, str(device))
    if key not in _DFT:
        j = np.arange(seg_len // 2 + 1)[:, None].astype(np.float64)
        k = np.arange(seg_len)[None, :].astype(np.float64)
        ang = 2 * np.pi * ((j * k) % seg_len) / seg_len
        w = np.hanning(seg_len)[None, :]
        rows = 2 * (seg_len // 2 + 1)
        m = np.zeros(((rows + 3) // 4 * 4, seg_len), np.float64)
        m[0:rows:2] = np.cos(ang) * w
        m[1:rows:2] = -np.sin(ang) * w
        _DFT[key] = torch.from_numpy(m.astype(np.float32)).to(device)
    return _DFT[key]


def median_bias(n: int) -> float:
    """pycbc.psd.estimate.median_bias (host scalar)."""
    if n >= 1000:
        return math.log(2)
    ans = 1.0
    for i in range(1, int((n - 1) / 2 + 1)):
        ans += 1.0 / (2 * i + 1) - 1.0 / (2 * i)
    return ans


def welch_median_psd(x: torch.Tensor, delta_t: float, segment_duration: float):
    """``TimeSeries(x, delta_t).psd(segment_duration)`` for every row of x [D, N] (fp32, unit scale): (psd fp64 [D, L / 2 + 1],
    delta_f)."""
    from . import ops
    D, n = x.shape
    seg_len = int(round(segment_duration / delta_t))
    stride = int(seg_len / 2)
    n_seg = int(n // stride)
    if (n_seg - 1) * stride + seg_len > n:
        n_seg -= 1
    data_len = (n_seg - 1) * stride + seg_len
    if n_seg < 1 or data_len > n:
        raise ValueError("whiten: not enough data for one PSD segment")
    diff = n - data_len
    start = diff // 2 + (diff % 2)
    n_bins = seg_len // 2 + 1
    dft = _welch_dft_matrix(seg_len, x.device)
    delta_f = 1.0 / delta_t / seg_len
    wsum = float((np.hanning(seg_len) ** 2).sum())
    out = torch.empty((D, n_bins), dtype=torch.float64, device=x.device)
    stream = torch.cuda.current_stream().cuda_stream
    for d in range(D):
        frames = x[d, start:start + data_len].as_strided((n_seg, seg_len), (stride, 1)).contiguous()
        spec = ops.gemm(frames, dft, None, 0)                                     # [n_seg, 2 n_bins (+pad)]
        power = torch.empty((n_seg, n_bins), dtype=torch.float32, device=x.device)
        med = torch.empty((n_bins,), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            check(lib().gww_welch_power_f32(spec.data_ptr(), spec.shape[1], n_seg, n_bins, float(delta_t * delta_t),
                                            power.data_ptr(), stream), "gww_welch_power_f32")
            check(lib().gww_column_median_f32(power.data_ptr(), n_seg, n_bins, med.data_ptr(), stream),
                  "gww_column_median_f32")
        out[d] = med.double() / median_bias(n_seg) * (2 * delta_f * seg_len / wsum)
    return out, delta_f


K_MAX = 4095          # longest half-length of the time-domain form (gww_fir_f32 takes up to 16384 taps)


def design_filter(psd_w: torch.Tensor, delta_f_w: float, n: int, delta_t: float, max_filter_len: int,
                  low_frequency_cutoff=None, trunc_method="hann", tail: float = 1e-6):
    """PyCBC's ``interpolate`` + ``inverse_spectrum_truncation`` for every row of psd_w (fp64 [D, bins]) at the resolution of an
    n-sample segment; returns the correlation taps g fp32 [D, taps4] (zero padded to a multiple of 4) and K with
    ``white[i] = sum_u g[u] x[(i - K + u) mod n]`` -- or ``(None, wf)`` with the frequency response ``|fft(q)|`` when the
    response does not fit ``K_MAX`` taps to within ``tail`` of its l1 norm."""
    D, bins = psd_w.shape
    dev = psd_w.device
    delta_f = 1.0 / (n * delta_t)
    new_n = int(np.rint((bins - 1) * delta_f_w / delta_f + 1))
    if new_n != n // 2 + 1:
        raise ValueError("whiten: the segment length does not match the PSD resolution (even sample count required)")
    f = torch.arange(new_n, dtype=torch.float64, device=dev) * delta_f
    pos = torch.clamp(f / delta_f_w, max=bins - 1)
    j = torch.clamp(pos.floor().long(), max=bins - 2)
    xp = j.double() * delta_f_w
    psd = psd_w[:, j] + (f - xp) * ((psd_w[:, j + 1] - psd_w[:, j]) / delta_f_w)          # numpy.interp
    N = (new_n - 1) * 2
    kmin = int(low_frequency_cutoff / delta_f) if low_frequency_cutoff else 1
    inv_asd = torch.zeros((D, new_n), dtype=torch.complex128, device=dev)
    inv_asd[:, kmin:N // 2] = (1.0 / psd[:, kmin:N // 2]).sqrt().to(torch.complex128)
    q = torch.fft.irfft(inv_asd, n=N, dim=1) * (N * delta_f)
    t0, t1 = max_filter_len // 2, N - max_filter_len // 2
    if t1 < t0:
        raise ValueError("Invalid value in inverse_spectrum_truncation")
    if trunc_method == "hann":
        win = torch.from_numpy(np.hanning(max_filter_len)).to(dev)
        q[:, :t0] *= win[-t0:]
        q[:, t1:] *= win[:max_filter_len // 2]
    if t0 < t1:
        q[:, t0:t1] = 0
    wf = (torch.fft.rfft(q, dim=1) * delta_t).abs()                 # (1 / psd_out) ** 0.5 = |fft(q)|
    h = torch.fft.irfft(wf.to(torch.complex128), n=N, dim=1)        # real, even: white = h (*) x circularly
    # keep the central taps until the rest is negligible (max_filter_len // 2 each side + what the modulus spreads: with a
    # low-frequency cutoff the rectified stop-band ripple decays slowly).  No K <= K_MAX reaches `tail`: return the
    # frequency response instead and let the caller apply it with the N-point transform.
    K = max(t0, 64)
    total = h.abs().sum(dim=1)
    while True:
        rest = total - h[:, :K + 1].abs().sum(dim=1) - h[:, N - K:].abs().sum(dim=1)
        if float((rest / total).max()) < tail:
            break
        if 2 * K > K_MAX or 2 * K >= N // 2 - 1:
            return None, wf
        K *= 2
    taps = torch.cat((h[:, N - K:], h[:, :K + 1]), dim=1)            # h[-K .. K]
    g = taps.flip(1)                                                 # g[u] = h[K - u]
    taps4 = (g.shape[1] + 3) // 4 * 4
    gp = torch.zeros((D, taps4), dtype=torch.float32, device=dev)
    gp[:, :g.shape[1]] = g.float()
    return gp, K


def whiten(strain, delta_t: float = 1.0 / 2048.0, segment_duration: float = 0.5, max_filter_duration: float = 0.25,
           trunc_method="hann", remove_corrupted: bool = True, low_frequency_cutoff=None, psd=None,
           return_psd: bool = False, device="cuda"):
    """Reference ``whiten`` (``MLGWSC-1/inference.py:56-137``): 1-D or [detectors, samples] strain -> whitened strain
    (fp32, on the GPU), ``max_filter_duration / 2`` seconds shorter on each side unless ``remove_corrupted=False``."""
    if psd is not None:
        raise NotImplementedError("whiten: only psd=None (estimate from the data, as MLGWSC-1/inference.py:225 calls it)")
    x = strain if torch.is_tensor(strain) else torch.from_numpy(np.asarray(strain))
    one_d = x.dim() == 1
    if one_d:
        x = x[None]
    if x.dim() != 2:
        raise ValueError("Strain must be 1D or 2D.")
    x = x.to(device)
    if not x.is_cuda:
        raise _lib.GwwError("whiten needs a GPU: gw_whisper_amd has no CPU path")
    D, n = x.shape
    if n % 2:
        raise ValueError("whiten: an even number of samples is required")
    # unit scale by an exact power of two (whitening is scale invariant; fp32 cannot hold the square of 1e-21)
    peak = float(x.abs().max())
    scale = 2.0 ** (-math.floor(math.log2(peak))) if peak > 0 and math.isfinite(peak) else 1.0
    xs = (x.double() * scale).float().contiguous()
    psd_w, df_w = welch_median_psd(xs, delta_t, segment_duration)
    max_filter_len = int(max_filter_duration * (1.0 / delta_t))
    g, K = design_filter(psd_w, df_w, n, delta_t, max_filter_len, low_frequency_cutoff, trunc_method)
    cut = max_filter_len // 2 if remove_corrupted else 0
    n_out = n - 2 * cut
    if g is None:
        # long response (low-frequency cutoff): the reference's own form, irfft(rfft(x) |fft(q)|), in fp64 (rocFFT)
        white = torch.fft.irfft(torch.fft.rfft(xs.double(), dim=1) * K, n=n, dim=1)[:, cut:n - cut].float().contiguous()
        if one_d:
            white = white[0]
        if return_psd:
            p = psd_w / (scale * scale)
            return white, (p[0] if one_d else p)
        return white
    # circular padding: output i (= sample cut + i) needs x[(cut + i - K + u) mod n], u = 0 .. 2 K
    idx = (torch.arange(n_out + g.shape[1], device=x.device) + (cut - K)) % n
    pad = (n_out + g.shape[1] + 3) // 4 * 4
    xp = torch.zeros((D, pad), dtype=torch.float32, device=x.device)
    xp[:, :idx.numel()] = xs[:, idx]
    out_stride = (n_out + 3) // 4 * 4
    out = torch.empty((D, out_stride), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        check(lib().gww_fir_f32(xp.data_ptr(), pad, g.data_ptr(), g.shape[1], D, out.data_ptr(), out_stride, n_out,
                                torch.cuda.current_stream().cuda_stream), "gww_fir_f32")
    white = out[:, :n_out]
    if one_d:
        white = white[0]
    if return_psd:
        # the estimate in the input's units (the scaling above squared out again)
        p = psd_w / (scale * scale)
        return white, (p[0] if one_d else p)
    return white
