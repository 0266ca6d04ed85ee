"""Whitening of the search pipeline -- CPU restatement (test infrastructure only).  **PARITY UNPINNED.**

The reference whitens every HDF5 segment before slicing it (``MLGWSC-1/inference.py:56-137``, called from
``SegmentSlicer.process`` ``:218-246``) with PyCBC (pinned ``PyCBC==2.4.0``, ``requirements.txt:204``; NOT installed in
any environment this build can run in, and the reference holds no test or vector for this step).  What follows
restates the published PyCBC 2.4.0 routines the reference calls, with their normalisation conventions:

  * ``TimeSeries.psd(segment_duration)``  -> ``pycbc.psd.welch(ts, seg_len = round(dur * fs), seg_stride = seg_len // 2,
    window='hann', avg_method='median')``: the series is trimmed symmetrically to a whole number of half-overlapping
    segments, every segment is multiplied by ``numpy.hanning(seg_len)``, transformed with PyCBC's ``fft`` (= ``rfft`` x
    ``delta_t``), squared, DC and Nyquist halved; the per-frequency MEDIAN over the segments is divided by
    ``median_bias(n)`` and scaled by ``2 delta_f seg_len / sum(w^2)``.
  * ``pycbc.psd.interpolate(psd, delta_f)``: ``numpy.interp`` onto ``arange(rint((len - 1) df_old / df_new + 1)) * df_new``.
  * ``pycbc.psd.inverse_spectrum_truncation(psd, max_filter_len, low_frequency_cutoff, trunc_method='hann')``:
    ``q = ifft(1 / sqrt(psd))`` (bins ``kmin .. N/2 - 1``, PyCBC's ``ifft`` = unnormalised C2R x ``delta_f``), the two
    ends of q tapered with the halves of ``numpy.hanning(max_filter_len)``, the middle zeroed,
    ``psd_out = 1 / |fft(q)|^2`` (``fft`` = ``rfft`` x ``delta_t``).
  * the filter: ``white = (ts.to_frequencyseries() * (1 / psd_out) ** 0.5).to_timeseries()`` -- with the two ``delta``
    factors cancelling this is ``irfft(rfft(x) * |fft(q)|)`` -- and ``max_filter_len // 2`` corrupted samples dropped
    on each side (``remove_corrupted``).

It pins the HIP path (``gw_whisper_amd/whiten.py``) to THIS restatement only and must be re-verified against PyCBC when
one is available; the trained networks depend on the absolute scale of the whitened strain.
"""

from __future__ import annotations

import numpy as np


def median_bias(n: int) -> float:
    """pycbc.psd.estimate.median_bias."""
    if n >= 1000:
        return float(np.log(2))
    ans = 1.0
    for i in range(1, int((n - 1) / 2 + 1)):
        ans += 1.0 / (2 * i + 1) - 1.0 / (2 * i)
    return ans


def welch_segments(n_samples: int, seg_len: int, seg_stride: int):
    """(num_segments, first sample) of pycbc.psd.welch with require_exact_data_fit=False."""
    num_segments = int(n_samples // seg_stride)
    if (num_segments - 1) * seg_stride + seg_len > n_samples:
        num_segments -= 1
    data_len = (num_segments - 1) * seg_stride + seg_len
    start = 0
    if data_len < n_samples:
        diff = n_samples - data_len
        start = diff // 2 + (diff % 2)
    if num_segments < 1 or data_len > n_samples:
        raise ValueError("not enough data for one PSD segment")
    return num_segments, start


def welch_median_psd(x: np.ndarray, delta_t: float, segment_duration: float):
    """(psd [seg_len / 2 + 1], delta_f) as ``TimeSeries(x, delta_t).psd(segment_duration)``."""
    x = np.asarray(x, np.float64)
    seg_len = int(round(segment_duration / delta_t))
    seg_stride = int(seg_len / 2)
    n_seg, start = welch_segments(len(x), seg_len, seg_stride)
    w = np.hanning(seg_len)
    delta_f = 1.0 / delta_t / seg_len
    psds = np.empty((n_seg, seg_len // 2 + 1))
    for i in range(n_seg):
        seg = x[start + i * seg_stride: start + i * seg_stride + seg_len]
        t = np.fft.rfft(seg * w) * delta_t
        p = np.abs(t * t.conj())
        p[0] /= 2
        p[-1] /= 2
        psds[i] = p
    psd = np.median(psds, axis=0) / median_bias(n_seg)
    psd *= 2 * delta_f * seg_len / (w * w).sum()
    return psd, delta_f


def interpolate(psd: np.ndarray, delta_f_old: float, delta_f_new: float) -> np.ndarray:
    new_n = (len(psd) - 1) * delta_f_old / delta_f_new + 1
    samples = np.arange(0, np.rint(new_n)) * delta_f_new
    return np.interp(samples, np.arange(len(psd)) * delta_f_old, psd)


def inverse_spectrum_truncation(psd: np.ndarray, delta_f: float, max_filter_len: int, low_frequency_cutoff=None,
                                trunc_method="hann") -> np.ndarray:
    N = (len(psd) - 1) * 2
    inv_asd = np.zeros(len(psd), np.complex128)
    kmin = int(low_frequency_cutoff / delta_f) if low_frequency_cutoff else 1
    with np.errstate(divide="ignore"):
        inv_asd[kmin:N // 2] = (1.0 / psd[kmin:N // 2]) ** 0.5
    delta_t = 1.0 / (N * delta_f)
    q = np.fft.irfft(inv_asd, N) * N * delta_f                 # PyCBC ifft: unnormalised C2R x delta_f
    trunc_start, trunc_end = max_filter_len // 2, N - max_filter_len // 2
    if trunc_end < trunc_start:
        raise ValueError("Invalid value in inverse_spectrum_truncation")
    if trunc_method == "hann":
        win = np.hanning(max_filter_len)
        q[0:trunc_start] *= win[-trunc_start:]
        q[trunc_end:N] *= win[0:max_filter_len // 2]
    if trunc_start < trunc_end:
        q[trunc_start:trunc_end] = 0
    psd_trunc = np.fft.rfft(q) * delta_t                        # PyCBC fft: rfft x delta_t
    psd_trunc = psd_trunc * psd_trunc.conj()
    with np.errstate(divide="ignore"):
        return 1.0 / np.abs(psd_trunc)


def whiten(strain: np.ndarray, delta_t: float = 1.0 / 2048.0, segment_duration: float = 0.5,
           max_filter_duration: float = 0.25, trunc_method="hann", remove_corrupted: bool = True,
           low_frequency_cutoff=None, return_psd: bool = False):
    """``MLGWSC-1/inference.py:56-137`` for a 1-D or [detectors, samples] array (an even number of samples, as the
    reference's own PSD-length warning assumes)."""
    strain = np.asarray(strain, np.float64)
    if strain.ndim == 2:
        res = [whiten(s, delta_t, segment_duration, max_filter_duration, trunc_method, remove_corrupted,
                      low_frequency_cutoff, return_psd) for s in strain]
        if return_psd:
            return np.stack([r[0] for r in res]), [r[1] for r in res]
        return np.stack(res)
    n = len(strain)
    if n % 2:
        raise ValueError("whiten: an even number of samples is required (the PSD has N / 2 + 1 bins)")
    psd_w, df_w = welch_median_psd(strain, delta_t, segment_duration)
    delta_f = 1.0 / (n * delta_t)
    psd = interpolate(psd_w, df_w, delta_f)
    max_filter_len = int(max_filter_duration * (1.0 / delta_t))
    psd = inverse_spectrum_truncation(psd, delta_f, max_filter_len, low_frequency_cutoff, trunc_method)
    with np.errstate(divide="ignore"):
        inv_psd = 1.0 / psd
    white = np.fft.irfft(np.fft.rfft(strain) * delta_t * inv_psd ** 0.5, n) * n * delta_f
    if remove_corrupted:
        white = white[max_filter_len // 2: n - max_filter_len // 2]
    return (white, psd_w) if return_psd else white
