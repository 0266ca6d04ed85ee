"""Q-transform front end #2 -- CPU restatement (test infrastructure only).  **PARITY UNPINNED.**

The reference gets its Q-scan from ``ml4gw.transforms.QScan`` (``MLGWSC-1/train.py:45,117-122``,
``inference.py:28``): a third-party dependency that is neither vendored nor pinned (absent from
``requirements.txt``) and not installed in any environment this build can run in, and the reference holds no test
or golden vector for it.  What follows restates the published constant-Q tiling (Chatterji et al. 2004, as
implemented by GWpy's ``qtransform`` and ported to torch by ml4gw ``transforms/qtransform.py``): ``QTile`` ->
``SingleQTransform`` -> ``QScan``, with ml4gw's conventions as known at the time of writing (forward-normalised
rFFT with the positive frequencies doubled, bisquare window, median normalisation, plane with the largest tile
energy over the whole batch, bicubic interpolation with PyTorch's ``align_corners=False`` / A = -0.75 rules).
It pins the HIP kernels to THIS restatement only; it must be re-verified against an ml4gw source when one is
available.
"""

from __future__ import annotations

import math

import numpy as np


class QTile:
    def __init__(self, q: float, frequency: float, duration: float, sample_rate: float, mismatch: float):
        self.q, self.frequency, self.duration, self.sample_rate, self.mismatch = q, frequency, duration, sample_rate, mismatch
        self.deltam = 2.0 * (mismatch / 3.0) ** 0.5
        self.qprime = q / 11 ** 0.5
        self.windowsize = 2 * int(frequency / self.qprime * duration) + 1
        tcum_mismatch = duration * 2 * math.pi * frequency / q
        self.ntiles = int(2 ** math.ceil(math.log2(tcum_mismatch / self.deltam)))
        half = int((self.windowsize - 1) / 2.0)
        k = np.arange(-half, half + 1)
        xfrequencies = (k / duration) * self.qprime / frequency
        norm = self.ntiles / (duration * sample_rate) * (315 * self.qprime / (128 * frequency)) ** 0.5
        self.window = (1 - xfrequencies ** 2) ** 2 * norm
        self.indices = np.round(k + 1 + frequency * duration).astype(np.int64)
        pad = self.ntiles - self.windowsize
        self.padding = (int((pad - 1) / 2.0), int((pad + 1) / 2.0))

    def energy(self, fseries: np.ndarray, norm: bool = True) -> np.ndarray:
        """fseries [B, n_freq] complex -> tile energies [B, ntiles] (median-normalised)."""
        windowed = fseries[..., self.indices] * self.window
        padded = np.pad(windowed, ((0, 0), self.padding))
        tdenergy = np.fft.ifft(np.fft.ifftshift(padded, axes=-1), axis=-1)
        energy = tdenergy.real ** 2 + tdenergy.imag ** 2
        if norm:
            energy = energy / np.quantile(energy, 0.5, axis=-1, keepdims=True)
        return energy


def plane_frequencies(q: float, duration: float, sample_rate: float, mismatch: float, frange=(0.0, math.inf)):
    qprime = q / 11 ** 0.5
    minf = max(frange[0], 50 * q / (2 * math.pi * duration))
    maxf = min(frange[1], sample_rate / 2 / (1 + 1 / qprime))
    fcum_mismatch = math.log(maxf / minf) * (2 + q ** 2) ** 0.5 / 2.0
    deltam = 2 * (mismatch / 3.0) ** 0.5
    nfreq = int(max(1, math.ceil(fcum_mismatch / deltam)))
    fstep = fcum_mismatch / nfreq
    fstepmin = 1 / duration
    freq_base = np.exp(2 / ((2 + q ** 2) ** 0.5) * (np.arange(0, nfreq) + 0.5) * fstep)
    freqs = (minf * freq_base // fstepmin) * fstepmin
    return np.unique(freqs)


def plane_qs(qrange, mismatch: float):
    deltam = 2 * (mismatch / 3.0) ** 0.5
    cumum = math.log(qrange[1] / qrange[0]) / 2 ** 0.5
    nplanes = int(max(math.ceil(cumum / deltam), 1))
    dq = cumum / nplanes
    return [qrange[0] * math.exp(2 ** 0.5 * dq * (i + 0.5)) for i in range(nplanes)]


def tiling(duration=1.0, sample_rate=2048, qrange=(4, 128), mismatch=0.2, frange=(0.0, math.inf)):
    """[[QTile, ...] per Q plane] -- the static geometry the HIP kernels are built from."""
    return [[QTile(q, f, duration, sample_rate, mismatch) for f in plane_frequencies(q, duration, sample_rate, mismatch, frange)]
            for q in plane_qs(qrange, mismatch)]


def _cubic_weights(t: np.ndarray, A: float = -0.75):
    """PyTorch upsample_bicubic2d coefficients for the 4 taps at offsets -1, 0, 1, 2."""
    def c1(x):
        return ((A + 2) * x - (A + 3)) * x * x + 1
    def c2(x):
        return ((A * x - 5 * A) * x + 8 * A) * x - 4 * A
    return np.stack([c2(t + 1.0), c1(t), c1(1.0 - t), c2(2.0 - t)], axis=-1)


def cubic_resize_last(x: np.ndarray, n_out: int) -> np.ndarray:
    """1-D cubic resampling of the last axis with ``F.interpolate(mode='bicubic', align_corners=False)`` rules."""
    n_in = x.shape[-1]
    if n_in == n_out:
        return x.copy()
    scale = n_in / n_out
    src = (np.arange(n_out) + 0.5) * scale - 0.5
    i0 = np.floor(src).astype(np.int64)
    w = _cubic_weights(src - i0)
    out = np.zeros(x.shape[:-1] + (n_out,), dtype=x.dtype)
    for tap in range(4):
        idx = np.clip(i0 - 1 + tap, 0, n_in - 1)
        out += x[..., idx] * w[:, tap]
    return out


def qscan(x: np.ndarray, duration=1.0, sample_rate=2048, spectrogram_shape=(128, 128), qrange=(4, 128),
          mismatch=0.2, return_plane: bool = False):
    """x [B, duration * sample_rate] -> [B, F, T] (``QScan(...)(x)``)."""
    x = np.asarray(x, np.float64)
    X = np.fft.rfft(x, axis=-1) / x.shape[-1]          # norm="forward"
    X[..., 1:] *= 2
    planes = tiling(duration, sample_rate, qrange, mismatch)
    energies = [[t.energy(X) for t in plane] for plane in planes]
    maxima = [max(float(e.max()) for e in plane) for plane in energies]       # over tiles AND the whole batch
    best = int(np.argmax(maxima))
    num_f, num_t = spectrogram_shape
    rows = np.stack([cubic_resize_last(e, num_t) for e in energies[best]], axis=-2)      # [B, nfreq, T]
    out = np.swapaxes(cubic_resize_last(np.swapaxes(rows, -1, -2), num_f), -1, -2)       # [B, F, T]
    return (out, best) if return_plane else out
