"""Oracle: Whisper log-mel front end (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Restates, in numpy, what the reference obtains from
``WhisperFeatureExtractor(audio, sampling_rate=16000, return_tensors="pt")``
(call sites: reference ``Signal_vs_Noise/src/dataset.py:20-21,40``,
``Glitch_classification/src/dataset.py:46``).  The arithmetic lives in the
third-party ``transformers`` package (pinned 4.37.2 in the reference's
``requirements.txt:296``; 5.15.0 is what is installed in the build container):

  * ``HF:models/whisper/feature_extraction_whisper.py:95-103``  filterbank
    construction (slaney scale + slaney norm, 0..8000 Hz, 201 x 80)
  * ``HF:audio_utils.py:448-520``  hertz_to_mel / mel_to_hertz (slaney)
  * ``HF:audio_utils.py:638-731``  mel_filter_bank (triangles + slaney norm)
  * ``HF:models/whisper/feature_extraction_whisper.py:135-168``
    ``_torch_extract_fbank_features``: zero-pad to 480000, ``torch.stft``
    (n_fft 400, hop 160, periodic Hann, center=True / reflect), drop the last
    frame, ``|X|^2``, ``mel.T @ P``, ``clamp(1e-10).log10()``, per-sample max,
    ``max(x, max-8)``, ``(x+4)/4``.

Pinned by ``tests/golden/logmel.npz`` (made from the real HF extractor).
"""

from __future__ import annotations

import numpy as np

SAMPLING_RATE = 16000
N_FFT = 400
HOP = 160
N_MELS = 80
N_FREQ = N_FFT // 2 + 1          # 201
CHUNK_SAMPLES = 30 * SAMPLING_RATE  # 480000
N_FRAMES = CHUNK_SAMPLES // HOP     # 3000


def hertz_to_mel_slaney(freq):
    """HF:audio_utils.py:470-481."""
    freq = np.asarray(freq, dtype=np.float64)
    mels = 3.0 * freq / 200.0
    logstep = 27.0 / np.log(6.4)
    log_region = freq >= 1000.0
    with np.errstate(divide="ignore", invalid="ignore"):
        mels = np.where(log_region, 15.0 + np.log(np.maximum(freq, 1e-300) / 1000.0) * logstep, mels)
    return mels


def mel_to_hertz_slaney(mels):
    """HF:audio_utils.py:484-520 (slaney branch)."""
    mels = np.asarray(mels, dtype=np.float64)
    freq = 200.0 * mels / 3.0
    logstep = np.log(6.4) / 27.0
    log_region = mels >= 15.0
    freq = np.where(log_region, 1000.0 * np.exp(logstep * (mels - 15.0)), freq)
    return freq


def mel_filter_bank(n_freq: int = N_FREQ, n_mels: int = N_MELS, fmin: float = 0.0,
                    fmax: float = 8000.0, sr: int = SAMPLING_RATE) -> np.ndarray:
    """[n_freq, n_mels] float64 slaney/slaney filterbank (HF:audio_utils.py:638-731)."""
    mel_min = hertz_to_mel_slaney(fmin)
    mel_max = hertz_to_mel_slaney(fmax)
    mel_freqs = np.linspace(mel_min, mel_max, n_mels + 2)
    filter_freqs = mel_to_hertz_slaney(mel_freqs)
    fft_freqs = np.linspace(0, sr // 2, n_freq)
    filter_diff = np.diff(filter_freqs)
    slopes = np.expand_dims(filter_freqs, 0) - np.expand_dims(fft_freqs, 1)
    down = -slopes[:, :-2] / filter_diff[:-1]
    up = slopes[:, 2:] / filter_diff[1:]
    fb = np.maximum(0.0, np.minimum(down, up))
    enorm = 2.0 / (filter_freqs[2:n_mels + 2] - filter_freqs[:n_mels])
    fb = fb * np.expand_dims(enorm, 0)
    return fb


def hann_periodic(n: int = N_FFT) -> np.ndarray:
    """``torch.hann_window(n)`` (periodic=True): 0.5 - 0.5 cos(2 pi k / n)."""
    k = np.arange(n, dtype=np.float64)
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * k / n)


def pad_or_trim(wave: np.ndarray) -> np.ndarray:
    """Zero-pad / truncate to 480000 samples (HF ``self.pad(..., max_length=n_samples, truncation=True)``)."""
    wave = np.asarray(wave)
    if wave.ndim == 1:
        wave = wave[None]
    out = np.zeros((wave.shape[0], CHUNK_SAMPLES), dtype=wave.dtype)
    n = min(wave.shape[1], CHUNK_SAMPLES)
    out[:, :n] = wave[:, :n]
    return out


def stft_power(padded: np.ndarray, dtype=np.float64) -> np.ndarray:
    """|STFT|^2 of the 480000-sample buffers, frames 0..2999 -> [N, 201, 3000].

    ``torch.stft(center=True, pad_mode="reflect")`` then ``[..., :-1]``.
    """
    x = np.asarray(padded, dtype=dtype)
    n = x.shape[0]
    xp = np.pad(x, ((0, 0), (N_FFT // 2, N_FFT // 2)), mode="reflect")
    win = hann_periodic().astype(dtype)
    # frame t covers xp[160 t : 160 t + 400]
    idx = (np.arange(N_FRAMES)[:, None] * HOP + np.arange(N_FFT)[None, :])
    out = np.empty((n, N_FREQ, N_FRAMES), dtype=dtype)
    for i in range(n):
        frames = xp[i][idx] * win[None, :]
        spec = np.fft.rfft(frames.astype(np.float64), axis=1)
        out[i] = (spec.real ** 2 + spec.imag ** 2).T.astype(dtype)
    return out


def log_mel(wave: np.ndarray, dtype=np.float32) -> np.ndarray:
    """[N, L] (or [L]) waveform -> [N, 80, 3000] ``input_features`` in ``dtype``.

    ``dtype=np.float32`` follows HF's fp32 torch path (filterbank cast to f32,
    f32 matmul / log10); ``np.float64`` is the high-precision variant used to
    bound the f32 rounding noise in the tests.
    """
    padded = pad_or_trim(np.asarray(wave, dtype=np.float32))
    power = stft_power(padded, dtype=dtype)
    fb = mel_filter_bank().astype(dtype)                      # [201, 80]
    mel = np.einsum("fm,nft->nmt", fb, power).astype(dtype)   # mel.T @ P
    # log10 evaluated in f64 and rounded once: numpy's f32 log10 is 1 ulp off at
    # 1e-10 (-10.000001) where torch gives the correctly rounded -10.0
    log_spec = np.log10(np.maximum(mel, dtype(1e-10)).astype(np.float64)).astype(dtype)
    mx = log_spec.reshape(log_spec.shape[0], -1).max(axis=1)[:, None, None]
    log_spec = np.maximum(log_spec, mx - dtype(8.0))
    return ((log_spec + dtype(4.0)) / dtype(4.0)).astype(dtype)


def live_frames(n_samples: int) -> int:
    """Number of leading frames that can see a non-zero sample.

    Frame t covers unpadded samples [160 t - 200, 160 t + 200); it is all-zero
    iff 160 t - 200 >= n_samples (and no reflect tail reaches it).  For
    n_samples = 16000 this is 102.
    """
    if n_samples >= CHUNK_SAMPLES - N_FFT:
        return N_FRAMES
    return min(N_FRAMES, -(-(n_samples + N_FFT // 2) // HOP))
