"""Oracle (numpy) log-mel front end vs golden vectors made from the real
HuggingFace WhisperFeatureExtractor (tools/make_golden.py).  CPU only."""

import numpy as np
import pytest

from gw_whisper_amd import synth
from oracle import logmel

# HF's own docstring promises 1e-5 between its numpy and torch paths; our f32
# restatement lands inside that.
TOL = 2e-5


def test_filterbank_shape_and_norm():
    fb = logmel.mel_filter_bank()
    assert fb.shape == (201, 80)
    assert fb.min() >= 0.0
    # slaney-normalised triangles: every filter non-empty, DC bin weightless
    assert (fb.sum(axis=0) > 0).all()
    assert fb[0].sum() == 0.0


def test_live_frames():
    assert logmel.live_frames(16000) == 102
    assert logmel.live_frames(1) == 2
    assert logmel.live_frames(480000) == 3000
    assert logmel.live_frames(0) <= 2


def test_seg16000_matches_hf(golden):
    g = golden("logmel.npz")
    seg = synth.strain_segments(4, seed=11)
    out = logmel.log_mel(seg, dtype=np.float32)
    assert out.shape == (4, 80, 3000) and out.dtype == np.float32
    np.testing.assert_allclose(out[:, :, :112], g["seg16000_frames0_112"], atol=TOL, rtol=0)
    # the padded region is one constant per sample
    for i in range(4):
        assert np.all(out[i, :, 103:] == out[i, 0, 2999])
        assert abs(out[i, 0, 2999] - g["seg16000_pad_value"][i]) < TOL


@pytest.mark.parametrize("n", [1, 159, 12345, 40000])
def test_ragged_lengths(golden, n):
    g = golden("logmel.npz")
    w = synth.strain_segments(1, seed=100 + n, n_samples=n)[0]
    out = logmel.log_mel(w, dtype=np.float32)[0]
    ref = g[f"len{n}_frames"]
    np.testing.assert_allclose(out[:, :ref.shape[1]], ref, atol=TOL, rtol=0)
    assert abs(out[0, 2999] - g[f"len{n}_pad_value"]) < TOL
    live = logmel.live_frames(n)
    assert np.all(out[:, live:] == out[0, 2999])


@pytest.mark.parametrize("n", [480000, 480321])
def test_full_and_truncated(golden, n):
    g = golden("logmel.npz")
    w = synth.strain_segments(1, seed=200 + n, n_samples=n)[0]
    out = logmel.log_mel(w, dtype=np.float32)[0]
    np.testing.assert_allclose(out[:, g[f"len{n}_cols"]], g[f"len{n}_frames"], atol=TOL, rtol=0)


def test_constant_collapse(golden):
    g = golden("logmel.npz")
    z = logmel.log_mel(np.zeros(16000, np.float32))[0]
    assert z.min() == g["zeros_value"][0] and z.max() == g["zeros_value"][1] == -1.5
    r = logmel.log_mel((synth.strain_segments(1, seed=5)[0] * 1e-21).astype(np.float32))[0]
    assert r.min() == g["raw1e21_value"][0] and r.max() == g["raw1e21_value"][1]


def test_f64_bounds_f32_noise():
    seg = synth.strain_segments(2, seed=3)
    a = logmel.log_mel(seg, dtype=np.float32)
    b = logmel.log_mel(seg, dtype=np.float64)
    assert np.abs(a - b).max() < 1e-5
