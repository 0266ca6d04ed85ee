"""End-to-end parity of the HIP encoder against the oracle and the golden vectors made
from the real HuggingFace encoder + the reference's src/model.py.  Needs an MI355X."""

import numpy as np
import pytest

from gw_whisper_amd import synth
from oracle import encoder as oenc
from oracle import heads as oheads
from oracle import logmel as olm

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import torch
    assert torch.cuda.is_available()
    return torch


def _small(T, precision):
    from gw_whisper_amd.encoder import WhisperConfig, WhisperEncoder
    sd = synth.encoder_state_dict(128, 2, 2, 512, seed=3)
    enc = WhisperEncoder.from_numpy_state_dict(sd, WhisperConfig(128, 2, 2, 512), precision=precision).cuda()
    return enc, sd


def test_small_encoder_fp32_matches_hf_golden(T, gww, golden):
    """fp32 MFMA path vs the HuggingFace fp32 encoder on the reduced config: 1e-3 is the
    north-star tolerance; the fp32 path lands two orders of magnitude inside it."""
    from gw_whisper_amd import ops
    g = golden("encoder_small.npz")
    enc, _ = _small(T, "fp32")
    mel = ops.logmel(T.from_numpy(synth.strain_segments(2, seed=21)).cuda())
    with T.no_grad():
        out = enc(mel).last_hidden_state.cpu().numpy()
        last = enc.last_token(mel).cpu().numpy()
    assert out.shape == (2, 1500, 128)
    np.testing.assert_allclose(out[:, g["rows"]], g["final"], atol=2e-4, rtol=1e-4)
    np.testing.assert_allclose(np.abs(out).mean(axis=(1, 2)), g["final_mean_abs"], rtol=1e-4)
    np.testing.assert_array_equal(last, out[:, -1])


def test_small_encoder_fp32_matches_oracle_everywhere(T, gww):
    enc, sd = _small(T, "fp32")
    mel = olm.log_mel(synth.strain_segments(2, seed=21))
    ref = oenc.encoder_forward(sd, mel, oenc.EncCfg(128, 2, 2, 512), dtype=np.float64)
    with T.no_grad():
        out = enc(T.from_numpy(mel).cuda()).last_hidden_state.cpu().numpy()
    np.testing.assert_allclose(out, ref, atol=2e-4, rtol=1e-4)


def test_small_encoder_bf16_matches_bf16_oracle(T, gww, golden):
    """bf16 MFMA path vs the oracle with the same operand rounding (tight), and vs the
    HF fp32 golden (loose: bf16 operand noise on unit-scale outputs)."""
    g = golden("encoder_small.npz")
    enc, sd = _small(T, "bf16")
    mel = olm.log_mel(synth.strain_segments(2, seed=21))
    ref = oenc.encoder_forward(sd, mel, oenc.EncCfg(128, 2, 2, 512), dtype=np.float32, emulate_bf16=True)
    with T.no_grad():
        out = enc(T.from_numpy(mel).cuda()).last_hidden_state.cpu().numpy()
    # same roundings, different accumulation order / bf16 storage of q,k,v,ctx,fc1
    assert np.abs(out - ref).max() < 3e-2
    assert np.sqrt(((out - ref) ** 2).mean()) < 3e-3
    assert np.abs(out[:, g["rows"]] - g["final"]).max() < 6e-2


def test_batch_independence_and_determinism(T, gww):
    """Size-independent properties: a segment's output does not depend on its batch
    neighbours, and two runs are bit-identical."""
    enc, _ = _small(T, "bf16")
    mel = T.from_numpy(olm.log_mel(synth.strain_segments(5, seed=4))).cuda()
    with T.no_grad():
        a = enc(mel).last_hidden_state
        b = enc(mel).last_hidden_state
        c = enc(mel[3:4]).last_hidden_state
    assert T.equal(a, b)
    assert T.equal(a[3:4], c)


def test_wrong_length_raises_like_hf(T, gww):
    enc, _ = _small(T, "bf16")
    with pytest.raises(ValueError, match="3000"):
        enc(T.zeros(1, 80, 2998).cuda())


@pytest.mark.parametrize("precision,tol_last,tol_logit", [("fp32", 5e-4, 1e-3), ("bf16", 8e-2, 1e-3)])
def test_config1_tiny_against_reference_golden(T, gww, golden, precision, tol_last, tol_logit):
    """BASELINE config 1: 64 two-detector segments, whisper-tiny, reference
    two_channel / one_channel classifiers.  Pooled token vs HF, logits within 1e-3,
    sigmoid().round() labels bit-exact."""
    from gw_whisper_amd import ops
    from gw_whisper_amd.encoder import WhisperConfig, WhisperEncoder
    g = golden("config1.npz")
    sd = synth.named_encoder_state_dict("tiny", seed=0)
    enc = WhisperEncoder.from_numpy_state_dict(sd, WhisperConfig.named("tiny"), precision=precision).cuda()
    n = 64
    h1 = synth.strain_segments(n, seed=0)
    l1 = synth.strain_segments(n, seed=1)
    t = np.arange(16000, dtype=np.float32) / 16000.0
    for i in range(0, n, 2):
        s = (3.0 * np.sin(2 * np.pi * (40.0 + 200.0 * t * (1 + 0.05 * i)) * t) * np.exp(-((t - 0.6) / 0.15) ** 2))
        h1[i] += s.astype(np.float32)
        l1[i] += s.astype(np.float32)
    with T.no_grad():
        la = enc.last_token(ops.logmel(T.from_numpy(h1).cuda())).cpu().numpy()
        lb = enc.last_token(ops.logmel(T.from_numpy(l1).cuda())).cpu().numpy()
    err = max(np.abs(la - g["last_token"][:, 0]).max(), np.abs(lb - g["last_token"][:, 1]).max())
    print(f"[{precision}] max |last_token - HF| = {err:.3e}")
    assert err < tol_last
    head2 = synth.head_state_dict([768, 1024, 512, 256, 1], seed=0)
    head2["6.bias"] = head2["6.bias"] + g["two_channel_bias_shift"]
    head1 = synth.head_state_dict([384, 512, 256, 128, 64, 1], seed=1)
    head1["8.bias"] = head1["8.bias"] + g["one_channel_bias_shift"]
    lg2 = oheads.two_channel_logits(la, lb, head2)
    lg1 = oheads.one_channel_logits(lb, head1)
    e2 = np.abs(lg2 - g["two_channel_logits"]).max()
    e1 = np.abs(lg1 - g["one_channel_logits"]).max()
    print(f"[{precision}] max |logit - reference|: two-channel {e2:.3e}, one-channel {e1:.3e}")
    assert e2 < tol_logit and e1 < tol_logit
    np.testing.assert_array_equal(oheads.binary_labels(lg2), g["two_channel_labels"])
    np.testing.assert_array_equal(oheads.binary_labels(lg1), g["one_channel_labels"])
