"""Search-pipeline pieces on the GPU (gw_whisper_amd/inference.py): FFT-resampling front end, MLGWSC classifier
shell, device slicer + on-device thresholding.  Needs an MI355X."""
import numpy as np
import pytest

from gw_whisper_amd import synth
from oracle import encoder as oenc
from oracle import logmel as olm

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import torch
    assert torch.cuda.is_available()
    return torch


def test_resample_matches_scipy(T, gww):
    """Signal_vs_Noise/utils/preprocess.py:44-51: scipy.signal.resample(x, 16000) of 2048-sample segments."""
    from scipy.signal import resample
    from gw_whisper_amd import inference as inf
    x = np.random.default_rng(0).standard_normal((37, 2048)).astype(np.float32)
    got = inf.resample(T.from_numpy(x).cuda()).cpu().numpy()
    ref = resample(x.astype(np.float64), 16000, axis=1)
    assert got.shape == (37, 16000)
    np.testing.assert_allclose(got, ref, atol=2e-5)


def test_resample_mel_adapter_and_classifier_shell(T, gww):
    """GWWhisperClassifier (MLGWSC-1/inference.py:353-392) with the resample + log-mel front end: equals the
    oracle pipeline scipy.resample -> log-mel -> encoder -> last token per detector -> cat -> MLP -> softmax."""
    from scipy.signal import resample
    from gw_whisper_amd import inference as inf
    from gw_whisper_amd.encoder import WhisperConfig, WhisperEncoder
    from oracle import heads as oheads
    rng = np.random.default_rng(1)
    x = rng.standard_normal((3, 2, 2048)).astype(np.float32)
    sd = synth.encoder_state_dict(128, 2, 2, 512, seed=3)
    enc = WhisperEncoder.from_numpy_state_dict(sd, WhisperConfig(128, 2, 2, 512), precision="fp32").cuda()
    model = inf.GWWhisperClassifier(enc, n_detectors=2, num_classes=2).cuda().eval()
    head = {k: v.detach().cpu().numpy() for k, v in model.classifier.state_dict().items()}
    with T.no_grad():
        feats = model.adapter(T.from_numpy(x).cuda())
        probs = model(T.from_numpy(x).cuda()).cpu().numpy()
    assert feats.shape == (3, 2, 80, 3000)
    wave = resample(x.reshape(6, 2048).astype(np.float64), 16000, axis=1).astype(np.float32)
    mel = olm.log_mel(wave)
    # log10 of small mel powers amplifies the fp32 rounding of the resampled wave (1e-5) in a few bins
    np.testing.assert_allclose(feats.reshape(6, 80, 3000).cpu().numpy(), mel, atol=2e-3)
    tok = oenc.encoder_forward(sd, mel, oenc.EncCfg(128, 2, 2, 512), dtype=np.float64)[:, -1, :].reshape(3, 2 * 128)
    logits = oheads.mlp(tok, head)
    e = np.exp(logits - logits.max(1, keepdims=True))
    ref = e / e.sum(1, keepdims=True)
    np.testing.assert_allclose(probs, ref, atol=1e-3)
    assert np.allclose(probs.sum(1), 1.0, atol=1e-6)
    inf.remove_softmax_from_classifier(model)
    with T.no_grad():
        raw = model(T.from_numpy(x).cuda()).cpu().numpy()
    np.testing.assert_allclose(raw, logits, atol=1e-3)


def test_device_slicer_pipeline_equals_per_window_evaluation(T, gww):
    """evaluate_slices over strided device windows == the network applied to each window separately, and the
    triggers are exactly the windows above threshold with the reference's time stamps."""
    from gw_whisper_amd import inference as inf
    from gw_whisper_amd.encoder import WhisperConfig, WhisperEncoder
    rng = np.random.default_rng(2)
    strain = rng.standard_normal((2, 2048 + 204 * 11)).astype(np.float32)
    sd = synth.encoder_state_dict(128, 2, 2, 512, seed=3)
    enc = WhisperEncoder.from_numpy_state_dict(sd, WhisperConfig(128, 2, 2, 512), precision="bf16").cuda()
    model = inf.GWWhisperClassifier(enc).cuda().eval()
    sl = inf.DeviceSegmentSlicer(strain, start_time=100.0)
    assert len(sl) == 12
    trig, vals = inf.evaluate_slices(sl, model, trigger_threshold=0.5, batch_size=5)
    scores = np.concatenate(vals)
    assert len(vals) == 3 and scores.shape == (12,)
    with T.no_grad():
        single = np.array([model(T.from_numpy(strain[None, :, 204 * i:204 * i + 2048]).cuda())[0, 0].item() for i in range(12)])
    np.testing.assert_allclose(scores, single, atol=2e-3)      # batch composition does not matter
    ref = [[100.0 + i * 204 / 2048 + 0.6, scores[i]] for i in range(12) if scores[i] > 0.5]
    assert len(trig) == len(ref)
    if ref:
        np.testing.assert_allclose(np.array(trig), np.array(ref), atol=1e-6)
