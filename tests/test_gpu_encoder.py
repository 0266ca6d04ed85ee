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


def test_dora_adapted_encoder_matches_oracle(T, gww):
    """get_peft_model(use_dora=True) on k_proj/v_proj with non-trivial A, B, m: the HIP merge
    kernel + packed forward equals the oracle run on oracle-merged weights
    (peft 0.12.0 dora.py formula), and an untouched adapter is the identity."""
    from gw_whisper_amd.encoder import WhisperConfig, WhisperEncoder
    from gw_whisper_amd.peft import LoraConfig, get_peft_model
    from oracle import dora as odora
    cfg = WhisperConfig(128, 2, 2, 512)
    sd = synth.encoder_state_dict(128, 2, 2, 512, seed=3)
    mel = olm.log_mel(synth.strain_segments(2, seed=21))
    base = oenc.encoder_forward(sd, mel, oenc.EncCfg(128, 2, 2, 512), dtype=np.float64)
    enc = WhisperEncoder.from_numpy_state_dict(sd, cfg, precision="fp32")
    targets = [f"layers.{i}.self_attn.{p}" for i in range(2) for p in ("k_proj", "v_proj")]
    peft = get_peft_model(enc, LoraConfig(use_dora=True, r=8, lora_alpha=32, target_modules=targets)).cuda()
    with T.no_grad():
        out0 = peft(T.from_numpy(mel).cuda()).last_hidden_state.cpu().numpy()
    np.testing.assert_allclose(out0, base, atol=2e-4, rtol=1e-4)          # identity at init
    sd2 = dict(sd)
    with T.no_grad():
        for j, name in enumerate(targets):
            lin = peft.base_model.model.get_submodule(name)
            W0 = sd[name + ".weight"]
            A, B, m = synth.dora_adapter(128, 128, 8, W0, seed=50 + j)
            lin.lora_A["default"].weight.copy_(T.from_numpy(A))
            lin.lora_B["default"].weight.copy_(T.from_numpy(B))
            lin.lora_magnitude_vector["default"].weight.copy_(T.from_numpy(m))
            sd2[name + ".weight"] = odora.dora_merge(W0.astype(np.float64), A.astype(np.float64),
                                                     B.astype(np.float64), m.astype(np.float64), 4.0)
        out1 = peft(T.from_numpy(mel).cuda()).last_hidden_state.cpu().numpy()
    ref1 = oenc.encoder_forward(sd2, mel, oenc.EncCfg(128, 2, 2, 512), dtype=np.float64)
    assert np.abs(ref1 - base).max() > 1e-2, "the adapter must actually change the output"
    np.testing.assert_allclose(out1, ref1, atol=2e-4, rtol=1e-4)


def test_two_channel_model_on_gpu_matches_reference_logits(T, gww, golden):
    """The counterpart of Signal_vs_Noise/src/model.py run end to end on the GPU (HIP front end,
    HIP encoder, torch MLP head): logits within 1e-3 of the reference's CPU logits, labels exact."""
    from gw_whisper_amd import ops
    from gw_whisper_amd.encoder import WhisperConfig, WhisperEncoder
    from gw_whisper_amd.models import one_channel_ligo_binary_classifier, two_channel_ligo_binary_classifier
    g = golden("config1.npz")
    sd = synth.named_encoder_state_dict("tiny", seed=0)
    enc = WhisperEncoder.from_numpy_state_dict(sd, WhisperConfig.named("tiny"), precision="bf16")
    m2 = two_channel_ligo_binary_classifier(enc)
    m1 = one_channel_ligo_binary_classifier(enc)
    head2 = synth.head_state_dict([768, 1024, 512, 256, 1], seed=0)
    head2["6.bias"] = head2["6.bias"] + g["two_channel_bias_shift"]
    head1 = synth.head_state_dict([384, 512, 256, 128, 64, 1], seed=1)
    head1["8.bias"] = head1["8.bias"] + g["one_channel_bias_shift"]
    m2.classifier.load_state_dict({k: T.from_numpy(v) for k, v in head2.items()})
    m1.classifier.load_state_dict({k: T.from_numpy(v) for k, v in head1.items()})
    m2.cuda().eval(); m1.cuda().eval()
    n = 64
    h1 = synth.strain_segments(n, seed=0)
    l1 = synth.strain_segments(n, seed=1)
    t = np.arange(16000, dtype=np.float32) / 16000.0
    for i in range(0, n, 2):
        s = (3.0 * np.sin(2 * np.pi * (40.0 + 200.0 * t * (1 + 0.05 * i)) * t) * np.exp(-((t - 0.6) / 0.15) ** 2))
        h1[i] += s.astype(np.float32)
        l1[i] += s.astype(np.float32)
    with T.no_grad():
        a = ops.logmel(T.from_numpy(h1).cuda())
        b = ops.logmel(T.from_numpy(l1).cuda())
        lg2 = m2(a, b).cpu().numpy()
        lg1 = m1(b).cpu().numpy()
    assert np.abs(lg2 - g["two_channel_logits"]).max() < 1e-3
    assert np.abs(lg1 - g["one_channel_logits"]).max() < 1e-3
    np.testing.assert_array_equal(oheads.binary_labels(lg2), g["two_channel_labels"])
    np.testing.assert_array_equal(oheads.binary_labels(lg1), g["one_channel_labels"])


def test_dual_stream_split_is_bit_identical(T, gww):
    """gww_encoder_set_split: two half batches on two streams give exactly the single-stream result."""
    from gw_whisper_amd.encoder import WhisperConfig, WhisperEncoder
    sd = synth.encoder_state_dict(128, 2, 2, 512, seed=3)
    enc = WhisperEncoder.from_numpy_state_dict(sd, WhisperConfig(128, 2, 2, 512), precision="bf16").cuda()
    g = T.Generator(device="cpu").manual_seed(1)
    mel = (T.randn(70, 80, 3000, generator=g) * 0.5).cuda()
    with T.no_grad():
        a_h, a_l = enc.forward_raw(mel, want_hidden=True, want_last=True)
        enc.set_split(True)
        b_h, b_l = enc.forward_raw(mel, want_hidden=True, want_last=True)
        T.cuda.synchronize()
    assert T.equal(a_h, b_h) and T.equal(a_l, b_l)


@pytest.mark.parametrize("name,B", [("tiny", 5), ("base", 2), ("small", 2)])
def test_pooled_forward_equals_last_row_of_full_forward(T, gww, name, B):
    """``encoder.last_token`` (bf16): the last layer runs on the B last-token rows only (attention for the one
    query tile holding token 1499, row-wise ops on B rows) -- the same function as ``last_hidden_state[:, -1]``
    (reference ``src/model.py:25-26``), up to bf16 rounding of different GEMM tilings; also through the two-stream
    split."""
    from gw_whisper_amd.encoder import WhisperConfig, WhisperEncoder
    sd = synth.named_encoder_state_dict(name, seed=2)
    enc = WhisperEncoder.from_numpy_state_dict(sd, WhisperConfig.named(name), precision="bf16").cuda()
    mel = T.from_numpy(olm.log_mel(synth.strain_segments(B, seed=14))).cuda()
    with T.no_grad():
        hidden, last_full = enc.forward_raw(mel, want_hidden=True, want_last=True)
        pooled = enc.last_token(mel)
        enc.set_split(True)
        pooled_split = enc.last_token(mel)
        T.cuda.synchronize()
    assert T.equal(last_full, hidden[:, -1])
    err = (pooled - last_full).abs().max().item()
    print(f"[{name}] max |pooled - full[:, -1]| = {err:.3e} (|x| max {last_full.abs().max().item():.2f})")
    assert err < 0.06
    assert (pooled_split - last_full).abs().max().item() < 0.06


@pytest.mark.parametrize("name", ["base", "small"])
def test_named_sizes_match_oracle(T, gww, name):
    """whisper-base (BASELINE config 4) and whisper-small (configs 3 and 5) geometry, one segment:
    fp32 path vs the fp32 oracle, bf16 path within bf16 tolerance of it, last_token == hidden[:, -1]."""
    from gw_whisper_amd.encoder import WhisperConfig, WhisperEncoder
    d, L, H, F = synth.ENCODER_SIZES[name]
    sd = synth.encoder_state_dict(d, L, H, F, seed=5)
    mel = olm.log_mel(synth.strain_segments(1, seed=9))
    ref = oenc.encoder_forward(sd, mel, oenc.EncCfg(d, L, H, F), dtype=np.float32)
    x = T.from_numpy(mel).cuda()
    enc = WhisperEncoder.from_numpy_state_dict(sd, WhisperConfig.named(name), precision="fp32").cuda()
    with T.no_grad():
        out32 = enc(x).last_hidden_state.cpu().numpy()
        last32 = enc.last_token(x).cpu().numpy()
    e32 = np.abs(out32 - ref).max()
    print(f"[{name}] fp32 max |hidden - oracle| = {e32:.3e}")
    assert e32 < 1e-3
    np.testing.assert_allclose(last32, out32[:, -1], atol=1e-5)
    enc.precision = "bf16"
    with T.no_grad():
        out16 = enc(x).last_hidden_state.cpu().numpy()
    e16 = np.abs(out16 - ref).max()
    print(f"[{name}] bf16 max |hidden - oracle| = {e16:.3e} (|hidden| max {np.abs(ref).max():.2f})")
    assert e16 < 0.15 and np.abs(out16 - ref).mean() < 6e-3


def _config4_segments():
    n = 12
    seg = synth.strain_segments(n, seed=44)
    t = np.arange(16000, dtype=np.float32) / 16000.0
    for i in range(n):      # same construction as tools/make_golden.py::make_config4
        amp = 0.5 * (400.0 ** (i / (n - 1)))
        seg[i] += (amp * np.sin(2 * np.pi * (30.0 + 35.0 * i) * t) * np.exp(-((t - 0.1 - 0.07 * i) / 0.03) ** 2)).astype(np.float32)
    return seg


@pytest.mark.parametrize("precision,tol_last", [("fp32", 5e-4), ("bf16", 8e-2)])
def test_config4_glitch_classifier_against_reference_golden(T, gww, golden, precision, tol_last):
    """BASELINE config 4 through the log-mel front end the reference's Glitch code uses
    (Glitch_classification/src/dataset.py:46): whisper-base, the 22-class head of
    Glitch_classification/src/model.py:10-38 (golden: that very file on an HF whisper-base encoder,
    tools/make_golden.py::make_config4).  ``models.glitch_classifier`` on the GPU: logits within 1e-3, argmax labels
    exact."""
    from gw_whisper_amd import ops
    from gw_whisper_amd.encoder import WhisperConfig, WhisperEncoder
    from gw_whisper_amd.models import glitch_classifier
    g = golden("config4.npz")
    d, L, H, F = synth.ENCODER_SIZES["base"]
    sd = synth.encoder_state_dict(d, L, H, F, seed=4)
    enc = WhisperEncoder.from_numpy_state_dict(sd, WhisperConfig.named("base"), precision=precision)
    model = glitch_classifier(enc, num_classes=22)
    head = synth.head_state_dict([d, 512, 256, 128, 22], seed=769, sequential_stride=3)
    head["9.bias"] = head["9.bias"] + g["class_bias_shift"]
    model.classifier.load_state_dict({k: T.from_numpy(v) for k, v in head.items()})   # keys 0 / 3 / 6 / 9: Dropout slots
    model = model.cuda().eval()
    mel = ops.logmel(T.from_numpy(_config4_segments()).cuda())
    with T.no_grad():
        last = enc.last_token(mel).cpu().numpy()
        logits = model(mel).cpu().numpy()
        hidden_last = enc(mel).last_hidden_state[:, -1, :].cpu().numpy()
    err_last = np.abs(last - g["last_token"]).max()
    err = np.abs(logits - g["logits"]).max()
    print(f"[config 4, {precision}] max |last_token - HF| = {err_last:.3e}, max |logit - reference| = {err:.3e}, "
          f"smallest top-2 margin of the golden = {g['top2_margin'].min():.3e}")
    assert logits.shape == (12, 22)
    assert err_last < tol_last
    np.testing.assert_allclose(hidden_last, last, atol=1e-6 if precision == "fp32" else 2e-2)
    assert err < 1e-3
    np.testing.assert_array_equal(logits.argmax(1), g["labels"])
    assert len(set(g["labels"].tolist())) >= 4


def test_tiny_batch_256_rows_equal_the_batch_64_rows(T, gww, golden):
    """BASELINE config 2 shape (B = 256): every segment's output is independent of the batch it rides in -- rows of
    a B = 256 whisper-tiny bf16 forward are bit-equal to the same rows of four B = 64 calls -- and the first 64
    (config 1's H1 segments) still hit the reference golden."""
    from gw_whisper_amd import ops
    from gw_whisper_amd.encoder import WhisperConfig, WhisperEncoder
    g = golden("config1.npz")
    sd = synth.named_encoder_state_dict("tiny", seed=0)
    enc = WhisperEncoder.from_numpy_state_dict(sd, WhisperConfig.named("tiny"), precision="bf16").cuda()
    h1 = synth.strain_segments(64, seed=0)
    t = np.arange(16000, dtype=np.float32) / 16000.0
    for i in range(0, 64, 2):
        h1[i] += (3.0 * np.sin(2 * np.pi * (40.0 + 200.0 * t * (1 + 0.05 * i)) * t) * np.exp(-((t - 0.6) / 0.15) ** 2)).astype(np.float32)
    wave = np.concatenate([h1, synth.strain_segments(192, seed=5)])
    mel = ops.logmel(T.from_numpy(wave).cuda())
    with T.no_grad():
        big_h, big_l = enc.forward_raw(mel, want_hidden=True, want_last=True)
        pooled = enc.last_token(mel)
        for q in range(4):
            h, l = enc.forward_raw(mel[64 * q:64 * (q + 1)], want_hidden=True, want_last=True)
            assert T.equal(big_h[64 * q:64 * (q + 1)], h), f"hidden rows of quarter {q} depend on the batch"
            assert T.equal(big_l[64 * q:64 * (q + 1)], l)
            assert T.equal(pooled[64 * q:64 * (q + 1)], enc.last_token(mel[64 * q:64 * (q + 1)]))
    assert T.equal(big_h[:, -1, :], big_l)
    err = np.abs(big_l[:64].cpu().numpy() - g["last_token"][:, 0]).max()
    print(f"B = 256: max |last_token[:64] - HF golden| = {err:.3e}")
    assert err < 8e-2
    assert (pooled - big_l).abs().max().item() < 2e-2
