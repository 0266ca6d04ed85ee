"""Oracle (numpy) encoder / heads / DoRA vs golden vectors made from the real
HuggingFace WhisperEncoder and the reference's Signal_vs_Noise/src/model.py
(tools/make_golden.py).  CPU only."""

import json
import os

import numpy as np
import pytest

from gw_whisper_amd import synth
from oracle import dora, encoder, heads, logmel

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_sinusoids_match_synth():
    np.testing.assert_array_equal(encoder.sinusoids(1500, 384), synth.sinusoid_table(1500, 384))


def test_bf16_round():
    x = np.array([1.0, 1.00390625, 1.001953125, 1.005859375, -3.14159, 0.0, 1e-30], np.float32)
    r = encoder.bf16_round(x)
    assert r[0] == 1.0
    assert r[1] == 1.0            # tie -> even (1.0 has even mantissa)
    assert r[2] == 1.0
    assert r[3] == 1.0078125      # tie -> even upwards
    assert abs(r[4] - x[4]) <= abs(x[4]) * 2 ** -8
    assert (r.view(np.uint32) & 0xFFFF == 0).all()


def test_small_encoder_stages_match_hf(golden):
    g = golden("encoder_small.npz")
    cfg = encoder.EncCfg(128, 2, 2, 512)
    sd = synth.encoder_state_dict(128, 2, 2, 512, seed=3)
    mel = logmel.log_mel(synth.strain_segments(2, seed=21))
    out, st = encoder.encoder_forward(sd, mel, cfg, dtype=np.float32, return_stages=True)
    rows = g["rows"]
    tol = dict(atol=5e-5, rtol=1e-4)
    np.testing.assert_allclose(st["embed"][:, rows], g["embed"], **tol)
    scale = np.float32(64 ** -0.5)
    np.testing.assert_allclose(st["l0.q"][:, rows], g["l0.q_proj"] * scale, **tol)
    np.testing.assert_allclose(st["l0.k"][:, rows], g["l0.k_proj"], **tol)
    np.testing.assert_allclose(st["l0.v"][:, rows], g["l0.v_proj"], **tol)
    np.testing.assert_allclose(st["l0.attn"][:, rows], g["l0.attn_ctx"], **tol)
    np.testing.assert_allclose(st["l0.out"][:, rows], g["l0.out"], **tol)
    np.testing.assert_allclose(st["l1.out"][:, rows], g["l1.out"], **tol)
    np.testing.assert_allclose(out[:, rows], g["final"], **tol)
    np.testing.assert_allclose(np.abs(out).mean(axis=(1, 2)), g["final_mean_abs"], rtol=1e-4)


def test_small_encoder_bf16_emulation_is_close(golden):
    """The bf16-operand emulation (what the MI355X throughput path computes) stays
    within a few 1e-2 of the fp32 result on unit-scale outputs."""
    g = golden("encoder_small.npz")
    cfg = encoder.EncCfg(128, 2, 2, 512)
    sd = synth.encoder_state_dict(128, 2, 2, 512, seed=3)
    mel = logmel.log_mel(synth.strain_segments(2, seed=21))
    out = encoder.encoder_forward(sd, mel, cfg, dtype=np.float32, emulate_bf16=True)
    err = np.abs(out[:, g["rows"]] - g["final"]).max()
    assert err < 5e-2, err


def test_rejects_wrong_length():
    cfg = encoder.EncCfg(128, 2, 2, 512)
    sd = synth.encoder_state_dict(128, 2, 2, 512, seed=3)
    with pytest.raises(ValueError, match="3000"):
        encoder.encoder_forward(sd, np.zeros((1, 80, 2999), np.float32), cfg)


@pytest.mark.skipif(not os.path.exists(os.path.join(GOLDEN, "config1.npz")), reason="config1 golden absent")
def test_config1_last_token_and_logits(golden):
    """BASELINE config 1 on a subset (the oracle runs whisper-tiny at ~1 s/segment):
    logits <= 1e-3 of the reference two/one-channel classifiers, labels exact."""
    g = golden("config1.npz")
    cfg = encoder.EncCfg.named("tiny")
    sd = synth.named_encoder_state_dict("tiny", seed=0)
    n = 64
    h1 = synth.strain_segments(n, seed=0)
    l1 = synth.strain_segments(n, seed=1)
    t = np.arange(16000, dtype=np.float32) / 16000.0
    for i in range(0, n, 2):
        s = (3.0 * np.sin(2 * np.pi * (40.0 + 200.0 * t * (1 + 0.05 * i)) * t) * np.exp(-((t - 0.6) / 0.15) ** 2))
        h1[i] += s.astype(np.float32)
        l1[i] += s.astype(np.float32)
    sub = [0, 1, 30, 63]
    la = encoder.last_token(sd, logmel.log_mel(h1[sub]), cfg)
    lb = encoder.last_token(sd, logmel.log_mel(l1[sub]), cfg)
    np.testing.assert_allclose(la, g["last_token"][sub, 0], atol=2e-4, rtol=1e-4)
    np.testing.assert_allclose(lb, g["last_token"][sub, 1], atol=2e-4, rtol=1e-4)
    head2 = synth.head_state_dict([768, 1024, 512, 256, 1], seed=0)
    head2["6.bias"] = head2["6.bias"] + g["two_channel_bias_shift"]
    head1 = synth.head_state_dict([384, 512, 256, 128, 64, 1], seed=1)
    head1["8.bias"] = head1["8.bias"] + g["one_channel_bias_shift"]
    lg2 = heads.two_channel_logits(la, lb, head2)
    lg1 = heads.one_channel_logits(lb, head1)
    np.testing.assert_allclose(lg2, g["two_channel_logits"][sub], atol=1e-3, rtol=0)
    np.testing.assert_allclose(lg1, g["one_channel_logits"][sub], atol=1e-3, rtol=0)
    np.testing.assert_array_equal(heads.binary_labels(lg2), g["two_channel_labels"][sub])
    np.testing.assert_array_equal(heads.binary_labels(lg1), g["one_channel_labels"][sub])
    # heads alone, on ALL 64 golden pooled vectors
    all2 = heads.two_channel_logits(g["last_token"][:, 0], g["last_token"][:, 1], head2)
    np.testing.assert_allclose(all2, g["two_channel_logits"], atol=1e-4, rtol=0)
    np.testing.assert_array_equal(heads.binary_labels(all2), g["two_channel_labels"])


# ----------------------------------------------------------------------------- DoRA
def _dora_case(seed=0, d_out=48, d_in=40, r=8, trained=True):
    rng = np.random.default_rng(seed)
    W0 = rng.standard_normal((d_out, d_in)) / np.sqrt(d_in)
    A, B, m = synth.dora_adapter(d_out, d_in, r, W0.astype(np.float32), seed + 1, trained=trained)
    x = rng.standard_normal((17, d_in))
    b = rng.standard_normal(d_out) * 0.1
    return x, W0, b, A.astype(np.float64), B.astype(np.float64), m.astype(np.float64)


def test_dora_identity_at_init():
    x, W0, b, A, B, m = _dora_case(trained=False)
    m = np.linalg.norm(W0, axis=1)
    y = dora.dora_linear_unmerged(x, W0, b, A, B, m, 4.0)
    np.testing.assert_allclose(y, x @ W0.T + b, atol=1e-12)


def test_dora_merged_equals_unmerged():
    x, W0, b, A, B, m = _dora_case()
    y0 = dora.dora_linear_unmerged(x, W0, b, A, B, m, 4.0)
    y1 = dora.dora_linear_merged(x, W0, b, A, B, m, 4.0)
    np.testing.assert_allclose(y0, y1, atol=1e-12)
    y2 = dora.dora_linear_merged(x, W0, None, A, B, m, 4.0)     # k_proj: no bias
    np.testing.assert_allclose(y2, y1 - b, atol=1e-12)


def test_dora_grads_match_finite_differences():
    """Norm DETACHED (peft dora.py: weight_norm.detach()): perturb A/B/m in the
    numerator only."""
    x, W0, b, A, B, m = _dora_case(seed=5)
    s = 4.0
    rng = np.random.default_rng(9)
    dy = rng.standard_normal((x.shape[0], W0.shape[0]))
    n_fixed = dora.dora_weight_norm(W0, A, B, s)

    def f(A_, B_, m_, x_):
        Wp = W0 + s * (B_ @ A_)
        return ((x_ @ Wp.T) * (m_ / n_fixed)[None] * dy).sum()

    dA, dB, dm, dx = dora.dora_grads(x, dy, W0, A, B, m, s)
    eps = 1e-6
    for name, arr, grad in (("A", A, dA), ("B", B, dB), ("m", m, dm), ("x", x, dx)):
        idxs = [tuple(rng.integers(0, n) for n in arr.shape) for _ in range(5)]
        for idx in idxs:
            ap = arr.copy(); ap[idx] += eps
            am = arr.copy(); am[idx] -= eps
            args_p = {"A": A, "B": B, "m": m, "x": x}; args_p[name] = ap
            args_m = {"A": A, "B": B, "m": m, "x": x}; args_m[name] = am
            fd = (f(args_p["A"], args_p["B"], args_p["m"], args_p["x"])
                  - f(args_m["A"], args_m["B"], args_m["m"], args_m["x"])) / (2 * eps)
            assert abs(fd - grad[idx]) < 1e-5 * max(1.0, abs(fd)), (name, idx, fd, grad[idx])


def test_adapter_schema_fixture():
    """The DoRA adapter the reference ships: 8 modules x {lora_A [8,384], lora_B [384,8],
    lora_magnitude_vector [384]}, f32, k_proj + v_proj only."""
    with open(os.path.join(GOLDEN, "adapter_schema.json")) as f:
        sc = json.load(f)
    t = sc["adapter_model.safetensors"]
    assert len(t) == 24
    for i in range(4):
        for proj in ("k_proj", "v_proj"):
            base = f"base_model.model.layers.{i}.self_attn.{proj}."
            assert t[base + "lora_A.weight"]["shape"] == [8, 384]
            assert t[base + "lora_B.weight"]["shape"] == [384, 8]
            assert t[base + "lora_magnitude_vector"]["shape"] == [384]
    cfg = sc["adapter_config.json"]
    assert cfg["use_dora"] is True and cfg["r"] == 8 and cfg["lora_alpha"] == 32
    assert cfg["peft_type"] == "LORA"


def test_torch_cpu_restatement_matches_hf_golden(golden):
    """oracle/encoder_torch.py (bench.py's cpu_baseline leg) is pinned by the same HF golden as the numpy oracle."""
    from oracle import encoder_torch
    g = golden("encoder_small.npz")
    sd = synth.encoder_state_dict(128, 2, 2, 512, seed=3)
    mel = logmel.log_mel(synth.strain_segments(2, seed=21))
    out = encoder_torch.encoder_forward(sd, mel, encoder.EncCfg(128, 2, 2, 512), chunk=1)
    np.testing.assert_allclose(out[:, g["rows"]], g["final"], atol=5e-5, rtol=1e-4)
    np.testing.assert_allclose(np.abs(out).mean(axis=(1, 2)), g["final_mean_abs"], rtol=1e-4)
