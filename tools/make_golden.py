#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REAL reference stack.

Runs only in the build container (needs /root/reference and ``transformers``):
  * HuggingFace ``WhisperFeatureExtractor``  (what Signal_vs_Noise/src/dataset.py:20-21 calls)
  * HuggingFace ``WhisperEncoder``           (what Signal_vs_Noise/src/train.py:227-228 builds)
  * the reference's own ``Signal_vs_Noise/src/model.py`` wrappers (:4-52)

Inputs and weights are regenerated from seeds by ``gw_whisper_amd.synth`` (numpy
PCG64), so only OUTPUTS are stored.  Nothing from the reference is copied: the
fixtures are arrays of numbers.

    python tools/make_golden.py            # writes tests/golden/*.npz
"""

from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
sys.path.insert(0, os.path.join(REF, "Signal_vs_Noise"))

from transformers import WhisperConfig, WhisperFeatureExtractor  # noqa: E402
from transformers.models.whisper.modeling_whisper import WhisperEncoder  # noqa: E402

from gw_whisper_amd import synth  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
ROWS = np.array([0, 1, 2, 49, 50, 51, 52, 53, 700, 1498, 1499])


def hf_encoder(d, L, H, ffn, sd, attn="eager"):
    cfg = WhisperConfig(d_model=d, encoder_layers=L, encoder_attention_heads=H, encoder_ffn_dim=ffn,
                        decoder_layers=1, decoder_attention_heads=H, decoder_ffn_dim=ffn,
                        attn_implementation=attn)
    enc = WhisperEncoder(cfg)
    missing = enc.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return enc.eval()


def make_logmel():
    fe = WhisperFeatureExtractor()
    out = {}
    # (a) four 1 s segments, the hot-path shape
    seg = synth.strain_segments(4, seed=11)
    f = fe([s for s in seg], sampling_rate=16000, return_tensors="np").input_features
    out["seg16000_frames0_112"] = f[:, :, :112].astype(np.float32)
    out["seg16000_pad_value"] = f[:, 0, 2999].astype(np.float32)
    assert np.all(f[:, :, 103:] == f[:, :1, 2999:3000]), "frames >= 103 must be one constant"
    # per-item call (python list of floats, like the Arrow rows) must equal the batched call
    f1 = fe(seg[0].tolist(), sampling_rate=16000, return_tensors="pt").input_features.numpy()
    assert np.array_equal(f1[0], f[0])
    # (b) ragged lengths incl. tiny, non-multiple-of-hop and > 1 s
    for n in (1, 159, 12345, 40000):
        w = synth.strain_segments(1, seed=100 + n, n_samples=n)[0]
        g = fe(w, sampling_rate=16000, return_tensors="np").input_features[0]
        live = min(3000, -(-(n + 200) // 160))
        out[f"len{n}_frames"] = g[:, :live + 2].astype(np.float32)
        out[f"len{n}_pad_value"] = g[0, 2999].astype(np.float32)
    # (c) a full 30 s buffer and an over-long one (truncated): reflect padding at the right edge
    for n in (480000, 480321):
        w = synth.strain_segments(1, seed=200 + n, n_samples=n)[0]
        g = fe(w, sampling_rate=16000, return_tensors="np").input_features[0]
        cols = np.concatenate([np.arange(0, 3000, 37), np.arange(2990, 3000)])
        out[f"len{n}_cols"] = cols
        out[f"len{n}_frames"] = g[:, cols].astype(np.float32)
    # (d) amplitude extremes: all-zero input and 1e-21-scale raw strain collapse to constants
    z = fe(np.zeros(16000, np.float32), sampling_rate=16000, return_tensors="np").input_features[0]
    out["zeros_value"] = np.array([z.min(), z.max()], np.float32)
    r = fe((synth.strain_segments(1, seed=5)[0] * 1e-21).astype(np.float32), sampling_rate=16000,
           return_tensors="np").input_features[0]
    out["raw1e21_value"] = np.array([r.min(), r.max()], np.float32)
    np.savez_compressed(os.path.join(GOLD, "logmel.npz"), **out)
    print("logmel.npz", {k: v.shape for k, v in out.items()})


def make_encoder_small():
    """Reduced config (d=128, H=2, L=2, ffn=512): per-stage activations on selected rows."""
    d, L, H, ffn = 128, 2, 2, 512
    sd = synth.encoder_state_dict(d, L, H, ffn, seed=3)
    enc = hf_encoder(d, L, H, ffn, sd)
    fe = WhisperFeatureExtractor()
    mel = fe([s for s in synth.strain_segments(2, seed=21)], sampling_rate=16000, return_tensors="pt").input_features
    got = {}

    def save(name, fn=lambda o: o):
        def hook(mod, inp, outp):
            got[name] = fn(outp).detach().numpy()
        return hook

    hs = [
        enc.conv1.register_forward_hook(save("conv1_pre_gelu", lambda o: o.permute(0, 2, 1))),
        enc.layers[0].register_forward_pre_hook(lambda m, i: got.__setitem__("embed", i[0].detach().numpy())),
        enc.layers[0].self_attn.q_proj.register_forward_hook(save("l0.q_proj")),
        enc.layers[0].self_attn.k_proj.register_forward_hook(save("l0.k_proj")),
        enc.layers[0].self_attn.v_proj.register_forward_hook(save("l0.v_proj")),
        enc.layers[0].self_attn.out_proj.register_forward_pre_hook(
            lambda m, i: got.__setitem__("l0.attn_ctx", i[0].detach().numpy())),
        enc.layers[0].fc1.register_forward_hook(save("l0.fc1_pre_gelu")),
        enc.layers[0].register_forward_hook(save("l0.out", lambda o: o[0] if isinstance(o, tuple) else o)),
        enc.layers[1].register_forward_hook(save("l1.out", lambda o: o[0] if isinstance(o, tuple) else o)),
    ]
    with torch.no_grad():
        final = enc(mel).last_hidden_state.numpy()
    for h in hs:
        h.remove()
    out = {"rows": ROWS, "final": final[:, ROWS]}
    for k, v in got.items():
        rows = ROWS * 2 if k == "conv1_pre_gelu" else ROWS
        out[k] = v[:, rows]
    out["final_mean_abs"] = np.abs(final).mean(axis=(1, 2))
    np.savez_compressed(os.path.join(GOLD, "encoder_small.npz"), **out)
    print("encoder_small.npz", {k: v.shape for k, v in out.items()})


def make_config1():
    """BASELINE config 1: 64 two-detector segments, whisper-tiny encoder, reference
    two_channel_ligo_binary_classifier on CPU."""
    from src.model import one_channel_ligo_binary_classifier, two_channel_ligo_binary_classifier

    d, L, H, ffn = synth.ENCODER_SIZES["tiny"]
    sd = synth.encoder_state_dict(d, L, H, ffn, seed=0)
    enc = hf_encoder(d, L, H, ffn, sd)
    fe = WhisperFeatureExtractor()
    n = 64
    h1 = synth.strain_segments(n, seed=0)
    l1 = synth.strain_segments(n, seed=1)
    # make half of them "signal-like": a loud chirp-ish sinusoid common to both detectors
    t = np.arange(16000, dtype=np.float32) / 16000.0
    for i in range(0, n, 2):
        s = (3.0 * np.sin(2 * np.pi * (40.0 + 200.0 * t * (1 + 0.05 * i)) * t) * np.exp(-((t - 0.6) / 0.15) ** 2))
        h1[i] += s.astype(np.float32)
        l1[i] += s.astype(np.float32)
    head2 = synth.head_state_dict([2 * d, 1024, 512, 256, 1], seed=0)
    head1 = synth.head_state_dict([d, 512, 256, 128, 64, 1], seed=1)
    m2 = two_channel_ligo_binary_classifier(enc).eval()
    m1 = one_channel_ligo_binary_classifier(enc).eval()
    m2.classifier.load_state_dict({k: torch.from_numpy(v) for k, v in head2.items()})
    m1.classifier.load_state_dict({k: torch.from_numpy(v) for k, v in head1.items()})
    logits2, logits1, last = [], [], []
    with torch.no_grad():
        for i in range(0, n, 8):
            a = fe([s for s in h1[i:i + 8]], sampling_rate=16000, return_tensors="pt").input_features
            b = fe([s for s in l1[i:i + 8]], sampling_rate=16000, return_tensors="pt").input_features
            logits2.append(m2(a, b).numpy())
            logits1.append(m1(b).numpy())
            last.append(np.stack([enc(a).last_hidden_state[:, -1, :].numpy(),
                                  enc(b).last_hidden_state[:, -1, :].numpy()], axis=1))
            print("config1 batch", i, flush=True)
    logits2 = np.concatenate(logits2)
    logits1 = np.concatenate(logits1)
    last = np.concatenate(last)
    # centre the synthetic heads so that sigmoid().round() gives mixed labels
    b2 = -np.median(logits2)
    b1 = -np.median(logits1)
    out = {
        "last_token": last.astype(np.float32),                       # [64, 2, 384]
        "two_channel_bias_shift": np.float32(b2),
        "one_channel_bias_shift": np.float32(b1),
        "two_channel_logits": (logits2 + b2).astype(np.float32),     # [64, 1]
        "one_channel_logits": (logits1 + b1).astype(np.float32),
    }
    out["two_channel_labels"] = torch.sigmoid(torch.from_numpy(out["two_channel_logits"])).round().numpy().astype(np.int64)
    out["one_channel_labels"] = torch.sigmoid(torch.from_numpy(out["one_channel_logits"])).round().numpy().astype(np.int64)
    np.savez_compressed(os.path.join(GOLD, "config1.npz"), **out)
    print("config1.npz", {k: np.shape(v) for k, v in out.items()})
    print("two-channel logits: min|.|", np.abs(out["two_channel_logits"]).min(), "std", out["two_channel_logits"].std(),
          "labels", out["two_channel_labels"].sum())


def make_adapter_schema():
    """Key names / shapes / dtypes of the DoRA adapter and head files the reference ships
    (format fixtures: SURVEY.md section 4)."""
    from safetensors import safe_open

    base = os.path.join(REF, "Signal_vs_Noise", "results", "Two_detectors", "models", "best_lora_weights")
    schema = {"adapter_model.safetensors": {}, "adapter_config.json": None, "heads": {}}
    with safe_open(os.path.join(base, "adapter_model.safetensors"), "np") as f:
        for k in f.keys():
            t = f.get_tensor(k)
            schema["adapter_model.safetensors"][k] = {"shape": list(t.shape), "dtype": str(t.dtype)}
    with open(os.path.join(base, "adapter_config.json")) as f:
        schema["adapter_config.json"] = json.load(f)
    for rel in ("Signal_vs_Noise/results/Single_detector/models/best_dense_layers.pth",
                "Glitch_classification/results/generic/multi_class_model_best_dense_weights.pth",
                "Glitch_classification/results/high_mass/multi_class_model_best_dense_weights.pth"):
        p = os.path.join(REF, rel)
        if os.path.exists(p):
            sd = torch.load(p, map_location="cpu", weights_only=True)
            schema["heads"][rel] = {k: list(v.shape) for k, v in sd.items()}
    with open(os.path.join(GOLD, "adapter_schema.json"), "w") as f:
        json.dump(schema, f, indent=1, sort_keys=True)
    print("adapter_schema.json", len(schema["adapter_model.safetensors"]), "tensors;", list(schema["heads"]))


def _reference_defs(rel_path, names, extra_ns=None):
    """Execute selected top-level definitions of a reference file WITHOUT importing the module (its module-level
    imports need h5py / pycbc / ml4gw / peft, absent here): the definitions are parsed out of the file where it lies
    and compiled as they are.  Build-container only; nothing of the text is stored."""
    import ast
    import logging
    from typing import Any, Dict, List, Optional, Tuple
    path = os.path.join(REF, rel_path)
    tree = ast.parse(open(path).read(), filename=path)
    keep = [n for n in tree.body if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in names]
    assert sorted(n.name for n in keep) == sorted(names), [n.name for n in keep]
    ns = {"np": np, "torch": torch, "nn": torch.nn, "logging": logging, "Any": Any, "Dict": Dict, "List": List,
          "Optional": Optional, "Tuple": Tuple, "__name__": "reference_defs"}
    ns.update(extra_ns or {})
    mod = ast.Module(body=[ast.ImportFrom(module="__future__", names=[ast.alias(name="annotations")], level=0)] + keep,
                     type_ignores=[])
    code = compile(ast.fix_missing_locations(mod), path, "exec")
    exec(code, ns)
    return ns


class _FakeDataset:
    """Duck-typed stand-in for an h5py dataset (test INPUT, not reference code): ds[()] and ds.attrs."""
    def __init__(self, data, attrs):
        self._d, self.attrs, self.dtype = data, attrs, data.dtype
    def __getitem__(self, k):
        return self._d[k]


def make_inference_host():
    """Time stamps, windows, triggers and clusters from the reference's OWN SegmentSlicer / evaluate_slices /
    get_clusters (MLGWSC-1/inference.py:140-166, 173-296, 454-489), run here on seeded strain."""
    from torch.utils.data import DataLoader, IterableDataset
    from tqdm import tqdm
    ns = _reference_defs("MLGWSC-1/inference.py", ["get_clusters", "SegmentSlicer", "TorchSegmentSlicer", "evaluate_slices"],
                         {"IterableDataset": IterableDataset, "DataLoader": DataLoader, "tqdm": tqdm, "h5py": None})
    out = {}
    # (a) a long segment: >= 1e5 windows, GPS-like start time, the stored delta_t attribute of the reference's files
    dt_attr = 1.0 / 2048
    start = np.float64(1238166018.0)
    n_long = 204 * 120000 + 2048
    long_data = np.zeros((2, n_long), np.float32)            # the time stamps do not depend on the samples
    f = {det: {"k": _FakeDataset(long_data[i], {"delta_t": dt_attr, "start_time": start})} for i, det in enumerate(["H1", "L1"])}
    sl = ns["SegmentSlicer"](f, "k", white=True)
    n = len(sl)
    it = iter(sl)
    ts = np.empty(n, np.float64)
    for i in range(n):
        ts[i] = it.get_next_slice()[1]
    idx = np.unique(np.concatenate([np.arange(0, 200), np.arange(0, n, 997), np.arange(n - 200, n)]))
    out["long_n_samples"], out["long_n_windows"] = np.int64(n_long), np.int64(n)
    out["long_start"], out["long_delta_t_attr"] = start, np.float64(dt_attr)
    out["long_idx"], out["long_times"] = idx, ts[idx]
    out["long_times_xor"] = np.bitwise_xor.reduce(ts.view(np.uint64))
    out["long_times_sum_u64"] = np.add.reduce(ts.view(np.uint64))     # wraps: a second, order-free checksum
    # (a') a sample rate that is not a power of two (4000 Hz): the running sum rounds at every step there
    dt_odd = 1.0 / 4000.0
    n_odd = 400 * 50000 + 2048
    odd = np.zeros((2, n_odd), np.float32)
    f = {det: {"k": _FakeDataset(odd[i], {"delta_t": dt_odd, "start_time": np.float64(1238166018.3)})} for i, det in enumerate(["H1", "L1"])}
    sl = ns["SegmentSlicer"](f, "k", white=True)
    it = iter(sl)
    to = np.array([it.get_next_slice()[1] for _ in range(len(sl))], np.float64)
    io = np.unique(np.concatenate([np.arange(0, 100), np.arange(0, len(sl), 499), np.arange(len(sl) - 100, len(sl))]))
    out["odd_n_samples"], out["odd_n_windows"], out["odd_delta_t_attr"] = np.int64(n_odd), np.int64(len(sl)), np.float64(dt_odd)
    out["odd_idx"], out["odd_times"] = io, to[io]
    out["odd_times_xor"] = np.bitwise_xor.reduce(to.view(np.uint64))
    out["odd_index_step"] = np.int64(sl.index_step_size)
    # (b) a short segment through the reference's evaluate_slices with a deterministic network
    strain = synth.strain_segments(2, seed=77, n_samples=2048 * 40)
    f = {det: {"seg": _FakeDataset(strain[i], {"delta_t": dt_attr, "start_time": np.float64(1000.25)})}
         for i, det in enumerate(["H1", "L1"])}
    tsl = ns["TorchSegmentSlicer"](f, "seg", white=True)
    from tests.helpers import search_toy_network
    net = search_toy_network()
    trig, vals = ns["evaluate_slices"](tsl, net, device="cpu", trigger_threshold=0.5)
    out["short_n_windows"] = np.int64(len(tsl))
    out["short_triggers"] = np.array(trig, np.float64).reshape(-1, 2)
    out["short_scores"] = np.concatenate(vals).astype(np.float32)
    out["short_window_7"] = tsl.dss[:, 7 * tsl.index_step_size: 7 * tsl.index_step_size + 2048].astype(np.float32)
    # (c) get_clusters on the triggers above plus a synthetic two-key set with gaps around the 0.35 s threshold
    rng = np.random.default_rng(5)
    gaps = rng.choice([0.1, 0.2, 0.30000000000000004, 0.35, 0.35000000000000003, 0.4, 1.0], size=400)
    t2 = 50.0 + np.add.accumulate(gaps)
    v2 = rng.random(400)
    trig2 = {"a": [[float(a), float(b)] for a, b in zip(t2[:250], v2[:250])],
             "b": [[float(a), float(b)] for a, b in zip(t2[250:], v2[250:])], "empty": []}
    out["cl_in_times"], out["cl_in_vals"] = t2, v2
    for name, tr in (("short", {"seg": trig}), ("two", trig2)):
        t, v, tv = ns["get_clusters"](tr, cluster_threshold=0.35)
        out[f"cl_{name}_times"], out[f"cl_{name}_vals"], out[f"cl_{name}_tvars"] = t, v, tv
    np.savez_compressed(os.path.join(GOLD, "inference_host.npz"), **out)
    print("inference_host.npz", {k: np.shape(v) for k, v in out.items()}, "triggers", len(trig))


def make_config4():
    """BASELINE config 4: the reference's Glitch_classification/src/model.py (:4-39, imports only torch) on an HF
    whisper-base encoder, num_classes = 22, 12 seeded segments through the log-mel front end the Glitch code uses
    (Glitch_classification/src/dataset.py:46)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("glitch_model", os.path.join(REF, "Glitch_classification", "src", "model.py"))
    gm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gm)
    d, L, H, ffn = synth.ENCODER_SIZES["base"]
    sd = synth.encoder_state_dict(d, L, H, ffn, seed=4)
    enc = hf_encoder(d, L, H, ffn, sd)
    model = gm.one_channel_ligo_binary_classifier(enc, num_classes=22).eval()      # eval(): Dropout is the identity
    head = synth.head_state_dict([d, 512, 256, 128, 22], seed=769, sequential_stride=3)   # of 1000 seeds: 5 labels, largest margin
    model.classifier.load_state_dict({k: torch.from_numpy(v) for k, v in head.items()})
    fe = WhisperFeatureExtractor()
    n = 12
    seg = synth.strain_segments(n, seed=44)
    t = np.arange(16000, dtype=np.float32) / 16000.0
    for i in range(n):      # glitch-like bursts of different frequency / time / loudness (SNR ~ 1 ... 300: the loud ones
        # move the log-mel clamp, i.e. the value of all 2900 padded frames) so that the classes differ
        amp = 0.5 * (400.0 ** (i / (n - 1)))
        seg[i] += (amp * np.sin(2 * np.pi * (30.0 + 35.0 * i) * t) * np.exp(-((t - 0.1 - 0.07 * i) / 0.03) ** 2)).astype(np.float32)
    logits, last = [], []
    with torch.no_grad():
        for i in range(0, n, 4):
            mel = fe([x for x in seg[i:i + 4]], sampling_rate=16000, return_tensors="pt").input_features
            logits.append(model(mel).numpy())
            last.append(enc(mel).last_hidden_state[:, -1, :].numpy())
            print("config4 batch", i, flush=True)
    logits = np.concatenate(logits).astype(np.float32)
    # centre every class of the synthetic head (as make_config1 does) so that argmax gives mixed labels: the shift
    # is added to the bias of the last Linear by the test
    shift = (-np.mean(logits, axis=0)).astype(np.float32)
    logits = logits + shift[None, :]
    out = {"logits": logits, "labels": logits.argmax(1).astype(np.int64), "last_token": np.concatenate(last).astype(np.float32),
           "class_bias_shift": shift}
    srt = np.sort(logits, axis=1)
    out["top2_margin"] = (srt[:, -1] - srt[:, -2]).astype(np.float32)
    np.savez_compressed(os.path.join(GOLD, "config4.npz"), **out)
    print("config4.npz", {k: v.shape for k, v in out.items()}, "labels", out["labels"], "min margin", out["top2_margin"].min())


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    torch.manual_seed(0)
    which = sys.argv[1:] or ["logmel", "encoder_small", "config1", "adapter_schema", "inference_host", "config4"]
    for w in which:
        {"logmel": make_logmel, "encoder_small": make_encoder_small, "config1": make_config1,
         "adapter_schema": make_adapter_schema, "inference_host": make_inference_host,
         "config4": make_config4}[w]()
