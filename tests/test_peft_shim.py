"""peft stand-in: runtime parameter names, trainable-parameter selection and the on-disk
adapter schema against the adapters the reference ships (tests/golden/adapter_schema.json).
CPU only (no kernels run)."""

import fnmatch
import json
import os

import pytest
import torch

from gw_whisper_amd.encoder import WhisperConfig, WhisperEncoder
from gw_whisper_amd.peft import LoraConfig, PeftModel, get_peft_model

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _targets(enc, pats):
    return [n for n, _ in enc.named_modules() if any(fnmatch.fnmatch(n, p) for p in pats)]


def test_reference_recipe_names_and_counts():
    """Signal_vs_Noise/src/train.py:230-237,263-267: fnmatch targets, get_peft_model,
    requires_grad = 'lora' in name  ->  12 modules, 78 336 trainable parameters on tiny."""
    enc = WhisperEncoder(WhisperConfig.named("tiny"))
    pats = ["layers.*.self_attn.q_proj", "layers.*.self_attn.k_proj", "layers.*.self_attn.v_proj",
            "layers.*.self_attn.o_proj"]
    targets = _targets(enc, pats)
    assert len(targets) == 12
    peft = get_peft_model(enc, LoraConfig(use_dora=True, r=8, lora_alpha=32, target_modules=targets))
    for name, p in peft.named_parameters():
        p.requires_grad = "lora" in name
    names = dict(peft.named_parameters())
    assert "base_model.model.layers.0.self_attn.k_proj.base_layer.weight" in names
    assert "base_model.model.layers.0.self_attn.k_proj.lora_A.default.weight" in names
    assert "base_model.model.layers.0.self_attn.k_proj.lora_B.default.weight" in names
    assert "base_model.model.layers.0.self_attn.k_proj.lora_magnitude_vector.default.weight" in names
    trainable = sum(p.numel() for p in peft.parameters() if p.requires_grad)
    assert trainable == 78336
    assert peft.config.d_model == 384          # src/model.py:11 reads encoder.config.d_model
    # identity at init: B = 0 and m = ||W0||
    lin = peft.base_model.model.layers[0].self_attn.k_proj
    assert torch.count_nonzero(lin.lora_B["default"].weight) == 0
    torch.testing.assert_close(lin.lora_magnitude_vector["default"].weight,
                               torch.linalg.norm(lin.base_layer.weight, dim=1))


def test_save_pretrained_matches_shipped_schema(tmp_path):
    with open(os.path.join(GOLDEN, "adapter_schema.json")) as f:
        shipped = json.load(f)
    enc = WhisperEncoder(WhisperConfig.named("tiny"))
    targets = shipped["adapter_config.json"]["target_modules"]       # k_proj + v_proj of 4 layers
    peft = get_peft_model(enc, LoraConfig(use_dora=True, r=8, lora_alpha=32, target_modules=targets,
                                          base_model_name_or_path="openai/whisper-tiny"))
    peft.save_pretrained(str(tmp_path))
    from safetensors import safe_open
    with safe_open(str(tmp_path / "adapter_model.safetensors"), "pt") as f:
        got = {k: {"shape": list(f.get_tensor(k).shape), "dtype": str(f.get_tensor(k).dtype).replace("torch.", "")}
               for k in f.keys()}
    assert got == shipped["adapter_model.safetensors"]
    cfg = json.load(open(tmp_path / "adapter_config.json"))
    ref = shipped["adapter_config.json"]
    assert set(cfg) == set(ref), set(cfg) ^ set(ref)
    for k in ref:
        if k == "target_modules":
            assert sorted(cfg[k]) == sorted(ref[k])
        else:
            assert cfg[k] == ref[k], k


def test_from_pretrained_round_trip(tmp_path):
    torch.manual_seed(0)
    enc = WhisperEncoder(WhisperConfig(128, 2, 2, 512))
    targets = _targets(enc, ["layers.*.self_attn.k_proj", "layers.*.self_attn.v_proj"])
    peft = get_peft_model(enc, LoraConfig(use_dora=True, r=4, lora_alpha=16, target_modules=targets))
    with torch.no_grad():
        for n, p in peft.named_parameters():
            if "lora_" in n:
                p.add_(torch.randn_like(p) * 0.1)
    peft.save_pretrained(str(tmp_path))
    enc2 = WhisperEncoder(WhisperConfig(128, 2, 2, 512))
    enc2.load_state_dict(enc.state_dict() if False else {k.replace(".base_layer", ""): v for k, v in enc.state_dict().items()
                                                         if "lora_" not in k})
    peft2 = PeftModel.from_pretrained(enc2, str(tmp_path))
    a, b = peft.state_dict(), peft2.state_dict()
    assert a.keys() == b.keys()
    for k in a:
        torch.testing.assert_close(a[k], b[k])
    assert not any(p.requires_grad for n, p in peft2.named_parameters() if "lora_" in n)   # inference mode


def test_unknown_target_raises():
    enc = WhisperEncoder(WhisperConfig(128, 1, 2, 512))
    with pytest.raises(ValueError, match="not found"):
        get_peft_model(enc, LoraConfig(use_dora=True, target_modules=["layers.0.self_attn.o_proj"]))


REF_ADAPTER = "/root/reference/Signal_vs_Noise/results/Two_detectors/models/best_lora_weights"


@pytest.mark.skipif(not os.path.isdir(REF_ADAPTER), reason="reference tree absent (GPU box): build-container pin only")
def test_the_adapter_the_reference_ships_loads_and_round_trips(tmp_path):
    """The REAL ``adapter_model.safetensors`` / ``adapter_config.json`` the authors publish (trained with peft 0.12.0),
    through ``PeftModel.from_pretrained`` exactly as ``Signal_vs_Noise/src/train.py:50`` does: every tensor lands in the
    module peft's key names, ``save_pretrained`` writes back a file with the same keys, shapes, dtypes AND bytes, and
    the config round-trips field by field."""
    from safetensors import safe_open
    enc = WhisperEncoder(WhisperConfig.named("tiny"))
    peft = PeftModel.from_pretrained(enc, REF_ADAPTER)
    with safe_open(os.path.join(REF_ADAPTER, "adapter_model.safetensors"), "pt") as f:
        shipped = {k: f.get_tensor(k) for k in f.keys()}
    assert len(shipped) == 24
    sd = peft.state_dict()
    for k, v in shipped.items():
        rk = (k + ".default.weight") if k.endswith("lora_magnitude_vector") else \
            k.replace(".lora_A.weight", ".lora_A.default.weight").replace(".lora_B.weight", ".lora_B.default.weight")
        assert rk in sd, rk
        assert torch.equal(sd[rk], v)
    # only k_proj / v_proj of the 4 layers are wrapped; q_proj / out_proj stay plain Linear; nothing trainable
    wrapped = [n for n, m in peft.named_modules() if hasattr(m, "lora_A")]
    assert len(wrapped) == 8 and all(n.endswith(("k_proj", "v_proj")) for n in wrapped)
    assert not any(p.requires_grad for p in peft.parameters())
    # the trained adapters are not the identity: B != 0 and m moved away from its init
    lin = peft.base_model.model.layers[0].self_attn.v_proj
    assert torch.count_nonzero(lin.lora_B["default"].weight) > 0
    peft.save_pretrained(str(tmp_path))
    with safe_open(str(tmp_path / "adapter_model.safetensors"), "pt") as f:
        again = {k: f.get_tensor(k) for k in f.keys()}
    assert again.keys() == shipped.keys()
    for k in shipped:
        assert again[k].dtype == shipped[k].dtype and torch.equal(again[k], shipped[k]), k
    ref_cfg = json.load(open(os.path.join(REF_ADAPTER, "adapter_config.json")))
    cfg = json.load(open(tmp_path / "adapter_config.json"))
    assert set(cfg) == set(ref_cfg)
    for k in ref_cfg:
        assert (sorted(cfg[k]) == sorted(ref_cfg[k])) if k == "target_modules" else (cfg[k] == ref_cfg[k]), k


def test_datasets_directory_round_trip_through_the_harness_loader(tmp_path):
    """The reference's training data are HF ``datasets`` directories with the columns h1_timeseries / l1_timeseries /
    labels / injection_snr (Signal_vs_Noise/src/dataset.py:18, utils/preprocess.py:111-132), optionally split into
    ``chunk*`` sub-directories (run_train.py:73-82): ``harness/run_train.py::load_arrays`` reads both layouts."""
    import importlib.util
    import types
    import numpy as np
    from datasets import Dataset
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("run_train_h", os.path.join(root, "harness", "run_train.py"))
    rt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rt)
    rng = np.random.default_rng(0)
    cols = {"h1_timeseries": rng.standard_normal((10, 16000)).astype(np.float32),
            "l1_timeseries": rng.standard_normal((10, 16000)).astype(np.float32),
            "labels": (np.arange(10) % 2).astype(np.float32), "injection_snr": rng.uniform(5, 20, 10).astype(np.float32)}
    as_rows = {k: [r.tolist() for r in v] if v.ndim == 2 else v.tolist() for k, v in cols.items()}   # Arrow float lists
    Dataset.from_dict(as_rows).save_to_disk(str(tmp_path / "whole"))
    for lo, hi, name in ((0, 6, "chunk0"), (6, 10, "chunk1")):
        Dataset.from_dict({k: v[lo:hi] for k, v in as_rows.items()}).save_to_disk(str(tmp_path / "chunked" / name))
    for path in ("whole", "chunked"):
        got = rt.load_arrays(types.SimpleNamespace(synthetic=0, data_path=str(tmp_path / path)))
        for g, k in zip(got, ("h1_timeseries", "l1_timeseries", "labels", "injection_snr")):
            assert g.dtype == np.float32 and np.array_equal(g, cols[k]), (path, k)


def test_cached_parameter_lists_follow_the_module_structure():
    """The encoder caches its parameter lists per structure epoch (encoder._EPOCH: the per-step host work of a training forward
    used to walk the module tree five times).  Attaching adapters, replacing a sub-module or re-assigning a parameter must
    invalidate the cache; requires_grad, data_ptr and _version are always read fresh."""
    from gw_whisper_amd import encoder as E, _lib
    enc = WhisperEncoder(WhisperConfig(d_model=128, encoder_layers=2, encoder_attention_heads=2, encoder_ffn_dim=256))
    named0 = enc._param_cache()[1]
    assert [n for n, _ in named0] == [n for n, _ in enc.named_parameters()]
    assert enc._param_cache() is enc._param_cache()                    # cached while nothing changes
    g0, l0 = enc._group_keys()
    with torch.no_grad():
        enc.layers[1].fc1.weight.add_(1.0)                              # an in-place update bumps _version: key of group 2 only
    g1, l1 = enc._group_keys()
    assert g1 == g0 and l1[0] == l0[0] and l1[1][2] != l0[1][2] and l1[1][0] == l0[1][0]
    targets = _targets(enc, ["layers.*.self_attn.q_proj", "layers.*.self_attn.v_proj"])
    peft = get_peft_model(enc, LoraConfig(use_dora=True, r=4, lora_alpha=8, target_modules=targets))
    names = [n for n, _ in enc._param_cache()[1]]
    assert names == [n for n, _ in enc.named_parameters()] and any("lora_A" in n for n in names)
    assert len(enc._group_keys()[1][0][0]) > len(l0[0][0])              # the wrappers' parameters joined the q/k/v group
    for n, p in peft.named_parameters():
        p.requires_grad = "lora" in n
    x = torch.zeros(1, 80, 3000)
    assert enc._wants_grad(x) and enc._has_trainable_adapters()
    with torch.no_grad():
        assert not enc._wants_grad(x)
    enc.layers[0].fc2.weight.requires_grad = True                       # a base weight un-frozen on purpose: refused, read fresh
    with pytest.raises(_lib.GwwError):
        enc._wants_grad(x)
    enc.layers[0].fc2.weight.requires_grad = False
    e = E._EPOCH[0]
    enc.layers[0].fc2 = torch.nn.Linear(256, 128)                       # a replaced sub-module: new epoch, new lists
    assert E._EPOCH[0] > e and enc._param_cache()[1][0][1] is not None
    assert any(p is enc.layers[0].fc2.weight for _, p in enc._param_cache()[1])
