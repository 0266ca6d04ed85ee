"""Functional stand-in for the ``peft`` names the reference uses:

    from peft import LoraConfig, get_peft_model, PeftModel, PeftConfig
    cfg = LoraConfig(use_dora=True, r=8, lora_alpha=32, target_modules=[...])      # src/train.py:263
    model = get_peft_model(encoder, cfg)                                           # src/train.py:264
    for n, p in model.named_parameters(): p.requires_grad = 'lora' in n            # src/train.py:266-267
    model.save_pretrained(dir)                                                     # src/train.py:135,196
    model = PeftModel.from_pretrained(encoder, dir)                                # src/train.py:50

Same runtime parameter names as peft 0.12.0 (``<module>.base_layer.weight``,
``<module>.lora_A.default.weight``, ``<module>.lora_B.default.weight``,
``<module>.lora_magnitude_vector.default.weight``) and the same on-disk schema as the
adapters the reference ships (``adapter_config.json`` + ``adapter_model.safetensors`` with
``base_model.model.<module>.lora_A.weight / .lora_B.weight / .lora_magnitude_vector``;
fixture ``tests/golden/adapter_schema.json``).

Arithmetic (peft 0.12.0 ``tuners/lora/dora.py``): ``W' = W0 + s B A``, ``n = ||W'||`` per
output row (detached), ``y = (m / n) (W' x) + b``.  peft recomputes ``n`` with an
identity-matrix matmul in EVERY forward; here the effective dense weight is merged once per
weight version by the HIP kernel ``gww_dora_merge_f32`` and handed to the encoder's packed
panels, so the adapted forward costs exactly the same as the un-adapted one.
"""

from __future__ import annotations

import json
import math
import os
from dataclasses import asdict, dataclass, field
from typing import List, Optional

import torch
import torch.nn as nn


@dataclass
class LoraConfig:
    r: int = 8
    lora_alpha: int = 8
    target_modules: Optional[List[str]] = None
    lora_dropout: float = 0.0
    use_dora: bool = False
    bias: str = "none"
    fan_in_fan_out: bool = False
    inference_mode: bool = False
    init_lora_weights: bool = True
    use_rslora: bool = False
    modules_to_save: Optional[List[str]] = None
    base_model_name_or_path: Optional[str] = None
    task_type: Optional[str] = None
    peft_type: str = "LORA"
    # carried through for schema compatibility with peft 0.12.0's adapter_config.json
    alpha_pattern: dict = field(default_factory=dict)
    rank_pattern: dict = field(default_factory=dict)
    loftq_config: dict = field(default_factory=dict)
    layer_replication: Optional[list] = None
    layers_pattern: Optional[list] = None
    layers_to_transform: Optional[list] = None
    megatron_config: Optional[dict] = None
    megatron_core: str = "megatron.core"
    revision: Optional[str] = None
    auto_mapping: Optional[dict] = None

    def __post_init__(self):
        if self.lora_dropout != 0.0:
            raise NotImplementedError("lora_dropout != 0 is not used by the reference (adapter_config.json: 0.0)")
        if self.bias != "none" or self.fan_in_fan_out or self.use_rslora or self.modules_to_save:
            raise NotImplementedError("only the LoRA/DoRA options the reference uses are implemented")
        if isinstance(self.target_modules, str):
            self.target_modules = [self.target_modules]

    @property
    def scaling(self) -> float:
        return self.lora_alpha / self.r

    def to_dict(self) -> dict:
        d = asdict(self)
        d["target_modules"] = list(self.target_modules or [])
        return d


PeftConfig = LoraConfig


class _Magnitude(nn.Module):
    """peft's ``DoraLinearLayer``: holds the magnitude as ``.weight`` ([out])."""

    def __init__(self, init: torch.Tensor):
        super().__init__()
        self.weight = nn.Parameter(init.clone(), requires_grad=True)


class DoraLinear(nn.Module):
    """LoRA / DoRA wrapper of an ``nn.Linear`` (peft ``tuners/lora/layer.py::Linear``)."""

    def __init__(self, base: nn.Linear, r: int, lora_alpha: int, use_dora: bool, adapter: str = "default"):
        super().__init__()
        self.base_layer = base
        self.in_features, self.out_features = base.in_features, base.out_features
        self.r, self.lora_alpha, self.use_dora = r, lora_alpha, use_dora
        self.scaling = lora_alpha / r
        self.adapter = adapter
        dev, dt = base.weight.device, base.weight.dtype
        self.lora_A = nn.ModuleDict({adapter: nn.Linear(self.in_features, r, bias=False, device=dev, dtype=dt)})
        self.lora_B = nn.ModuleDict({adapter: nn.Linear(r, self.out_features, bias=False, device=dev, dtype=dt)})
        nn.init.kaiming_uniform_(self.lora_A[adapter].weight, a=math.sqrt(5))
        nn.init.zeros_(self.lora_B[adapter].weight)
        self.lora_magnitude_vector = nn.ModuleDict()
        if use_dora:
            with torch.no_grad():   # m = ||W0 + s B A||_row with B = 0 -> identity at init
                self.lora_magnitude_vector[adapter] = _Magnitude(torch.linalg.norm(base.weight.detach(), dim=1))
        from . import encoder as _enc   # (a wrapper that reaches an encoder by any route invalidates its cached parameter lists)
        _enc._EPOCH[0] += 1

    # what peft exposes on the wrapper
    @property
    def weight(self):
        return self.base_layer.weight

    @property
    def bias(self):
        return self.base_layer.bias

    def _merge_operands(self):
        W0 = self.base_layer.weight.detach()
        A = self.lora_A[self.adapter].weight.detach()
        B = self.lora_B[self.adapter].weight.detach()
        if self.use_dora:
            m = self.lora_magnitude_vector[self.adapter].weight.detach()
        else:   # plain LoRA: the same kernel with m := 1, the row norm multiplied back in afterwards
            m = torch.ones(self.out_features, device=W0.device)
        return W0.float(), A.float(), B.float(), m.float(), self.scaling

    def _merged(self, merged, nrm):
        if self.use_dora:
            self._last_norm = nrm     # ||W'|| rows: the backward's detached norm
            return merged
        return merged * nrm[:, None]  # merge = W0 + s B A

    def effective_weight(self) -> torch.Tensor:
        """Dense fp32 weight this layer represents right now (HIP merge kernel on the GPU)."""
        from . import ops
        merged, nrm = ops.dora_merge(*self._merge_operands(), return_norm=True)
        return self._merged(merged, nrm)

    @staticmethod
    def effective_weights(mods) -> list:
        """``effective_weight()`` of several wrappers through ONE launch (``gww_dora_merge_batch_f32``): what the encoder
        calls after an optimizer step, which changes every adapted projection at once."""
        from . import ops
        outs = ops.dora_merge_batch([m._merge_operands() for m in mods])
        return [m._merged(w, n) for m, (w, n) in zip(mods, outs)]


def _match(name: str, targets) -> bool:
    return any(name == t or name.endswith("." + t) for t in targets)


class LoraModel(nn.Module):
    def __init__(self, model: nn.Module, config: LoraConfig):
        super().__init__()
        self.model = model
        self.peft_config = {"default": config}
        hit = 0
        for name, mod in list(model.named_modules()):
            if isinstance(mod, nn.Linear) and _match(name, config.target_modules or []):
                parent = model
                *path, leaf = name.split(".")
                for p in path:
                    parent = getattr(parent, p)
                setattr(parent, leaf, DoraLinear(mod, config.r, config.lora_alpha, config.use_dora))
                hit += 1
        if hit == 0:
            raise ValueError(f"Target modules {config.target_modules} not found in the base model.")

    def forward(self, *a, **k):
        return self.model(*a, **k)


class PeftModel(nn.Module):
    def __init__(self, model: nn.Module, peft_config: LoraConfig):
        super().__init__()
        self.base_model = LoraModel(model, peft_config)
        self.peft_config = {"default": peft_config}
        self.active_adapter = "default"

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            if name == "base_model":
                raise
            return getattr(self.base_model.model, name)   # e.g. .config, .last_token, .precision

    def forward(self, *a, **k):
        return self.base_model.model(*a, **k)

    def get_nb_trainable_parameters(self):
        tr = sum(p.numel() for p in self.parameters() if p.requires_grad)
        return tr, sum(p.numel() for p in self.parameters())

    def print_trainable_parameters(self):
        tr, al = self.get_nb_trainable_parameters()
        print(f"trainable params: {tr:,d} || all params: {al:,d} || trainable%: {100 * tr / al:.4f}")

    # ---- on-disk format (peft utils/save_and_load.py)
    def _adapter_state_dict(self) -> dict:
        out = {}
        for k, v in self.state_dict().items():
            if "lora_" not in k:
                continue
            k = k.replace(".default", "")
            if k.endswith("lora_magnitude_vector.weight"):
                k = k[: -len(".weight")]
            out[k] = v.detach().to("cpu", torch.float32).contiguous()
        return out

    def save_pretrained(self, save_directory: str, **kwargs):
        from safetensors.torch import save_file
        os.makedirs(save_directory, exist_ok=True)
        save_file(self._adapter_state_dict(), os.path.join(save_directory, "adapter_model.safetensors"),
                  metadata={"format": "pt"})
        cfg = self.peft_config["default"].to_dict()
        cfg["inference_mode"] = True
        if cfg.get("auto_mapping") is None:
            cfg["auto_mapping"] = {"base_model_class": "WhisperEncoder",
                                   "parent_library": "transformers.models.whisper.modeling_whisper"}
        with open(os.path.join(save_directory, "adapter_config.json"), "w") as f:
            json.dump(cfg, f, indent=2, sort_keys=True)

    @classmethod
    def from_pretrained(cls, model: nn.Module, model_id: str, is_trainable: bool = False, **kwargs) -> "PeftModel":
        from safetensors.torch import load_file
        with open(os.path.join(model_id, "adapter_config.json")) as f:
            raw = json.load(f)
        known = {f for f in LoraConfig.__dataclass_fields__}
        cfg = LoraConfig(**{k: v for k, v in raw.items() if k in known and k != "inference_mode"})
        cfg.inference_mode = not is_trainable
        for p in model.parameters():     # peft freezes the base model when it injects the adapter
            p.requires_grad = False
        peft = cls(model, cfg)
        sd = load_file(os.path.join(model_id, "adapter_model.safetensors"))
        remap = {}
        for k, v in sd.items():
            if k.endswith("lora_magnitude_vector"):
                k2 = k + ".default.weight"
            else:
                k2 = k.replace(".lora_A.weight", ".lora_A.default.weight").replace(".lora_B.weight",
                                                                                    ".lora_B.default.weight")
            remap[k2] = v
        missing, unexpected = peft.load_state_dict(remap, strict=False)
        if unexpected:
            raise KeyError(f"unexpected adapter tensors: {unexpected[:4]}")
        lost = [k for k in missing if "lora_" in k]
        if lost:
            raise KeyError(f"adapter file lacks tensors for: {lost[:4]}")
        dev = next(model.parameters()).device
        peft.to(dev)
        if not is_trainable:
            for n, p in peft.named_parameters():
                if "lora_" in n:
                    p.requires_grad = False
        return peft


def get_peft_model(model: nn.Module, peft_config: LoraConfig, adapter_name: str = "default") -> PeftModel:
    for p in model.parameters():     # peft freezes the base model
        p.requires_grad = False
    peft = PeftModel(model, peft_config)
    for n, p in peft.named_parameters():
        if "lora_" in n:
            p.requires_grad = True
    return peft
