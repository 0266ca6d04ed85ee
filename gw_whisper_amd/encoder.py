"""``WhisperEncoder``-compatible module backed by libgww.so.

Mirrors the HuggingFace surface the reference uses:

* built like ``WhisperModel.from_pretrained(id).encoder``
  (reference ``Signal_vs_Noise/src/train.py:227-228``): here
  ``WhisperEncoder(WhisperConfig.named("tiny"))`` + ``load_state_dict`` of an HF
  encoder ``state_dict`` (same key names: ``conv1``, ``conv2``,
  ``embed_positions``, ``layers.N.self_attn.{k,v,q,out}_proj``,
  ``layers.N.self_attn_layer_norm``, ``layers.N.fc1/fc2``,
  ``layers.N.final_layer_norm``, ``layer_norm``);
* ``named_modules()`` yields the names the reference's ``fnmatch`` target search
  consumes (``src/train.py:230-237``);
* ``encoder(mel)`` takes ``[B, 80, 3000]`` fp32 on the GPU and returns an object
  with ``.last_hidden_state [B, 1500, d]`` (``src/model.py:25-26``);
  ``encoder.config.d_model`` exists (``src/model.py:11``);
* ``gradient_checkpointing_enable()`` is accepted (``MLGWSC-1/train.py:662``).

The submodules hold parameters only; all arithmetic runs in the HIP library.
Calling the module with CPU tensors raises -- there is no CPU fallback.
"""

from __future__ import annotations

import ctypes as C
import sys
import weakref
from dataclasses import dataclass

import torch
import torch.nn as nn

from . import _lib, synth
from ._lib import check, lib


@dataclass
class WhisperConfig:
    d_model: int = 384
    encoder_layers: int = 4
    encoder_attention_heads: int = 6
    encoder_ffn_dim: int = 1536
    num_mel_bins: int = 80
    max_source_positions: int = 1500
    dropout: float = 0.0

    @staticmethod
    def named(name: str) -> "WhisperConfig":
        d, L, H, F = synth.ENCODER_SIZES[name]
        return WhisperConfig(d, L, H, F)


@dataclass
class BaseModelOutput:
    last_hidden_state: torch.Tensor

    def __getitem__(self, i):
        return (self.last_hidden_state,)[i]


# Structure epoch: bumped whenever a sub-module or parameter is (re)assigned on one of the encoder's own module classes -- which is
# how adapters get attached (peft's / the shim's ``setattr(parent, leaf, wrapper)``).  The per-step host work of a training
# forward used to walk the module tree five times (``parameters()`` / ``named_parameters()`` are recursive generators: ~0.3 ms
# per step with the GPU idle behind the loop's per-step sync); the parameter lists are now cached per epoch and only the cheap
# per-parameter fields (requires_grad, data_ptr, _version) are read fresh.
_EPOCH = [0]


class _Tracked(nn.Module):
    def __setattr__(self, name, value):
        if isinstance(value, (nn.Module, nn.Parameter)) or name in self.__dict__.get("_modules", ()) \
                or name in self.__dict__.get("_parameters", ()):
            _EPOCH[0] += 1
        super().__setattr__(name, value)

    def __delattr__(self, name):
        _EPOCH[0] += 1
        super().__delattr__(name)

    def add_module(self, name, module):
        _EPOCH[0] += 1
        super().add_module(name, module)

    def register_module(self, name, module):
        _EPOCH[0] += 1
        super().register_module(name, module)

    def register_parameter(self, name, param):
        _EPOCH[0] += 1
        super().register_parameter(name, param)


class _Attention(_Tracked):
    def __init__(self, d):
        super().__init__()
        self.k_proj = nn.Linear(d, d, bias=False)
        self.v_proj = nn.Linear(d, d, bias=True)
        self.q_proj = nn.Linear(d, d, bias=True)
        self.out_proj = nn.Linear(d, d, bias=True)


class _EncoderLayer(_Tracked):
    def __init__(self, d, ffn):
        super().__init__()
        self.self_attn = _Attention(d)
        self.self_attn_layer_norm = nn.LayerNorm(d)
        self.fc1 = nn.Linear(d, ffn)
        self.fc2 = nn.Linear(ffn, d)
        self.final_layer_norm = nn.LayerNorm(d)


# Encoders that have run a training forward.  An optimizer step changes their adapters, and the next forward would start by
# re-preparing the packed weights (DoRA merge, panel packs, LayerNorm folds: host work + a handful of launches) with the GPU
# idle behind the step's loss.item() / synchronize -- the reference loop syncs every step (Signal_vs_Noise/src/train.py:163-168).
# A global optimizer post-step hook does that work right behind optimizer.step(), while the GPU is still busy with the step's
# backward: the next forward finds its change keys equal and returns at once.  Nothing is assumed about the caller's loop: a
# weight changed later (load_state_dict, another optimizer) changes the keys again and is re-synced as before.
_TRAINED = weakref.WeakSet()
_HOOKED = False


def _post_step_sync(*_args, **_kw):
    for enc in list(_TRAINED):
        try:
            if enc._handle is None or not enc._has_trainable_adapters():
                continue
            dev = enc._param_cache()[1][0][1].device
            if dev.type != "cuda":
                continue
            with torch.no_grad(), torch.cuda.device(dev):
                enc._sync_weights()
        except Exception:
            pass   # (a failed early sync is not an error: the next forward syncs and reports)


def _note_training(enc):
    global _HOOKED
    _TRAINED.add(enc)
    if not _HOOKED:
        from torch.optim.optimizer import register_optimizer_step_post_hook
        register_optimizer_step_post_hook(_post_step_sync)
        _HOOKED = True


def _effective_weight(linear) -> torch.Tensor:
    """Dense fp32 weight a (possibly DoRA-wrapped) projection currently represents."""
    if hasattr(linear, "effective_weight"):
        return linear.effective_weight()
    return linear.weight


def _bias(linear):
    base = getattr(linear, "base_layer", linear)
    return base.bias


class WhisperEncoder(_Tracked):
    """Parameter container + launcher for the HIP encoder forward."""

    def __init__(self, config: WhisperConfig, precision: str = "bf16"):
        super().__init__()
        self.config = config
        d = config.d_model
        self.conv1 = nn.Conv1d(config.num_mel_bins, d, kernel_size=3, padding=1)
        self.conv2 = nn.Conv1d(d, d, kernel_size=3, stride=2, padding=1)
        self.embed_positions = nn.Embedding(config.max_source_positions, d)
        self.embed_positions.requires_grad_(False)
        with torch.no_grad():
            self.embed_positions.weight.copy_(torch.from_numpy(synth.sinusoid_table(config.max_source_positions, d)))
        self.layers = nn.ModuleList([_EncoderLayer(d, config.encoder_ffn_dim) for _ in range(config.encoder_layers)])
        self.layer_norm = nn.LayerNorm(d)
        # The base is born frozen: no base-weight backward exists here, so a freshly built
        # encoder runs inference with or without torch.no_grad(); only an explicit un-freeze (the reference's
        # full_finetune, Signal_vs_Noise/src/train.py:244-250) meets the refusal in _wants_grad.
        self._freeze_parameters()
        self.precision = precision
        self.gradient_checkpointing = False
        self._handle = None
        self._packed_key = None
        self._ws = None

    # ---- HF surface the reference touches
    def gradient_checkpointing_enable(self, *a, **k):
        self.gradient_checkpointing = True   # activations are recomputed-by-design in the HIP backward

    def _freeze_parameters(self):
        for p in self.parameters():
            p.requires_grad = False

    @staticmethod
    def from_numpy_state_dict(sd: dict, config: WhisperConfig, **kw) -> "WhisperEncoder":
        enc = WhisperEncoder(config, **kw)
        enc.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        return enc

    # ---- library plumbing
    def __del__(self):
        try:
            h = self.__dict__.get("_handle")
            self.__dict__["_handle"] = None
            # at interpreter shutdown the HIP runtime may already be torn down: the process exit frees the
            # device memory, calling into the library then can block
            if h is not None and not sys.is_finalizing():
                lib().gww_encoder_destroy(h)
        except Exception:
            pass

    def _ensure_handle(self):
        if self._handle is None:
            c = self.config
            cfg = _lib.EncCfg(c.d_model, c.encoder_layers, c.encoder_attention_heads, c.encoder_ffn_dim,
                              c.num_mel_bins, 2 * c.max_source_positions)
            h = C.c_void_p()
            check(lib().gww_encoder_create(C.byref(cfg), C.byref(h)), "gww_encoder_create")
            self._handle = h
        return self._handle

    def _param_cache(self):
        """(all (name, parameter) pairs, parameter lists per weight group) -- rebuilt when the module structure changed."""
        c = self.__dict__.get("_pcache")
        if c is not None and c[0] == _EPOCH[0]:
            return c
        plist = lambda *mods: [p for m in mods for p in m.parameters()]
        groups = [plist(self.conv1, self.conv2, self.embed_positions, self.layer_norm)]
        for L in self.layers:
            a = L.self_attn
            groups.append((plist(a.q_proj, a.k_proj, a.v_proj, L.self_attn_layer_norm), plist(a.out_proj),
                           plist(L.fc1, L.final_layer_norm), plist(L.fc2)))
        c = (_EPOCH[0], list(self.named_parameters()), groups)
        self.__dict__["_pcache"] = c
        return c

    def _group_keys(self):
        """Change keys per weight group -- (data_ptr, version) of every parameter of the group, DoRA wrappers included:
        globals, and per layer the four groups of gww_encoder_update_weights (bit 0 q/k/v + LN1, bit 1 out_proj, bit 2 fc1 + LN2,
        bit 3 fc2)."""
        groups = self._param_cache()[2]
        key = lambda ps: tuple((p.data_ptr(), p._version) for p in ps)
        return key(groups[0]), [tuple(key(ps) for ps in g) for g in groups[1:]]

    def _sync_weights(self):
        """Re-pack into the library's bf16/fp32 panels what changed since the last call (optimizer step,
        load_state_dict, DoRA update): everything the first time, afterwards only the dirty weight groups
        (a DoRA step touches the attention projections: the frozen fc1 / fc2 / stem panels are packed once)."""
        gkey, lkeys = self._group_keys()
        old = self._packed_key
        if old is not None and old == (gkey, lkeys):
            # The panels may have been packed on ANOTHER stream (the optimizer post-step hook runs on whatever stream
            # optimizer.step() ran on): order this stream behind that work once.
            ev = self.__dict__.get("_packed_event")
            if ev is not None and ev[0] != torch.cuda.current_stream().cuda_stream:
                torch.cuda.current_stream().wait_event(ev[1])
            return
        full = old is None
        g_dirty = full or old[0] != gkey
        masks = [15 if full else sum((1 << b) for b in range(4) if old[1][i][b] != lkeys[i][b])
                 for i in range(len(self.layers))]
        f32 = lambda t: t.detach().to(torch.float32).contiguous()
        keep = []   # keep temporaries alive until the async packing kernels are enqueued

        def ptr(t):
            t = f32(t)
            keep.append(t)
            return t.data_ptr()

        # every DoRA-wrapped projection of a dirty group merged by one launch
        wrapped = []
        for L, m in zip(self.layers, masks):
            a = L.self_attn
            for bit, mods in ((1, (a.q_proj, a.k_proj, a.v_proj)), (2, (a.out_proj,)), (4, (L.fc1,)), (8, (L.fc2,))):
                if m & bit:
                    wrapped += [x for x in mods if hasattr(x, "effective_weights")]
        merged = {}
        if wrapped:
            merged = {id(x): w for x, w in zip(wrapped, type(wrapped[0]).effective_weights(wrapped))}
        eff = lambda lin: merged[id(lin)] if id(lin) in merged else _effective_weight(lin)

        g = None
        if g_dirty:
            g = _lib.EncGlobals(ptr(self.conv1.weight), ptr(self.conv1.bias), ptr(self.conv2.weight),
                                ptr(self.conv2.bias), ptr(self.embed_positions.weight), ptr(self.layer_norm.weight),
                                ptr(self.layer_norm.bias))
        n = len(self.layers)
        arr = (_lib.EncLayer * n)()
        for i, L in enumerate(self.layers):
            a, m = L.self_attn, masks[i]
            z = lambda cond, fn: fn() if cond else None       # clean groups: pointers are not read
            arr[i] = _lib.EncLayer(
                z(m & 1, lambda: ptr(L.self_attn_layer_norm.weight)), z(m & 1, lambda: ptr(L.self_attn_layer_norm.bias)),
                z(m & 1, lambda: ptr(eff(a.q_proj))), z(m & 1, lambda: ptr(_bias(a.q_proj))),
                z(m & 1, lambda: ptr(eff(a.k_proj))),
                z(m & 1, lambda: ptr(eff(a.v_proj))), z(m & 1, lambda: ptr(_bias(a.v_proj))),
                z(m & 2, lambda: ptr(eff(a.out_proj))), z(m & 2, lambda: ptr(_bias(a.out_proj))),
                z(m & 4, lambda: ptr(L.final_layer_norm.weight)), z(m & 4, lambda: ptr(L.final_layer_norm.bias)),
                z(m & 4, lambda: ptr(eff(L.fc1))), z(m & 4, lambda: ptr(_bias(L.fc1))),
                z(m & 8, lambda: ptr(eff(L.fc2))), z(m & 8, lambda: ptr(_bias(L.fc2))))
        stream = torch.cuda.current_stream().cuda_stream
        if full:
            check(lib().gww_encoder_set_weights(self._ensure_handle(), C.byref(g), arr, n, stream),
                  "gww_encoder_set_weights")
        else:
            dirty = (C.c_uint * n)(*masks)
            check(lib().gww_encoder_update_weights(self._ensure_handle(), C.byref(g) if g is not None else None, arr, n,
                                                   dirty, stream), "gww_encoder_update_weights")
        self._packed_key = (gkey, lkeys)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.__dict__["_packed_event"] = (stream, ev)

    def _workspace(self, batch: int, prec: int, device) -> torch.Tensor:
        need = lib().gww_encoder_workspace_bytes(self._ensure_handle(), batch, prec)
        if self._ws is None or self._ws.numel() < need or self._ws.device != device:
            self._ws = None
            self._ws = torch.empty((need,), dtype=torch.uint8, device=device)
        return self._ws

    def forward_raw(self, input_features: torch.Tensor, want_hidden: bool = True, want_last: bool = False):
        """Launch the HIP forward; returns (last_hidden_state | None, last_token | None)."""
        x = input_features
        if not x.is_cuda:
            raise _lib.GwwError("WhisperEncoder.forward needs GPU tensors: gw_whisper_amd has no CPU fallback "
                                f"(got input on {x.device})")
        c = self.config
        t_in = 2 * c.max_source_positions
        if x.dim() != 3 or x.shape[1] != c.num_mel_bins or x.shape[-1] != t_in:
            raise ValueError(f"Whisper expects the mel input features to be of length {t_in}, but found "
                             f"{x.shape[-1]}. Make sure to pad the input mel features to {t_in}.")
        if next(self.parameters()).device != x.device:
            raise _lib.GwwError("encoder parameters and input are on different devices")
        x = x.to(torch.float32).contiguous()
        B = x.shape[0]
        prec = {"bf16": _lib.PREC_BF16, "fp32": _lib.PREC_F32}[self.precision]
        with torch.cuda.device(x.device):
            self._sync_weights()
            ws = self._workspace(B, prec, x.device)
            hidden = torch.empty((B, c.max_source_positions, c.d_model), dtype=torch.float32,
                                 device=x.device) if want_hidden else None
            last = torch.empty((B, c.d_model), dtype=torch.float32, device=x.device) if want_last else None
            check(lib().gww_encoder_forward(self._handle, x.data_ptr(), B, prec, ws.data_ptr(), ws.numel(),
                                            hidden.data_ptr() if want_hidden else None,
                                            last.data_ptr() if want_last else None,
                                            torch.cuda.current_stream().cuda_stream), "gww_encoder_forward")
        return hidden, last

    def set_split(self, on: bool = True):
        """Process large batches as two half batches on two streams (see gww_encoder_set_split)."""
        check(lib().gww_encoder_set_split(self._ensure_handle(), int(on)), "gww_encoder_set_split")
        self._ws = None

    # ---- per-kernel event trace (bench.py roofline)
    def trace_enable(self, on: bool = True):
        check(lib().gww_encoder_trace_enable(self._ensure_handle(), int(on)), "gww_encoder_trace_enable")

    def trace_read(self) -> dict:
        """{kernel class: (total ms, launches)} since the last read (synchronises)."""
        n = lib().gww_encoder_trace_classes()
        ms = (C.c_float * n)()
        cnt = (C.c_int * n)()
        check(lib().gww_encoder_trace_read(self._ensure_handle(), ms, cnt), "gww_encoder_trace_read")
        return {lib().gww_encoder_trace_class_name(i).decode(): (float(ms[i]), int(cnt[i])) for i in range(n)}

    def forward(self, input_features, attention_mask=None, **kwargs):
        self._check_input(input_features)
        if self._wants_grad(input_features):
            # DoRA training step: HIP forward that keeps activations + HIP backward (training.py)
            from .training import encoder_train_forward
            return BaseModelOutput(last_hidden_state=encoder_train_forward(self, input_features))
        hidden, _ = self.forward_raw(input_features, want_hidden=True, want_last=False)
        return BaseModelOutput(last_hidden_state=hidden)

    def _wants_grad(self, input_features) -> bool:
        if not torch.is_grad_enabled():
            return False
        # Only the frozen-base + DoRA backward exists (and the input gradient).  A base parameter that was un-frozen on
        # purpose -- the reference's `full_finetune` method, Signal_vs_Noise/src/train.py:244-250 -- would silently get no
        # gradient: refuse.  (Cheap path first: nothing trainable at all is the inference case.)
        named = self._param_cache()[1]
        if not any(p.requires_grad for _, p in named):
            return torch.is_tensor(input_features) and input_features.requires_grad
        base_trainable = [n for n, p in named if p.requires_grad and "lora_" not in n]
        if base_trainable:
            raise _lib.GwwError(
                "WhisperEncoder: autograd is on and base parameters require grad (e.g. " + base_trainable[0] + "): only "
                "the frozen-base + DoRA training step (and the input gradient) is implemented -- freeze the encoder "
                "(get_peft_model does) or run under torch.no_grad()")
        return self._has_trainable_adapters() or (torch.is_tensor(input_features) and input_features.requires_grad)

    def _has_trainable_adapters(self) -> bool:
        for layer in self.layers:
            for name in ("q_proj", "k_proj", "v_proj", "out_proj"):
                mod = getattr(layer.self_attn, name)
                if hasattr(mod, "lora_A") and any(p.requires_grad for p in mod.lora_A.parameters()):
                    return True
        return False

    def _check_input(self, x):
        c = self.config
        t_in = 2 * c.max_source_positions
        if not x.is_cuda:
            raise _lib.GwwError("WhisperEncoder.forward needs GPU tensors: gw_whisper_amd has no CPU fallback "
                                f"(got input on {x.device})")
        if x.dim() != 3 or x.shape[1] != c.num_mel_bins or x.shape[-1] != t_in:
            raise ValueError(f"Whisper expects the mel input features to be of length {t_in}, but found "
                             f"{x.shape[-1]}. Make sure to pad the input mel features to {t_in}.")

    def last_token(self, input_features) -> torch.Tensor:
        """``self(mel).last_hidden_state[:, -1, :]`` without materialising the other
        1499 rows of the final LayerNorm (reference ``src/model.py:25-26``).  Differentiable: with trainable
        adapters (or an input that requires grad) this is the pooled training step of ``training.py``."""
        self._check_input(input_features)
        if self._wants_grad(input_features):
            from .training import encoder_train_forward
            return encoder_train_forward(self, input_features, pooled=True)
        return self.forward_raw(input_features, want_hidden=False, want_last=True)[1]
