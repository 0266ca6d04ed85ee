"""``WhisperFeatureExtractor``-compatible callable backed by the HIP log-mel kernel.

Mirrors the call surface the reference uses:

    fe = WhisperFeatureExtractor.from_pretrained(f"openai/whisper-{enc}")     # src/dataset.py:12
    x = fe(audio, sampling_rate=16000, return_tensors="pt").input_features    # src/dataset.py:20-21
    x = fe([a0, a1, ...], sampling_rate=16000, return_tensors="pt")           # Efficiency_test/src/tools.py:125

Same argument meaning and error behaviour as HF
(``HF:models/whisper/feature_extraction_whisper.py:193-346``): ``ValueError`` when
``sampling_rate != 16000``, zero-pad / truncate to 30 s, always returns a batch.

Two entry points of libgww.so do the arithmetic, chosen by ``device`` (constructor or call):

* ``"cpu"`` -- the default, and what HF itself is: ``gww_logmel_host_f32``, plain C++ with no HIP call.  It is
  fork-safe, so the reference's ``dataset.py`` works unchanged inside its forked ``DataLoader`` workers
  (``Signal_vs_Noise/src/train.py:224-225``, ``--num_workers 12``);
* ``"cuda"`` (or a CUDA tensor as input) -- the HIP kernels (``gww_logmel_f32``), the batched variant SURVEY.md
  section 8b calls the additional entry point; ``return_device="cuda"`` keeps the features on the GPU and skips
  the PCIe round trip.

Neither is a fallback of the other and neither is torch / numpy: a missing ``libgww.so`` raises ``GwwError``.
"""

from __future__ import annotations

import numpy as np
import torch

from . import ops


class BatchFeature(dict):
    """Minimal stand-in for ``transformers.BatchFeature``: dict + attribute access."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


class WhisperFeatureExtractor:
    model_input_names = ["input_features"]

    def __init__(self, feature_size=80, sampling_rate=16000, hop_length=160, chunk_length=30, n_fft=400,
                 padding_value=0.0, device="cpu", **kwargs):
        if (feature_size, sampling_rate, hop_length, chunk_length, n_fft) != (80, 16000, 160, 30, 400):
            raise ValueError("gw_whisper_amd implements the Whisper front end only for its published "
                             "configuration (80 mels, 16 kHz, hop 160, 30 s chunks, n_fft 400)")
        self.feature_size = feature_size
        self.sampling_rate = sampling_rate
        self.hop_length = hop_length
        self.chunk_length = chunk_length
        self.n_fft = n_fft
        self.n_samples = chunk_length * sampling_rate
        self.nb_max_frames = self.n_samples // hop_length
        self.padding_value = padding_value
        self.device = device

    @classmethod
    def from_pretrained(cls, name_or_path=None, **kwargs):
        # every openai/whisper-{tiny,base,small,...} preprocessor_config.json is this default
        return cls(**kwargs)

    def __call__(self, raw_speech, truncation=True, pad_to_multiple_of=None, return_tensors=None,
                 return_attention_mask=None, padding="max_length", max_length=None, sampling_rate=None,
                 do_normalize=None, device=None, return_device="cpu", **kwargs):
        if sampling_rate is not None and sampling_rate != self.sampling_rate:
            raise ValueError(
                f"The model corresponding to this feature extractor: {self.__class__.__name__} was trained using a"
                f" sampling rate of {self.sampling_rate}. Please make sure that the provided `raw_speech` input"
                f" was sampled with {self.sampling_rate} and not {sampling_rate}.")
        if do_normalize or return_attention_mask or padding != "max_length" or max_length is not None:
            raise NotImplementedError("only the default padding='max_length' path the reference uses is implemented")
        if device is None:
            device = raw_speech.device if isinstance(raw_speech, torch.Tensor) and raw_speech.is_cuda else self.device
        dev = torch.device(device)
        if isinstance(raw_speech, torch.Tensor):
            wave = raw_speech.to(dev, torch.float32)
            if wave.dim() == 1:
                wave = wave[None]
        else:
            is_batched_numpy = isinstance(raw_speech, np.ndarray) and raw_speech.ndim > 1
            if is_batched_numpy and raw_speech.ndim > 2:
                raise ValueError(f"Only mono-channel audio is supported for input to {self}")
            is_batched = is_batched_numpy or (isinstance(raw_speech, (list, tuple))
                                              and isinstance(raw_speech[0], (np.ndarray, tuple, list)))
            if is_batched:
                rows = [np.asarray(r, dtype=np.float32).reshape(-1) for r in raw_speech]
            else:
                rows = [np.asarray(raw_speech, dtype=np.float32).reshape(-1)]
            n = min(max(len(r) for r in rows), self.n_samples)      # HF truncates at 30 s
            host = np.zeros((len(rows), max(n, 1)), dtype=np.float32)   # ragged rows: zero fill == HF's padding
            for i, r in enumerate(rows):
                m = min(len(r), n)
                host[i, :m] = r[:m]
            wave = torch.from_numpy(host).to(dev)
        feats = ops.logmel(wave) if dev.type == "cuda" else ops.logmel_host(wave)
        if return_device == "cpu":
            feats = feats.cpu()
        if return_tensors == "np":
            feats = feats.cpu().numpy()
        elif return_tensors not in (None, "pt"):
            raise ValueError(f"unsupported return_tensors={return_tensors!r}")
        elif return_tensors is None:
            feats = [f for f in feats.cpu().numpy()]
        elif return_device != "cpu":
            feats = feats.to(return_device)
        return BatchFeature({"input_features": feats})
