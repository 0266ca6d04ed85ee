"""gw_whisper_amd -- MI355X (gfx950) native hot path of GW-Whisper.

1 s strain segment -> log-mel front end -> Whisper encoder (Conv1d stem + MHSA stack,
DoRA-adapted projections) -> last-token pooling -> small MLP head, behind the
HuggingFace ``WhisperFeatureExtractor`` / ``WhisperEncoder`` + peft-DoRA call surface
(SURVEY.md section 8).  All arithmetic runs in hand-written HIP kernels behind the C ABI
of ``include/gww.h`` (``libgww.so``); PyTorch supplies device memory, streams and
``torch.distributed`` only.  There is no CPU or PyTorch fallback.
"""

from ._lib import GwwError, LIB_PATH, lib  # noqa: F401

__all__ = ["GwwError", "LIB_PATH", "lib"]


def __getattr__(name):
    # torch-dependent pieces are imported lazily so `import gw_whisper_amd` stays cheap
    import importlib
    if name in ("ops", "encoder", "synth", "feature_extraction", "peft", "models", "dist", "training", "inference", "qscan"):
        return importlib.import_module(f"{__name__}.{name}")
    if name in ("WhisperEncoder", "WhisperConfig"):
        return getattr(importlib.import_module(f"{__name__}.encoder"), name)
    if name == "WhisperFeatureExtractor":
        return getattr(importlib.import_module(f"{__name__}.feature_extraction"), name)
    raise AttributeError(name)
