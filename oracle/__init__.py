"""CPU oracle for the GW-Whisper hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

This package is a plain-numpy restatement of the arithmetic the reference's hot
path dispatches to (HuggingFace ``WhisperFeatureExtractor`` / ``WhisperEncoder``
and peft's DoRA ``Linear``).  It exists only so that
the HIP kernels can be checked against something independent.

Who may import it: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``.  Nothing under ``gw_whisper_amd/`` imports
it, and the product path raises if the HIP library is missing instead of
falling back to this code.

Pinning status:
  * log-mel front end, encoder forward, pooling + heads: pinned by golden
    vectors generated in the build container from the real
    ``transformers==5.15.0`` classes and the reference's own
    ``Signal_vs_Noise/src/model.py`` (script: ``tools/make_golden.py``, data:
    ``tests/golden/*.npz``).
  * DoRA linear: ``peft`` is not installed anywhere we can run, so the formula
    (peft 0.12.0 ``tuners/lora/dora.py``) is pinned only by its algebraic
    invariants and by the on-disk schema of the adapters the reference ships.
    PARITY UNPINNED for DoRA numerics beyond those invariants.
  * Q-transform: ``ml4gw`` is absent and unpinned upstream.  PARITY UNPINNED.
"""

from . import logmel, encoder, dora, heads  # noqa: F401
