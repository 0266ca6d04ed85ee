"""Search-pipeline side of the hot path (reference ``MLGWSC-1/inference.py`` / ``train.py``), device resident.

What the reference does per HDF5 segment (``inference.py:173-296, 454-489, 140-166``): a python iterator cuts
overlapping ``[D, 2048]`` windows every 0.1 s on the host, a ``DataLoader`` batches 256 of them, every batch is
copied to the GPU, the network runs, and the scores come back element by element through ``.item()``.  Here the
whitened strain is uploaded ONCE, the windows are strided views of it (10x fewer bytes cross PCIe at a 0.1 s
step), the threshold is applied on the device and triggers leave the GPU in one copy per segment.

    DeviceSegmentSlicer      <->  SegmentSlicer / TorchSegmentSlicer in ``white=True`` mode (:173-296)
    evaluate_slices          <->  evaluate_slices (:454-489), same return value
    get_clusters             <->  get_clusters (:140-166), host side, unchanged arithmetic
    cluster_triggers_device       the same clustering on the device (``gww_cluster_triggers_f64``): scores are thresholded and
                                  clustered where they were computed, one small copy of the clusters leaves the GPU
    GWWhisperClassifier      <->  GWWhisperClassifier (:353-392 / train.py:170-214): adapter -> encoder per detector
                                  -> last token -> MLP (+ Softmax)
    RegBCELoss               <->  train.py:358-370
    ResampleMelAdapter            the reference's OTHER front end (Signal_vs_Noise/utils/preprocess.py:44-51 +
                                  WhisperFeatureExtractor): FFT resampling 2048 -> 16000 Hz and log-mel, on device

Whitening (PyCBC, :56-137): ``whiten.py``.  The HDF5 reader stays host-side reference code.
The Q-transform adapter of the reference (``QTransformAdapter``, both the train.py and the inference.py variant) is
``qscan.QTransformAdapter``; ``GWWhisperClassifier`` takes any adapter mapping ``[B, D, 2048] -> [B, D, 80, 3000]``.
"""

from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import _lib


class RegBCELoss(nn.BCELoss):
    """BCELoss on ``eps + (1 - dim * eps) * p`` (reference ``MLGWSC-1/train.py:358-370``)."""

    def __init__(self, *args, epsilon: float = 1e-6, dim: int = 2, **kwargs):
        super().__init__(*args, **kwargs)
        assert isinstance(dim, int)
        self.regularization_dim = dim
        self.regularization_A = epsilon
        self.regularization_B = 1.0 - epsilon * self.regularization_dim

    def forward(self, inputs: torch.Tensor, target: torch.Tensor, *args, **kwargs) -> torch.Tensor:
        assert inputs.shape[-1] == self.regularization_dim
        return super().forward(self.regularization_A + self.regularization_B * inputs, target, *args, **kwargs)


# --------------------------------------------------------------------------------------- FFT resampling
_RESAMPLE = {}


def resample_matrix(n_in: int = 2048, n_out: int = 16000) -> np.ndarray:
    """``scipy.signal.resample(x, n_out)`` is linear in x: R [n_out, n_in] with resample(x) = R @ x, built by
    resampling the identity (so every convention of scipy -- Nyquist split, scaling -- is inherited).
    Reference: ``Signal_vs_Noise/utils/preprocess.py:44-51`` (``len * 16000 // 2048``)."""
    from scipy.signal import resample
    return resample(np.eye(n_in, dtype=np.float64), n_out, axis=0).astype(np.float32)


def resample(x: torch.Tensor, n_out: int = 16000) -> torch.Tensor:
    """[B, n_in] fp32 on the GPU -> [B, n_out]: one fp32 MFMA GEMM against the cached interpolation matrix."""
    from . import ops
    if not x.is_cuda:
        raise _lib.GwwError("resample needs a GPU tensor: gw_whisper_amd has no CPU path")
    n_in = x.shape[-1]
    key = (n_in, n_out, x.device.index)
    if key not in _RESAMPLE:
        _RESAMPLE[key] = torch.from_numpy(resample_matrix(n_in, n_out)).to(x.device)
    return ops.gemm(x.to(torch.float32).contiguous(), _RESAMPLE[key], None, 0)


class ResampleMelAdapter(nn.Module):
    """``[B, D, 2048]`` whitened strain at 2048 Hz -> ``[B, D, 80, 3000]`` Whisper input features:
    scipy-convention FFT resampling to 16 kHz, then the HIP log-mel front end (no trainable parameters)."""

    def __init__(self, n_detectors: int = 2):
        super().__init__()
        self.n_detectors = n_detectors

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        from . import ops
        B, D, n = x.shape
        wave = resample(x.reshape(B * D, n))
        return ops.logmel(wave).reshape(B, D, 80, 3000)


# --------------------------------------------------------------------------------------- model shell
class GWWhisperClassifier(nn.Module):
    """Reference ``MLGWSC-1/inference.py:353-392``: adapter -> encoder per detector -> token -> MLP (+ Softmax).
    All detectors go through the encoder in ONE stacked batch."""

    def __init__(self, whisper_encoder, n_detectors: int = 2, num_classes: int = 2, adapter: Optional[nn.Module] = None,
                 use_last_token: bool = True):
        super().__init__()
        self.n_detectors = n_detectors
        self.encoder = whisper_encoder
        self.adapter = adapter if adapter is not None else ResampleMelAdapter(n_detectors)
        self.use_last_token = use_last_token
        hidden = self.encoder.config.d_model
        self.classifier = nn.Sequential(
            nn.Linear(hidden * n_detectors, 512), nn.ReLU(), nn.Linear(512, 256), nn.ReLU(), nn.Linear(256, 128),
            nn.ReLU(), nn.Linear(128, 64), nn.ReLU(), nn.Linear(64, num_classes), nn.Softmax(dim=1))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        feats = self.adapter(x)                                   # [B, D, 80, 3000]
        B, D = feats.shape[:2]
        flat = feats.reshape(B * D, *feats.shape[2:])
        if self.use_last_token and not torch.is_grad_enabled():
            tok = self.encoder.last_token(flat)                   # only row 1499 of the final LayerNorm
        else:
            seq = self.encoder(flat).last_hidden_state
            tok = seq[:, -1, :] if self.use_last_token else seq.mean(dim=1)
        combined = tok.reshape(B, D * tok.shape[-1])              # == cat of the per-detector tokens
        return self.classifier(combined)


def remove_softmax_from_classifier(model: GWWhisperClassifier) -> None:
    """USR mode of the reference (``inference.py:395-401``): drop the trailing Softmax."""
    if isinstance(model.classifier[-1], nn.Softmax):
        model.classifier = nn.Sequential(*list(model.classifier.children())[:-1])


# --------------------------------------------------------------------------------------- slicing + triggers
class DeviceSegmentSlicer:
    """Overlapping windows of multi-detector (already whitened) strain, as strided views of ONE device tensor.

    Same indexing and time stamps as the reference ``SegmentSlicer`` with ``white=True`` (``inference.py:173-264``):
    ``delta_t = 1 / (1 / attrs["delta_t"])``, ``index_step_size = int(step_size / delta_t)``, window i covers samples
    ``[i * step, i * step + slice_length)``.  The reference stamps window i with a RUNNING float64 sum
    (``current_time += time_step_size``, ``:262-263``), not with ``start + i * step``: the two differ in the last
    bits and ``get_clusters`` thresholds on time differences, so ``times()`` reproduces the running sum exactly
    (``np.add.accumulate`` is the same sequence of float64 additions), once per segment on the host."""

    def __init__(self, strain, start_time: float = 0.0, delta_t: float = 1.0 / 2048, step_size: float = 0.1,
                 peak_offset: float = 0.6, slice_length: int = 2048, key: str = "segment", device="cuda",
                 white: bool = True, low_frequency_cutoff=None, segment_duration: float = 0.5,
                 max_filter_duration: float = 0.25):
        t = torch.as_tensor(np.asarray(strain) if not torch.is_tensor(strain) else strain)
        if t.dim() != 2:
            raise ValueError("strain must be [detectors, samples]")
        self.key = key
        self.start_time = start_time
        self.delta_t = 1.0 / (1.0 / delta_t)          # the reference's double inversion (:196), kept bit for bit
        if white:
            self.dss = t.to(device=device, dtype=torch.float32).contiguous()   # the ONE host-to-device copy
        else:
            # SegmentSlicer.process (:218-246): whiten every detector on the device (whiten.py) and move the start time
            # by the corrupted edge (the reference adds a flat 0.125 s)
            from .whiten import whiten
            self.dss = whiten(t.to(device), delta_t=self.delta_t, low_frequency_cutoff=low_frequency_cutoff,
                              segment_duration=segment_duration, max_filter_duration=max_filter_duration,
                              device=device).contiguous()
            self.start_time = self.start_time + 0.125
        self.step_size = step_size
        self.peak_offset = peak_offset
        self.slice_length = slice_length
        self.index_step_size = int(self.step_size / self.delta_t)
        self.time_step_size = self.delta_t * self.index_step_size
        self.white = True
        self._time_table = None

    def __len__(self) -> int:
        n = self.dss.shape[1]
        return 0 if n < self.slice_length else 1 + (n - self.slice_length) // self.index_step_size

    def windows(self, i0: int, i1: int) -> torch.Tensor:
        """Windows i0..i1-1 as ``[n, D, slice_length]`` (a view of the strain: no bytes move)."""
        D, _ = self.dss.shape
        n = i1 - i0
        base = self.dss[:, i0 * self.index_step_size:]
        return base.as_strided((n, D, self.slice_length), (self.index_step_size, self.dss.stride(0), 1))

    def times_host(self, i0: int, i1: int) -> np.ndarray:
        """float64 time stamps of windows i0..i1-1, bit-identical to the reference iterator's."""
        if self._time_table is None or len(self._time_table) != len(self):
            steps = np.full((max(len(self), 1),), self.time_step_size, dtype=np.float64)
            steps[0] = self.start_time
            self._time_table = np.add.accumulate(steps)        # t_0 = start, t_i = t_(i-1) + step: sequential adds
        return self._time_table[i0:i1] + self.peak_offset

    def times(self, i0: int, i1: int) -> torch.Tensor:
        return torch.from_numpy(self.times_host(i0, i1)).to(self.dss.device)


def evaluate_slices(slicer: DeviceSegmentSlicer, network: nn.Module, device: str = "cuda",
                    trigger_threshold: float = 0.2, verbose: bool = False,
                    batch_size: int = 256, window_range: Optional[Tuple[int, int]] = None,
                    cluster_threshold: Optional[float] = None):
    """Reference ``evaluate_slices`` (``inference.py:454-489``): run ``network`` over all windows in batches of 256,
    keep ``outputs[:, 0]`` as the signal score, return ``([[time, score], ...] above threshold, [scores per batch])``.
    Scores and times stay on the device; the threshold is one comparison + ``nonzero`` and everything leaves the
    GPU in two copies per segment instead of one ``.item()`` per window.  ``window_range``: evaluate only windows
    ``[w0, w1)`` (a rank's batch-aligned shard).  ``cluster_threshold``: also cluster the triggers of these windows on the
    device (``cluster_triggers_device``) and return ``(triggers, scores per batch, (times, values, variances))``."""
    w0, w1 = (0, len(slicer)) if window_range is None else window_range      # one rank's shard (shard_windows)
    n = w1 - w0
    scores = torch.empty((n,), dtype=torch.float32, device=slicer.dss.device)
    with torch.no_grad():
        for i0 in range(0, n, batch_size):
            i1 = min(n, i0 + batch_size)
            out = network(slicer.windows(w0 + i0, w0 + i1).contiguous())
            scores[i0:i1] = out[:, 0].to(torch.float32)
        keep = torch.nonzero(scores > trigger_threshold).flatten()
        times = slicer.times(w0, w1)
        trig = torch.stack((times[keep], scores[keep].to(torch.float64)), dim=1).cpu().numpy()
        all_scores = scores.cpu().numpy()
        clusters = None if cluster_threshold is None else cluster_triggers_device(times, scores, trigger_threshold,
                                                                                  cluster_threshold)
    new_triggers = [[float(t), float(s)] for t, s in trig]
    all_vals = [all_scores[i0:min(n, i0 + batch_size)] for i0 in range(0, n, batch_size)]
    return (new_triggers, all_vals) if cluster_threshold is None else (new_triggers, all_vals, clusters)


def cluster_triggers_device(times: torch.Tensor, scores: torch.Tensor, trigger_threshold: float = 0.2,
                            cluster_threshold: float = 0.35) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """``get_clusters`` of ONE trigger list (reference ``inference.py:140-166``; the list being the windows of a segment
    whose score exceeds ``trigger_threshold``, ``:484-487``) on the device: ``times`` fp64 [n] (the reference's stamps,
    ``DeviceSegmentSlicer.times``), ``scores`` fp32 [n], both on the GPU.  Returns ``(times, values, time variances)``
    exactly as ``get_clusters`` does -- the per-window scores are never copied to the host for it."""
    import ctypes as C
    from ._lib import check, lib
    if not (times.is_cuda and scores.is_cuda):
        raise _lib.GwwError("cluster_triggers_device needs GPU tensors")
    times = times.to(torch.float64).contiguous()
    scores = scores.to(torch.float32).contiguous()
    n = scores.numel()
    cap = max(1, n)                                       # a cluster holds at least one window
    out_t = torch.empty((cap,), dtype=torch.float64, device=scores.device)
    out_v = torch.empty((cap,), dtype=torch.float32, device=scores.device)
    cnt = torch.zeros((1,), dtype=torch.int32, device=scores.device)
    with torch.cuda.device(scores.device):
        check(lib().gww_cluster_triggers_f64(times.data_ptr(), scores.data_ptr(), n, float(trigger_threshold),
                                             float(cluster_threshold), out_t.data_ptr(), out_v.data_ptr(), cnt.data_ptr(), cap,
                                             torch.cuda.current_stream().cuda_stream), "gww_cluster_triggers_f64")
    k = int(cnt.item())
    t = out_t[:k].cpu().numpy()
    v = out_v[:k].cpu().numpy().astype(np.float64)     # the reference's values are python floats of the fp32 scores
    return t, v, np.full((k,), 0.2)


def get_clusters(triggers: Dict[str, List[List[float]]],
                 cluster_threshold: float = 0.35) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Reference ``get_clusters`` (``inference.py:140-166``): time-ordered triggers closer than
    ``cluster_threshold`` s form a cluster; report the time and value of its maximum, time variance 0.2."""
    times, vals, tvars = [], [], []
    for trig_list in triggers.values():
        cluster: List[List[float]] = []

        def flush():
            if cluster:
                k = int(np.argmax([c[1] for c in cluster]))
                times.append(cluster[k][0]); vals.append(cluster[k][1]); tvars.append(0.2)

        for trig in trig_list:
            if cluster and (trig[0] - cluster[-1][0]) > cluster_threshold:
                flush()
                cluster = []
            cluster.append(trig)
        flush()
    return np.array(times), np.array(vals), np.array(tvars)


def shard_windows(n_windows: int, rank: int, world: int, batch_size: int = 256) -> Tuple[int, int]:
    """Contiguous, batch-aligned range of window indices for one rank (partition the window
    index range on batch boundaries so every rank sees the batches the single-GPU run would)."""
    n_batches = (n_windows + batch_size - 1) // batch_size
    b0 = rank * n_batches // world
    b1 = (rank + 1) * n_batches // world
    return min(n_windows, b0 * batch_size), min(n_windows, b1 * batch_size)
