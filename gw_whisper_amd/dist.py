"""Data-parallel plumbing: one process per GPU, ``torch.distributed``
(backend "nccl" == RCCL over xGMI on ROCm; "gloo" for the CPU tests).

The reference has no multi-GPU code at all (``Signal_vs_Noise/src/train.py:214`` picks one
device).  Segments / sliding windows are independent, so

* inference (``MLGWSC-1/inference.py:454-489``): a static, contiguous partition of the segment
  (or window) index range -- contiguous so the time-ordered trigger clustering
  (``inference.py:140-166``) can run per rank -- with batch boundaries aligned to the global
  batch grid (``global index // batch``) so results do not depend on the number of ranks;
  NO collective on the data path, one gather of the scores at the end;
* training: identical frozen base on every rank, replicated DoRA + head parameters, ONE flat
  fp32 bucket ``all_reduce(SUM)`` of the trainable gradients per step (<= 11.4 MB: latency
  bound over xGMI), divided by the world size.
"""

from __future__ import annotations

import os
from typing import Iterable, List, Tuple

import torch
import torch.distributed as dist


def init(backend: str | None = None, force: bool = False) -> Tuple[int, int, int]:
    """Initialise from the torchrun environment; returns (rank, world, local_rank).  A single process needs no group
    and gets none unless ``force`` asks for one (a world-size-1 RCCL group still runs every collective through the
    library: the hardware check of this path on a one-GPU box, tests/test_gpu_dist.py)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    return rank, world, local


def _group_up() -> bool:
    return dist.is_available() and dist.is_initialized()


def shard_range(n_items: int, rank: int, world: int, batch: int = 1) -> Tuple[int, int]:
    """Contiguous [start, stop) of this rank, aligned to whole global batches."""
    n_batches = (n_items + batch - 1) // batch
    per, extra = divmod(n_batches, world)
    b0 = rank * per + min(rank, extra)
    b1 = b0 + per + (1 if rank < extra else 0)
    return min(b0 * batch, n_items), min(b1 * batch, n_items)


def epoch_steps(n_items: int, world: int, batch: int) -> int:
    """Optimizer steps of one epoch -- THE SAME NUMBER ON EVERY RANK (each step is one collective): global step s
    takes the ``world`` consecutive batches ``s * world ... s * world + world - 1``, one per rank; when the number of
    batches is not a multiple of ``world`` the last step has fewer active ranks (``step_slice`` returns None for the
    others, which contribute zeros; ``FlatGradBucket.all_reduce_mean`` then divides by the active count)."""
    n_batches = (n_items + batch - 1) // batch
    return (n_batches + world - 1) // world


def step_slice(n_items: int, step: int, rank: int, world: int, batch: int):
    """[lo, hi) of the items this rank trains on in global step ``step``, or None when it has no batch there."""
    lo = (step * world + rank) * batch
    return None if lo >= n_items else (lo, min(n_items, lo + batch))


def step_active(n_items: int, step: int, world: int, batch: int) -> int:
    """Ranks that hold a batch in global step ``step`` (known to every rank without communication)."""
    return sum(step_slice(n_items, step, r, world, batch) is not None for r in range(world))


def broadcast_scalar(value: float, world: int, device=None, src: int = 0) -> float:
    """Rank ``src``'s value on every rank: early stopping / best-checkpoint decisions must not diverge."""
    if world == 1 and not _group_up():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.broadcast(t, src=src)
    return float(t.item())


def gather_concat(local: torch.Tensor, n_items: int, rank: int, world: int, batch: int = 1) -> torch.Tensor | None:
    """Gather the per-rank result rows (in ``shard_range`` order) on rank 0."""
    if world == 1 and not _group_up():
        return local
    sizes = [shard_range(n_items, r, world, batch) for r in range(world)]
    pad = max(b - a for a, b in sizes)
    buf = torch.zeros((pad,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    buf[: local.shape[0]] = local
    out = [torch.empty_like(buf) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, out, dst=0)
    if rank != 0:
        return None
    return torch.cat([o[: b - a] for o, (a, b) in zip(out, sizes)], dim=0)


class FlatGradBucket:
    """One flat fp32 buffer over the trainable parameters' gradients (+ one trailing word for the sample count): a
    single ``all_reduce(SUM)`` per step, then an in-place divide."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev = self.params[0].device
        self.numel = sum(p.numel() for p in self.params)
        self._buf = torch.zeros(self.numel + 1, dtype=torch.float32, device=dev)   # [gradients | sample count]
        self.flat = self._buf[: self.numel]
        off = 0
        for p in self.params:          # gradients become views into the flat buffer
            p.grad = self.flat[off: off + p.numel()].view_as(p)
            off += p.numel()

    def zero(self):
        self._buf.zero_()

    def all_reduce_mean(self, world: int, active: int | None = None, n_local: int | None = None):
        """Sum over ranks and divide.

        ``n_local`` (the samples behind this rank's batch-MEAN gradient, 0 for a rank without a batch) gives the
        sample-weighted mean ``sum_r n_r g_r / sum_r n_r`` -- the gradient of the mean loss over the whole global batch,
        so a 3-sample tail batch on one rank weighs 3 samples, not as much as a full batch elsewhere, and the update does
        not depend on how the batch was cut over ranks.  The count travels in the bucket's trailing word: still ONE
        collective.  Without it: the plain mean over ``active`` (default ``world``) ranks.
        """
        grouped = _group_up()
        if world > 1 and not grouped:
            raise RuntimeError("FlatGradBucket.all_reduce_mean: world > 1 but no process group is initialised "
                               "(call gw_whisper_amd.dist.init first): the gradients would be divided without being summed")
        if n_local is not None:
            if not grouped:   # a single process: n g / n is g -- leave the gradient bit for bit as the backward wrote it
                return
            self.flat.mul_(float(n_local))
            self._buf[self.numel] = float(n_local)
            dist.all_reduce(self._buf, op=dist.ReduceOp.SUM)
            self.flat.div_(self._buf[self.numel].clamp_min(1.0))
            return
        if grouped:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        if world > 1:
            self.flat.div_(world if active is None else max(int(active), 1))
