"""RCCL on hardware: a world-size-1 ``nccl`` process group on the one leased GPU runs ``dist.init``, the flat gradient
bucket's all-reduce, the score gather and the scalar broadcast through the RCCL library (backend "nccl" IS RCCL on
ROCm).  No scaling is measured here -- that needs a node -- but init + every collective of the N > 1 path has then
executed on a GPU (the N = 2 semantics are covered on CPU by tests/test_dist_gloo.py)."""

import os
import socket

import pytest
import torch

from gw_whisper_amd import dist as gdist

pytestmark = pytest.mark.gpu


def test_world_size_one_rccl_group_runs_every_collective_of_the_training_path():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    saved = {k: os.environ.get(k) for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    try:
        rank, world, local = gdist.init("nccl", force=True)
        assert (rank, world, local) == (0, 1, 0) and torch.distributed.is_initialized()
        assert torch.distributed.get_backend() == "nccl"
        dev = torch.device("cuda", 0)
        torch.manual_seed(0)
        lin = torch.nn.Linear(64, 32).to(dev)
        bucket = gdist.FlatGradBucket(lin.parameters())
        bucket.zero()
        lin(torch.randn(16, 64, device=dev)).square().mean().backward()
        ref = bucket.flat.clone()
        bucket.all_reduce_mean(world)                       # RCCL all-reduce on a CUDA tensor
        torch.cuda.synchronize()
        assert torch.equal(bucket.flat, ref)
        bucket.all_reduce_mean(world, n_local=16)           # the sample-weighted form: gradients + count in one collective
        torch.cuda.synchronize()
        torch.testing.assert_close(bucket.flat, ref)
        scores = torch.arange(1003, dtype=torch.float32, device=dev)[:, None]
        full = gdist.gather_concat(scores, 1003, rank, world, 64)     # RCCL gather
        assert torch.equal(full, scores)
        assert gdist.broadcast_scalar(0.25, world, dev) == 0.25       # RCCL broadcast
        torch.distributed.barrier()
    finally:
        if torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
