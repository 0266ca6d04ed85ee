"""world_size-2 gloo tests (CPU) of the N > 1 path: segment sharding + gather, and the flat
gradient bucket all-reduce used for DoRA + head gradients."""

import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from gw_whisper_amd import dist as gdist


def test_shard_range_partitions_whole_batches():
    for n, world, batch in [(2_590_000, 8, 256), (64, 2, 32), (1000, 3, 256), (5, 4, 1), (0, 2, 8)]:
        spans = [gdist.shard_range(n, r, world, batch) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        for (a, b), (c, d) in zip(spans, spans[1:]):
            assert b == c and a <= b and c <= d
        for a, b in spans:
            assert a % batch == 0 or a == n            # batch boundaries == the single-GPU run's


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmp):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, w, _ = gdist.init("gloo")
    assert (r, w) == (rank, world)
    # --- inference sharding: each rank "scores" its contiguous span; rank 0 gathers in order
    n, batch = 1003, 64
    a, b = gdist.shard_range(n, rank, world, batch)
    local = torch.arange(a, b, dtype=torch.float32)[:, None] * 2.0
    full = gdist.gather_concat(local, n, rank, world, batch)
    if rank == 0:
        assert torch.equal(full[:, 0], torch.arange(n, dtype=torch.float32) * 2.0)
    # --- training: flat bucket all-reduce == mean of the per-rank gradients
    torch.manual_seed(0)
    lin = torch.nn.Linear(8, 4)
    bucket = gdist.FlatGradBucket(lin.parameters())
    x = torch.full((3, 8), float(rank + 1))
    bucket.zero()
    lin(x).sum().backward()
    local_grad = bucket.flat.clone()
    bucket.all_reduce_mean(world)
    gathered = [torch.empty_like(local_grad) for _ in range(world)]
    torch.distributed.all_gather(gathered, local_grad)
    torch.testing.assert_close(bucket.flat, torch.stack(gathered).mean(0))
    assert lin.weight.grad.data_ptr() == bucket.flat.data_ptr()     # grads ARE the bucket
    torch.save(bucket.flat, os.path.join(tmp, f"g{rank}.pt"))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g0 = torch.load(tmp_path / "g0.pt")
    g1 = torch.load(tmp_path / "g1.pt")
    torch.testing.assert_close(g0, g1)          # replicated parameters see identical gradients
