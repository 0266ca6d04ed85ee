"""world_size-2 gloo tests (CPU) of the N > 1 path: segment sharding + gather, and the flat
gradient bucket all-reduce used for DoRA + head gradients."""

import os
import socket

import numpy as np
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


def test_epoch_plan_gives_every_rank_the_same_number_of_steps():
    for n, world, batch in [(77, 2, 16), (77, 3, 16), (5, 8, 1), (64, 2, 32), (1, 4, 32), (0, 2, 8)]:
        steps = gdist.epoch_steps(n, world, batch)
        seen = []
        for s in range(steps):
            act = 0
            for r in range(world):
                sl = gdist.step_slice(n, s, r, world, batch)
                if sl is not None:
                    seen += list(range(*sl))
                    act += 1
            assert act == gdist.step_active(n, s, world, batch) >= 1
        assert seen == list(range(n))                 # every item exactly once, in order
        assert gdist.step_slice(n, steps, 0, world, batch) is None


def _epoch_worker(rank, world, port, tmp):
    """The harness's epoch loop (harness/run_train.py) with a toy model: 77 items at batch 16 = 5 batches on 2 ranks --
    3 steps on EVERY rank, the last one with a single active rank."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    gdist.init("gloo")
    torch.manual_seed(0)
    model = torch.nn.Linear(6, 1)
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    bucket = gdist.FlatGradBucket(model.parameters())
    g = torch.Generator().manual_seed(1)
    X, y = torch.randn(77, 6, generator=g), torch.randn(77, 1, generator=g)
    n_collectives = 0
    for epoch in range(2):
        for step in range(gdist.epoch_steps(77, world, 16)):
            sl = gdist.step_slice(77, step, rank, world, 16)
            bucket.zero()
            if sl is not None:
                torch.nn.functional.mse_loss(model(X[sl[0]:sl[1]]), y[sl[0]:sl[1]]).backward()
            bucket.all_reduce_mean(world, active=gdist.step_active(77, step, world, 16))
            n_collectives += 1
            opt.step()
        val = gdist.broadcast_scalar(float(rank) + 0.25, world)        # decisions come from rank 0 on every rank
        assert val == 0.25
    torch.save({"w": model.weight.detach().clone(), "b": model.bias.detach().clone(), "n": n_collectives},
               os.path.join(tmp, f"e{rank}.pt"))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_epoch_loop_with_a_ragged_number_of_batches(tmp_path):
    port = _free_port()
    mp.spawn(_epoch_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    e0, e1 = torch.load(tmp_path / "e0.pt"), torch.load(tmp_path / "e1.pt")
    assert e0["n"] == e1["n"] == 2 * 3
    assert torch.equal(e0["w"], e1["w"]) and torch.equal(e0["b"], e1["b"])      # replicas stay in lock step
    # single process, same schedule by hand: mean over the active batches of every global step
    torch.manual_seed(0)
    model = torch.nn.Linear(6, 1)
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    g = torch.Generator().manual_seed(1)
    X, y = torch.randn(77, 6, generator=g), torch.randn(77, 1, generator=g)
    for epoch in range(2):
        for step in range(3):
            opt.zero_grad()
            sls = [gdist.step_slice(77, step, r, 2, 16) for r in range(2)]
            sls = [s for s in sls if s is not None]
            for lo, hi in sls:
                (torch.nn.functional.mse_loss(model(X[lo:hi]), y[lo:hi]) / len(sls)).backward()
            opt.step()
    torch.testing.assert_close(e0["w"], model.weight.detach())
    torch.testing.assert_close(e0["b"], model.bias.detach())


def _weighted_worker(rank, world, port, tmp):
    """Sample-weighted bucket mean: 77 items at batch 16 on 2 ranks -- the last global step holds ONE batch of 13 items,
    the one before it two full batches; every step must equal the single-process gradient of the mean loss over the
    step's items, whatever the cut over ranks."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    gdist.init("gloo")
    torch.manual_seed(0)
    model = torch.nn.Linear(6, 1)
    bucket = gdist.FlatGradBucket(model.parameters())
    g = torch.Generator().manual_seed(1)
    X, y = torch.randn(77, 6, generator=g), torch.randn(77, 1, generator=g)
    grads = []
    for n_items in (77, 35):          # 35: step 1 = one batch of 3 items on rank 0 next to nothing on rank 1
        for step in range(gdist.epoch_steps(n_items, world, 16)):
            sl = gdist.step_slice(n_items, step, rank, world, 16)
            bucket.zero()
            if sl is not None:
                torch.nn.functional.mse_loss(model(X[sl[0]:sl[1]]), y[sl[0]:sl[1]]).backward()
            bucket.all_reduce_mean(world, n_local=0 if sl is None else sl[1] - sl[0])
            grads.append(bucket.flat.clone())
    torch.save(grads, os.path.join(tmp, f"w{rank}.pt"))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_sample_weighted_mean_equals_the_single_process_gradient(tmp_path):
    port = _free_port()
    mp.spawn(_weighted_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    w0, w1 = torch.load(tmp_path / "w0.pt"), torch.load(tmp_path / "w1.pt")
    torch.manual_seed(0)
    model = torch.nn.Linear(6, 1)
    g = torch.Generator().manual_seed(1)
    X, y = torch.randn(77, 6, generator=g), torch.randn(77, 1, generator=g)
    k = 0
    for n_items in (77, 35):
        for step in range(gdist.epoch_steps(n_items, 2, 16)):
            lo, hi = step * 32, min(n_items, step * 32 + 32)      # the global batch of this step, as ONE batch
            model.zero_grad()
            torch.nn.functional.mse_loss(model(X[lo:hi]), y[lo:hi]).backward()
            ref = torch.cat([model.weight.grad.flatten(), model.bias.grad.flatten()])
            torch.testing.assert_close(w0[k], ref)
            assert torch.equal(w0[k], w1[k])
            k += 1
    assert k == len(w0) == 5


def test_single_process_bucket_needs_no_group():
    model = torch.nn.Linear(3, 2)
    bucket = gdist.FlatGradBucket(model.parameters())
    model(torch.ones(4, 3)).sum().backward()
    ref = bucket.flat.clone()
    bucket.all_reduce_mean(1)
    assert torch.equal(bucket.flat, ref)
    bucket.all_reduce_mean(1, n_local=4)
    torch.testing.assert_close(bucket.flat, ref)


def _search_worker(rank, world, port, tmp):
    """One rank of the sharded search: scores of its batch-aligned window range (a deterministic function of the window
    index stands in for the network -- the data path has no collective), thresholded as evaluate_slices does, gathered by
    harness/run_inference.py's gather_shards over a real 2-process group."""
    import importlib.util
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    from gw_whisper_amd import inference as inf
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("run_inference", os.path.join(root, "harness", "run_inference.py"))
    ri = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ri)
    n_win, batch, thr = 1003, 64, 0.2
    scores = _fake_scores(n_win)
    times = 1.0e9 + 0.1 * np.arange(n_win)
    w0, w1 = inf.shard_windows(n_win, rank, world, batch)
    mine = scores[w0:w1]
    trig = [[float(times[w0 + i]), float(v)] for i, v in enumerate(mine) if v > thr]
    vals = [mine[i:i + batch] for i in range(0, len(mine), batch)]
    trig, vals = ri.gather_shards(trig, vals, world)
    if rank == 0:
        np.save(os.path.join(tmp, "trig.npy"), np.array(trig))
        np.save(os.path.join(tmp, "vals.npy"), np.concatenate(vals))
        t, v, var = inf.get_clusters({"seg": trig}, 0.35)
        np.save(os.path.join(tmp, "clusters.npy"), np.stack([t, v, var]))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def _fake_scores(n):
    rng = np.random.default_rng(42)
    s = rng.random(n).astype(np.float32) * 0.25          # mostly below the threshold ...
    for c in (40, 41, 42, 500, 777, 778, 1000):           # ... with a few bursts, one straddling the shard boundary region
        s[c] = 0.9
    s[510:515] = 0.6
    return s


@pytest.mark.timeout(120)
def test_two_rank_sharded_search_gathers_to_the_single_rank_result(tmp_path):
    """SURVEY.md section 8e, inference: replicas over batch-aligned window shards, results gathered in rank order, clustered
    once -- two gloo ranks give the triggers, the per-window scores and the clusters of the unsharded run, bit for bit."""
    from gw_whisper_amd import inference as inf
    port = _free_port()
    mp.spawn(_search_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    n_win, thr = 1003, 0.2
    scores = _fake_scores(n_win)
    times = 1.0e9 + 0.1 * np.arange(n_win)
    ref_trig = [[float(times[i]), float(v)] for i, v in enumerate(scores) if v > thr]
    np.testing.assert_array_equal(np.load(tmp_path / "trig.npy"), np.array(ref_trig))
    np.testing.assert_array_equal(np.load(tmp_path / "vals.npy"), scores)
    t, v, var = inf.get_clusters({"seg": ref_trig}, 0.35)
    np.testing.assert_array_equal(np.load(tmp_path / "clusters.npy"), np.stack([t, v, var]))
    assert len(t) >= 4

