"""Host-side logic of the search pipeline (gw_whisper_amd/inference.py): against ``golden/inference_host.npz``,
produced by the reference's OWN ``SegmentSlicer`` / ``evaluate_slices`` / ``get_clusters`` (``tools/make_golden.py
inference_host`` runs those definitions of ``MLGWSC-1/inference.py`` in the build container), and against
restatements of the reference for randomised cases; runs on CPU."""
import numpy as np
import pytest
import torch

from gw_whisper_amd import inference as inf


def _reference_get_clusters(triggers, cluster_threshold=0.35):
    # restated from MLGWSC-1/inference.py:140-166
    all_clusters = []
    for trig_list in triggers.values():
        clusters = []
        for trig in trig_list:
            if not clusters or (trig[0] - clusters[-1][-1][0]) > cluster_threshold:
                clusters.append([trig])
            else:
                clusters[-1].append(trig)
        all_clusters.extend(clusters)
    times, vals = [], []
    for cl in all_clusters:
        vs = np.array([x[1] for x in cl])
        k = int(np.argmax(vs))
        times.append(cl[k][0]); vals.append(vs[k])
    return np.array(times), np.array(vals)


def test_get_clusters_matches_reference_algorithm():
    rng = np.random.default_rng(0)
    trig = {}
    for key in ("a", "b", "c"):
        t = np.sort(rng.uniform(0, 50, 200))
        trig[key] = [[float(x), float(v)] for x, v in zip(t, rng.uniform(0.2, 1.0, 200))]
    trig["empty"] = []
    t0, v0 = _reference_get_clusters(trig)
    t1, v1, tv = inf.get_clusters(trig)
    np.testing.assert_array_equal(t0, t1)
    np.testing.assert_array_equal(v0, v1)
    assert (tv == 0.2).all() and len(tv) == len(t1)


def test_slicer_indexing_matches_reference_iteration():
    """SegmentSlicer (white=True): window i = samples [i*step, i*step + 2048), time start + i*dt*step + 0.6."""
    rng = np.random.default_rng(1)
    strain = rng.standard_normal((2, 2048 * 7 + 123)).astype(np.float32)
    sl = inf.DeviceSegmentSlicer(strain, start_time=1000.25, device="cpu")
    assert sl.index_step_size == 204 and abs(sl.time_step_size - 204 / 2048) < 1e-15
    # reference iteration
    ref, idx, t = [], 0, 1000.25
    while idx + 2048 <= strain.shape[1]:
        ref.append((strain[:, idx:idx + 2048], t + 0.6))
        idx += 204
        t += 204 / 2048
    assert len(sl) == len(ref) == 1 + (strain.shape[1] - 2048) // 204
    w = sl.windows(0, len(sl)).numpy()
    ts = sl.times(0, len(sl)).numpy()
    for i in (0, 1, 17, len(ref) - 1):
        np.testing.assert_array_equal(w[i], ref[i][0])
        assert abs(ts[i] - ref[i][1]) < 1e-9
    part = sl.windows(5, 9).numpy()
    np.testing.assert_array_equal(part[2], ref[7][0])
    assert len(inf.DeviceSegmentSlicer(strain[:, :100], device="cpu")) == 0


def test_evaluate_slices_thresholds_like_the_reference_loop():
    rng = np.random.default_rng(2)
    strain = rng.standard_normal((2, 2048 * 40)).astype(np.float32)
    sl = inf.DeviceSegmentSlicer(strain, start_time=5.0, device="cpu")

    class Net(torch.nn.Module):   # a deterministic stand-in: score = sigmoid(mean of detector 0 * 40)
        def forward(self, x):
            s = torch.sigmoid(x[:, 0].mean(dim=1) * 40)
            return torch.stack((s, 1 - s), dim=1)

    trig, vals = inf.evaluate_slices(sl, Net(), device="cpu", trigger_threshold=0.6, batch_size=256)
    n = len(sl)
    assert sum(len(v) for v in vals) == n and len(vals) == (n + 255) // 256
    scores = np.concatenate(vals)
    ref = [[5.0 + i * 204 / 2048 + 0.6, float(scores[i])] for i in range(n) if scores[i] > 0.6]
    assert len(trig) == len(ref) > 0
    np.testing.assert_allclose(np.array(trig), np.array(ref), rtol=0, atol=1e-9)


def test_slicer_time_stamps_are_bit_identical_to_the_reference_iterator(golden):
    """>= 1e5 windows from a GPS-like start: the reference accumulates ``current_time += time_step_size``
    (MLGWSC-1/inference.py:262-263); every time stamp must carry the same bits (sampled values + two checksums over
    ALL windows), not merely agree to 1e-9."""
    g = golden("inference_host.npz")
    n = int(g["long_n_samples"])
    sl = inf.DeviceSegmentSlicer(torch.zeros((2, n)), start_time=g["long_start"][()], delta_t=float(g["long_delta_t_attr"]),
                                 device="cpu")
    assert len(sl) == int(g["long_n_windows"]) >= 100000
    ts = sl.times(0, len(sl)).numpy()
    assert ts.dtype == np.float64
    np.testing.assert_array_equal(ts[g["long_idx"]].view(np.uint64), g["long_times"].view(np.uint64))
    assert np.bitwise_xor.reduce(ts.view(np.uint64)) == g["long_times_xor"]
    assert np.add.reduce(ts.view(np.uint64)) == g["long_times_sum_u64"]
    # a rank's shard sees the same stamps as the whole-segment run
    a, b = inf.shard_windows(len(sl), 3, 8)
    np.testing.assert_array_equal(sl.times(a, b).numpy().view(np.uint64), ts[a:b].view(np.uint64))
    # at a power-of-two sample rate the step (204 / 2048 s) is dyadic and every partial sum is exact: there the
    # closed form start + i * step is the same function
    closed = float(g["long_start"]) + np.arange(len(sl)) * sl.time_step_size + 0.6
    np.testing.assert_array_equal(closed, ts)
    # ... at 4000 Hz it is not, and the running sum is what has to be reproduced
    n = int(g["odd_n_samples"])
    so = inf.DeviceSegmentSlicer(torch.zeros((2, n)), start_time=np.float64(1238166018.3),
                                 delta_t=float(g["odd_delta_t_attr"]), device="cpu")
    assert len(so) == int(g["odd_n_windows"]) and so.index_step_size == int(g["odd_index_step"])
    to = so.times(0, len(so)).numpy()
    np.testing.assert_array_equal(to[g["odd_idx"]].view(np.uint64), g["odd_times"].view(np.uint64))
    assert np.bitwise_xor.reduce(to.view(np.uint64)) == g["odd_times_xor"]
    closed = 1238166018.3 + np.arange(len(so)) * so.time_step_size + 0.6
    assert (closed != to).any()


def test_search_loop_equals_the_reference_loop_on_its_fixture(golden):
    """Windows, scores, triggers ([time, score] pairs) and clusters of a 40 s two-detector segment: the reference's
    TorchSegmentSlicer + DataLoader(256) + evaluate_slices + get_clusters against this build's strided views +
    on-device threshold, same deterministic network, CPU fp32: everything bit-identical."""
    from gw_whisper_amd import synth
    from tests.helpers import search_toy_network
    g = golden("inference_host.npz")
    strain = synth.strain_segments(2, seed=77, n_samples=2048 * 40)
    sl = inf.DeviceSegmentSlicer(strain, start_time=np.float64(1000.25), device="cpu")
    assert len(sl) == int(g["short_n_windows"])
    np.testing.assert_array_equal(sl.windows(7, 8)[0].numpy(), g["short_window_7"])
    trig, vals = inf.evaluate_slices(sl, search_toy_network(), device="cpu", trigger_threshold=0.5)
    np.testing.assert_array_equal(np.concatenate(vals), g["short_scores"])
    assert [len(v) for v in vals] == [256, len(sl) - 256]
    got = np.array(trig, np.float64).reshape(-1, 2)
    assert len(got) == len(g["short_triggers"]) > 50
    np.testing.assert_array_equal(got.view(np.uint64), g["short_triggers"].view(np.uint64))
    t, v, tv = inf.get_clusters({"seg": trig})
    np.testing.assert_array_equal(t, g["cl_short_times"])
    np.testing.assert_array_equal(v, g["cl_short_vals"])
    np.testing.assert_array_equal(tv, g["cl_short_tvars"])
    # sharded evaluation (two ranks) concatenates to the same triggers
    parts = []
    for r in range(2):
        parts += inf.evaluate_slices(sl, search_toy_network(), device="cpu", trigger_threshold=0.5,
                                     window_range=inf.shard_windows(len(sl), r, 2))[0]
    np.testing.assert_array_equal(np.array(parts).view(np.uint64), g["short_triggers"].view(np.uint64))


def test_get_clusters_on_the_reference_fixture(golden):
    """Gaps drawn around the 0.35 s threshold (0.35 and its float neighbour included), two keys + an empty one."""
    g = golden("inference_host.npz")
    t2, v2 = g["cl_in_times"], g["cl_in_vals"]
    trig = {"a": [[float(a), float(b)] for a, b in zip(t2[:250], v2[:250])],
            "b": [[float(a), float(b)] for a, b in zip(t2[250:], v2[250:])], "empty": []}
    t, v, tv = inf.get_clusters(trig, cluster_threshold=0.35)
    np.testing.assert_array_equal(t, g["cl_two_times"])
    np.testing.assert_array_equal(v, g["cl_two_vals"])
    np.testing.assert_array_equal(tv, g["cl_two_tvars"])


def test_shard_windows_is_a_batch_aligned_partition():
    for n in (1, 255, 256, 257, 100000):
        for world in (1, 2, 3, 8):
            cover = []
            for r in range(world):
                a, b = inf.shard_windows(n, r, world)
                assert a % 256 == 0 and a <= b <= n
                cover += list(range(a, b, 256))
            assert cover == list(range(0, n, 256))


def test_reg_bce_loss_matches_formula():
    # MLGWSC-1/train.py:358-370
    rng = np.random.default_rng(3)
    p = torch.softmax(torch.from_numpy(rng.standard_normal((16, 2))), dim=1)
    y = torch.from_numpy(np.eye(2)[rng.integers(0, 2, 16)])
    got = inf.RegBCELoss(dim=2)(p, y).item()
    x = 1e-6 + (1 - 2e-6) * p.numpy()
    ref = -(y.numpy() * np.log(x) + (1 - y.numpy()) * np.log(1 - x)).mean()
    assert abs(got - ref) < 1e-12
    assert torch.isfinite(inf.RegBCELoss(dim=2)(torch.tensor([[1.0, 0.0]], dtype=torch.float64),
                                                torch.tensor([[0.0, 1.0]], dtype=torch.float64)))


def test_resample_matrix_is_scipy_resample():
    # Signal_vs_Noise/utils/preprocess.py:44-51: scipy.signal.resample(x, len * 16000 // 2048)
    from scipy.signal import resample
    R = inf.resample_matrix(2048, 16000).astype(np.float64)
    x = np.random.default_rng(4).standard_normal((3, 2048))
    np.testing.assert_allclose(x @ R.T, resample(x, 16000, axis=1), atol=5e-6)


def test_run_inference_harness_host_side(tmp_path):
    """harness/run_inference.py without a GPU: the reference's argument surface, the segment file layout
    (``/<detector>/<key>`` + start_time / delta_t), the result writer, and the two refusals that need no device."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("run_inference", os.path.join(root, "harness", "run_inference.py"))
    ri = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ri)
    a = ri.parse_args(["in.hdf", "out.hdf", "--white", "--lora-weights", "L", "--dense-weights", "D",
                       "--adapter-weights", "A", "--trigger-threshold=-0.5", "--step-size", "0.1"])
    assert (a.inputfile, a.outputfile, a.white, a.softmax) == ("in.hdf", "out.hdf", True, False)
    assert a.cluster_threshold == 0.35 and a.trigger_threshold == -0.5 and a.device == "cuda"
    rng = np.random.default_rng(0)
    arrays = {}
    for key, n in (("100", 5000), ("200", 3000)):
        for d in ri.DETECTORS:
            arrays[f"{d}/{key}"] = rng.standard_normal(n).astype(np.float32)
        arrays[f"{key}/start_time"], arrays[f"{key}/delta_t"] = float(key), 1.0 / 2048
    src = str(tmp_path / "in.npz")
    np.savez(src, **arrays)
    segs = ri.read_segments(src)
    assert sorted(segs) == ["100", "200"]
    x, st, dt = segs["200"]
    assert x.shape == (2, 3000) and st == 200.0 and dt == 1.0 / 2048
    np.testing.assert_array_equal(x[1], arrays["L1/200"])
    out = str(tmp_path / "res.npz")
    ri.write_result(out, {"time": np.array([1.5]), "stat": np.array([0.9]), "var": np.array([0.2])})
    z = np.load(out)
    assert z["time"][0] == 1.5 and z["var"][0] == 0.2
    with pytest.raises(SystemExit, match="no GPU"):
        ri.main([src, str(tmp_path / "o.npz")])                       # un-whitened input is accepted; no GPU here
    with pytest.raises(RuntimeError):
        ri.main([src, out, "--white"])                                 # output exists, no --force
