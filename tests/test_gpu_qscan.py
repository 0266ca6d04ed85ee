"""Q-transform front end #2 on the GPU against the CPU restatement oracle/qscan.py.  PARITY UNPINNED with respect
to ml4gw (absent, unpinned upstream): these tests pin the HIP kernels to the restatement only."""
import numpy as np
import pytest

from oracle import qscan as oq

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import torch
    assert torch.cuda.is_available()
    return torch


def _signals(n, seed):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, 2048))
    t = np.arange(2048) / 2048.0
    for i in range(0, n, 2):      # chirps of different loudness: different Q planes win
        x[i] += (3.0 + i) * np.sin(2 * np.pi * (60 + 300 * t) * t) * np.exp(-((t - 0.55) / 0.08) ** 2)
    return x


@pytest.mark.parametrize("n,seed,shape", [(1, 0, (128, 128)), (5, 1, (128, 128)), (9, 2, (128, 128)),
                                          (3, 3, (512, 512)), (2, 4, (80, 300))])
def test_qscan_matches_restatement(T, gww, n, seed, shape):
    """128 x 128 is the train.py adapter's resolution (MLGWSC-1/train.py:109), 512 x 512 the inference.py one
    (inference.py:310: up-sampling in both axes for most rows), 80 x 300 an odd, non-square one."""
    from gw_whisper_amd.qscan import QScan
    x = _signals(n, seed)
    ref, best = oq.qscan(x, spectrogram_shape=shape, return_plane=True)
    qs = QScan(duration=1.0, sample_rate=2048, spectrogram_shape=list(shape), qrange=[4, 128])
    out = qs(T.from_numpy(x.astype(np.float32)).cuda())
    assert out.shape == (n, *shape)
    assert int(qs.last_plane.item()) == best          # plane with the largest energy over the WHOLE batch
    got = out.cpu().numpy()
    scale = np.abs(ref).max()
    err = np.abs(got - ref).max()
    print(f"qscan n={n}: plane {best}, max |err| {err:.3e} (max value {scale:.1f})")
    assert err < 2e-3 * max(scale, 1.0)


def test_qscan_plane_choice_is_batch_global(T, gww):
    """The same segment gives a different spectrogram when a loud neighbour moves the batch-wide argmax to another
    plane (SURVEY.md section 8e caveat) -- the kernels reproduce that, it is not a per-sample choice."""
    from gw_whisper_amd.qscan import QScan
    qs = QScan(duration=1.0, sample_rate=2048, spectrogram_shape=[128, 128], qrange=[4, 128])
    rng = np.random.default_rng(7)
    quiet = rng.standard_normal((1, 2048))
    t = np.arange(2048) / 2048.0
    loud = 40.0 * np.sin(2 * np.pi * 700 * t)[None] * np.exp(-((t - 0.5) / 0.3) ** 2) + rng.standard_normal((1, 2048))
    alone = qs(T.from_numpy(quiet.astype(np.float32)).cuda())
    p_alone = int(qs.last_plane.item())
    both = qs(T.from_numpy(np.concatenate([quiet, loud]).astype(np.float32)).cuda())
    p_both = int(qs.last_plane.item())
    ref_alone, b0 = oq.qscan(quiet, return_plane=True)
    ref_both, b1 = oq.qscan(np.concatenate([quiet, loud]), return_plane=True)
    assert (p_alone, p_both) == (b0, b1)
    np.testing.assert_allclose(alone.cpu().numpy(), ref_alone, atol=2e-3 * np.abs(ref_alone).max())
    np.testing.assert_allclose(both.cpu().numpy(), ref_both, atol=2e-3 * np.abs(ref_both).max())


def test_q_adapter_feeds_the_encoder_and_trains_through_it(T, gww):
    """QTransformAdapter (MLGWSC-1/train.py:78-154) -> [B, D, 80, 3000] -> frozen encoder; the adapter's CNN and
    FiLM parameters receive gradients through the encoder's input gradient (train.py:494-504)."""
    from gw_whisper_amd import synth
    from gw_whisper_amd.encoder import WhisperConfig, WhisperEncoder
    from gw_whisper_amd.inference import GWWhisperClassifier
    from gw_whisper_amd.qscan import QTransformAdapter
    T.manual_seed(0)
    sd = synth.encoder_state_dict(128, 2, 2, 512, seed=3)
    enc = WhisperEncoder.from_numpy_state_dict(sd, WhisperConfig(128, 2, 2, 512), precision="bf16").cuda()
    for p in enc.parameters():
        p.requires_grad = False
    model = GWWhisperClassifier(enc, n_detectors=2, num_classes=2, adapter=QTransformAdapter(n_detectors=2)).cuda()
    names = {n for n, _ in model.adapter.named_parameters()}
    assert {"scale", "bias", "film_gamma", "film_beta", "freq_adapter.0.weight", "freq_adapter.8.weight"} <= names
    x = T.from_numpy(_signals(3, 4).astype(np.float32)).cuda()[:, None, :].repeat(1, 2, 1)
    feats = model.adapter(x)
    assert feats.shape == (3, 2, 80, 3000) and T.isfinite(feats).all()
    probs = model(x)
    assert probs.shape == (3, 2) and T.allclose(probs.sum(1), T.ones(3, device="cuda"), atol=1e-5)
    loss = -T.log(probs[:, 0] + 1e-6).mean()
    loss.backward()
    for n, p in model.adapter.named_parameters():
        assert p.grad is not None and T.isfinite(p.grad).all(), n
    assert model.adapter.film_gamma.grad.abs().max() > 0 and model.adapter.freq_adapter[0].weight.grad.abs().max() > 0


@pytest.mark.parametrize("Hin,Win", [(32, 32), (128, 128), (80, 3000), (33, 50)])
def test_adapter_tail_kernel_matches_the_torch_composition(T, gww, Hin, Win):
    """gww_qadapter_tail_f32 (pool -> scale / bias -> FiLM -> stack, one kernel) against the reference's own sequence of
    torch ops (MLGWSC-1/train.py:146-153): forward to 1e-6, and the gradients of y, scale, bias, film_gamma, film_beta.
    32 x 32 is what the train.py CNN hands over (128 x 128 Q-scan, two MaxPool2d(2)), 128 x 128 the inference.py one."""
    from gw_whisper_amd.qscan import _AdapterTail
    T.manual_seed(Hin * 1000 + Win)
    B, D = 3, 2
    ys = [T.randn(B, Hin, Win, device="cuda", requires_grad=True) for _ in range(D)]
    scale = T.tensor([0.7], device="cuda", requires_grad=True)
    bias = T.tensor([-0.2], device="cuda", requires_grad=True)
    gamma = T.tensor([1.3, 0.8], device="cuda", requires_grad=True)
    beta = T.tensor([0.05, -0.1], device="cuda", requires_grad=True)
    out = T.empty((B, D, 80, 3000), device="cuda")
    for i in range(D):
        out = _AdapterTail.apply(ys[i], scale, bias, gamma, beta, out, i)
    ref = T.stack([(scale * T.nn.functional.adaptive_avg_pool2d(ys[i][:, None], (80, 3000))[:, 0] + bias) * gamma[i] + beta[i]
                   for i in range(D)], dim=1)
    assert out.shape == ref.shape == (B, D, 80, 3000)
    assert (out - ref).abs().max().item() < 2e-6
    w = T.randn_like(ref)
    leaves = ys + [scale, bias, gamma, beta]
    g_hip = T.autograd.grad((out * w).sum(), leaves)
    g_ref = T.autograd.grad((ref * w).sum(), leaves)
    for a, b in zip(g_hip, g_ref):
        assert (a - b).abs().max().item() < 2e-3 * (b.abs().max().item() + 1e-6), (a.shape, (a - b).abs().max().item())


@pytest.mark.parametrize("variant,n,hw", [("inference", 3, 512), ("train", 5, 128), ("inference", 2, 128)])
def test_adapter_cnn_kernels_match_the_fp64_torch_cnn(T, gww, variant, n, hw):
    """csrc/qadapter_cnn.hip (conv1 VALU kernel + two implicit-GEMM MFMA kernels, bf16-pair operands) against the SAME
    torch.nn stack run in fp64 on the CPU -- the arithmetic of `self.freq_adapter` (MLGWSC-1/train.py:117-122,
    inference.py:320-330), which the reference computes in fp32.  Bound: 1e-4 of the output scale (round-2 verdict, item 6);
    the map has the dynamic range of a real Q-scan (a few loud tiles over a unit-mean floor)."""
    from gw_whisper_amd.qscan import QTransformAdapter
    T.manual_seed(11 + n)
    ad = (QTransformAdapter.inference_variant() if variant == "inference" else QTransformAdapter.train_variant()).cuda()
    with T.no_grad():
        for p in ad.freq_adapter.parameters():            # away from the initialisation's symmetric tiny biases
            p.mul_(1.5).add_(0.02 * T.randn_like(p))
    g = T.Generator().manual_seed(5)
    q = T.rand(n, hw, hw, generator=g, dtype=T.float64) * 2.0
    q[:, hw // 3: hw // 3 + 7, hw // 2: hw // 2 + 40] += 60.0            # a loud track
    q[0, 0, :] = 25.0                                                       # borders: the zero padding must be the conv's
    q[-1, :, -1] = 30.0
    ref = ad.freq_adapter.double().cpu()(q[:, None])[:, 0]
    ad.freq_adapter.float().cuda()
    with T.no_grad():
        got = ad.cnn_forward(q.float().cuda())
    T.cuda.synchronize()
    assert got.shape == ref.shape == (n, hw // 4, hw // 4)
    err = (got.double().cpu() - ref).abs().max().item()
    scale = ref.abs().max().item()
    print(f"adapter CNN [{variant}, {hw}^2]: max |err| {err:.3e} of scale {scale:.3f} ({err / scale:.2e})")
    assert err < 1e-4 * scale


@pytest.mark.parametrize("variant,n,hw,smooth", [("train", 3, 128, False), ("inference", 2, 256, False), ("train", 2, 256, False),
                                                 ("train", 2, 128, True), ("inference", 2, 256, True), ("train", 1, 256, True)])
def test_adapter_cnn_backward_kernels_match_the_fp64_torch_gradients(T, gww, variant, n, hw, smooth):
    """gww_qadapter_cnn_backward_f32 (conv3 / conv2 recomputed with the ReLU / max-pool routing as their epilogues, the data
    gradients as the forward's implicit GEMM on transposed weights, weight gradients by the fp32 pixel reduction, conv1 on
    the VALU) against torch autograd through the SAME nn.Conv2d / ReLU / MaxPool2d stack in fp64 on the CPU
    (MLGWSC-1/train.py:117-122 trained at :494-504).  ReLU and max-pool make the gradient DISCONTINUOUS in the
    pre-activations: the kernels carry activations as 16-bit pairs (1.5e-5 relative), so of ~10^6 pre-activations a handful
    within 1e-5 of zero (or of their window neighbour) take the other branch than fp64 does, and each moves a gradient by a
    whole term -- the kernels' gradient is exact for THEIR forward (same bits, same masks), and torch's own fp32 run shows
    the same effect (printed as the yardstick).  Two regimes therefore: `smooth` (biases lifted so that no ReLU clamps:
    every index / tiling / border / transposition mistake would show) holds 2e-4 of each gradient's scale (or three times
    the error of torch's own fp32 run where the sum cancels); the generic regime, with a clamping ReLU, 6e-3."""
    from gw_whisper_amd.qscan import QTransformAdapter, _CnnFunction
    T.manual_seed(23 + n)
    ad = (QTransformAdapter.inference_variant() if variant == "inference" else QTransformAdapter.train_variant()).cuda()
    with T.no_grad():
        for p in ad.freq_adapter.parameters():
            p.mul_(1.5).add_(0.02 * T.randn_like(p))
        if smooth:
            for i, lift in ((0, 40.0), (3, 400.0), (6, 4000.0)):   # pre-activations stay positive: ReLU never clamps
                ad.freq_adapter[i].bias.add_(lift)
    g = T.Generator().manual_seed(7)
    q = T.rand(n, hw, hw, generator=g, dtype=T.float64) * 2.0
    q[:, hw // 3: hw // 3 + 7, hw // 2: hw // 2 + 40] += 20.0
    q[0, 0, :] = 9.0
    wgt = T.randn(n, hw // 4, hw // 4, generator=g, dtype=T.float64)
    import copy
    ref_net = copy.deepcopy(ad.freq_adapter).double().cpu()
    (ref_net(q[:, None])[:, 0] * wgt).sum().backward()
    ref = [p.grad for p in ref_net.parameters()]
    params = ad._cnn_params()
    y = _CnnFunction.apply(ad, q.float().cuda(), *params)
    (y * wgt.float().cuda()).sum().backward()
    T.cuda.synchronize()
    # yardstick: the same torch stack in fp32 on the CPU (what the reference's arithmetic is) against its fp64 self -- the
    # ReLU / max-pool routing is discrete, a rounding that flips one decision moves a gradient by a whole term
    net32 = copy.deepcopy(ad.freq_adapter).float().cpu()
    (net32(q.float()[:, None])[:, 0] * wgt.float()).sum().backward()
    names = ["w1", "b1", "w2", "b2", "w3", "b3", "w4", "b4"]
    worst = []
    for name, p, r, p32 in zip(names, params, ref, net32.parameters()):
        assert p.grad is not None and p.grad.shape == r.shape, name
        err = (p.grad.double().cpu() - r).abs().max().item()
        err32 = (p32.grad.double() - r).abs().max().item()
        scale = r.abs().max().item()
        print(f"adapter CNN backward [{variant}, {hw}^2] d{name}: max |err| {err:.3e} of scale {scale:.3e} ({err / max(scale, 1e-30):.2e}); "
              f"torch fp32 CPU against fp64: {err32 / max(scale, 1e-30):.2e}")
        worst.append((name, err, err32, scale))
    for name, err, err32, scale in worst:
        # (dw1 / dw2 of the smooth regime cancel heavily -- sums of ~10^5 terms of both signs -- and torch's own fp32 run is
        # then 5e-4 .. 1e-2 off its fp64 self: the yardstick bounds what fp32 accumulation can deliver there)
        assert err <= max((2e-4 if smooth else 6e-3) * scale, 3.0 * err32) + 1e-12, name


def test_adapter_forward_runs_the_hip_cnn_with_and_without_autograd(T, gww):
    """Inference (no_grad) AND a training step take the HIP CNN (round 4: forward + backward kernels, _CnnFunction); the
    torch.nn modules only hold the parameters.  Same numbers as the torch.nn stack."""
    from gw_whisper_amd.qscan import QTransformAdapter
    T.manual_seed(3)
    ad = QTransformAdapter.train_variant().cuda()
    x = T.from_numpy(_signals(4, 9).astype(np.float32)).cuda().reshape(2, 2, 2048)
    assert ad._use_hip_cnn(x) and ad._cnn_needs_grad()    # parameters require grad, autograd on: _CnnFunction
    y_train = ad(x)
    assert y_train.requires_grad
    with T.no_grad():
        assert ad._use_hip_cnn(x) and not ad._cnn_needs_grad()
        y_hip = ad(x)
        q = T.stack([ad.q_transform(x[:, i]) for i in range(2)], dim=1)                      # [B, D, F, T]
        y_torch = T.stack([(ad.scale * ad.final_pool(ad.freq_adapter(q[:, i:i + 1]))[:, 0] + ad.bias) * ad.film_gamma[i]
                           + ad.film_beta[i] for i in range(2)], dim=1)
    assert T.equal(y_hip, y_train.detach())
    assert y_hip.shape == y_torch.shape == (2, 2, 80, 3000)
    err = (y_hip - y_torch.detach()).abs().max().item()
    scale = y_torch.detach().abs().max().item()
    print(f"adapter forward: HIP CNN vs torch.nn (fp32) {err:.3e} of {scale:.3f}")
    assert err < 2e-4 * scale
    with T.no_grad():                                      # a parameter update re-packs the kernels' weights
        ad.freq_adapter[3].weight.mul_(1.25)
        y2 = ad(x)
    assert (y2 - y_hip).abs().max().item() > 1e-6
