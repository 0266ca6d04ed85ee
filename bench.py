#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): 1 s-strain segments/s through the whisper-tiny
encoder forward at batch 256 x (80 x 3000) log-mel, bf16 MFMA, on N MI355X GPUs.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch: [256, 80, 3000] fp32 log-mel
(already resident in HBM, produced by the HIP front end from seeded 1 s strain) ->
conv stem -> 4 pre-LN layers (LN, QKV, MHSA over all 1500 tokens, out-proj, FFN) ->
final LayerNorm -> last_hidden_state [256, 1500, 384] fp32 AND the pooled last token.
Data-parallel: every rank runs its own batch (weak scaling), no collective on the
inference path.

Rank 0 prints ONE JSON line with the contract fields plus
  roofline     : the dominant kernel (by time) against the gfx950 bf16 dense MFMA peak,
                 its duration measured live with HIP events on the launch stream
  cpu_baseline : oracle/encoder_torch.py -- the torch-CPU restatement of the encoder (a port, not the
                 reference, which cannot travel to the GPU box), pinned by the HF goldens -- timed on
                 this box's host cores on a bounded sample of the same workload
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

MFMA_BF16_PEAK_TFLOPS = 2500.0      # MI355X_MICROARCH.md: ~2.5 PFLOP/s dense bf16
HBM_PEAK_GBS = 8000.0
PMC_TRAFFIC_FILE = "r04_pmc_traffic.json"   # collected by tools/pmc_traffic.py (separate rocprofv3 --pmc passes)
T_TOK, T_IN, DH = 1500, 3000, 64


def launcher_command(argv, n_gpus, port, script=None):
    """The N-rank launch of this script, as the driver itself would issue it (one rank per GPU over RCCL)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), script or os.path.abspath(__file__), *argv]


def free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def self_launch(argv, n_gpus, run=None):
    """``python bench.py --gpus N`` without a torchrun environment: start N fresh rank processes as CHILDREN and
    return their exit code.  Called before anything in this process touches the GPU (never exec / re-exec a process
    that has initialised HIP: forbidden on the GPU boxes)."""
    import subprocess
    run = run or subprocess.run
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return run(launcher_command(argv, n_gpus, free_port()), env=env).returncode


def csrc_hash():
    """Hash of the kernel sources + ABI header: profiles/*pmc_traffic.json is stamped with it, and bench.py drops
    `roofline.traffic` when the stamp no longer matches (a kernel change makes the pasted counters stale)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    root = os.path.dirname(os.path.abspath(__file__))
    for f in sorted(glob.glob(os.path.join(root, "gw_whisper_amd", "csrc", "*.hip")) +
                    glob.glob(os.path.join(root, "gw_whisper_amd", "csrc", "*.h")) + [os.path.join(root, "include", "gww.h")]):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def flops_per_segment(d, L, H, ffn, n_mels=80):
    """Algorithmic FLOPs (2 x MAC) of one encoder forward, BASELINE.md section 2."""
    conv1 = 2 * T_IN * (3 * n_mels) * d
    conv2 = 2 * T_TOK * (3 * d) * d
    proj = 2 * T_TOK * d * d * 4
    attn = 2 * 2 * T_TOK * T_TOK * DH * H
    ffn_f = 2 * 2 * T_TOK * d * ffn
    return {"conv1": conv1, "conv2": conv2, "proj": proj, "attn": attn, "ffn": ffn_f,
            "total": conv1 + conv2 + L * (proj + attn + ffn_f)}


def time_kernel(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters   # ms per launch


def kernel_breakdown(enc_name, B, dev):
    """Per-kernel durations at the bench shapes through the kernel-level C-ABI entry
    points (HIP events on the launch stream)."""
    from gw_whisper_amd import ops, synth
    d, L, H, ffn = synth.ENCODER_SIZES[enc_name]
    M = B * T_TOK
    g = torch.Generator(device="cpu").manual_seed(0)
    rnd = lambda *s: torch.randn(*s, generator=g)
    x32 = rnd(M, d).to(dev)
    h = x32.bfloat16()
    wqkv = (rnd(3 * d, d) / d ** 0.5).to(dev).bfloat16()
    wo = (rnd(d, d) / d ** 0.5).to(dev).bfloat16()
    w1 = (rnd(ffn, d) / d ** 0.5).to(dev).bfloat16()
    w2 = (rnd(d, ffn) / ffn ** 0.5).to(dev).bfloat16()
    bqkv, bo, b1 = rnd(3 * d).to(dev), rnd(d).to(dev), rnd(ffn).to(dev)
    lnw, lnb = torch.ones(d, device=dev), torch.zeros(d, device=dev)
    qkv = (rnd(B, T_TOK, 3 * d) * 0.5).to(dev).bfloat16()
    f1 = rnd(M, ffn).to(dev).bfloat16()
    fl = flops_per_segment(d, L, H, ffn)
    rows = []

    def add(name, fn, flops, bytes_):
        ms = time_kernel(fn)
        rows.append({"kernel": name, "ms": ms, "launches_per_fwd": L, "tflops": flops / ms / 1e9 if flops else None,
                     "gbs": bytes_ / ms / 1e6})

    astat = d in (384, 512)
    if astat:
        # the encoder uses the A-stationary kernels with LayerNorm fused into QKV / fc1
        wq_f, uq, cq = ops.ln_fold_weights(wqkv.float(), lnw, lnb, bqkv)
        w1_f, u1, c1 = ops.ln_fold_weights(w1.float(), lnw, lnb, b1)
        add("ln+gemm_qkv", lambda: ops.gemm_astat(x32, wq_f, None, 0, ln=(uq, cq)),
            B * 2 * T_TOK * d * 3 * d, M * d * 4 + M * 3 * d * 2)
        add("attention", lambda: ops.attention(qkv, H), B * fl["attn"], M * 4 * d * 2)
        add("gemm_out", lambda: ops.gemm_astat(h, wo, bo, 0), B * 2 * T_TOK * d * d, M * d * 4)
        add("resid+ln+gemm_fc1_gelu", lambda: ops.gemm_astat(x32, w1_f, None, 1, ln=(u1, c1), delta=h, return_x=True),
            B * 2 * T_TOK * d * ffn, M * d * 10 + M * ffn * 2)
    else:
        add("layernorm", lambda: ops.layernorm(x32, lnw, lnb, out_bf16=True), 0, M * d * 6)
        rows[-1]["launches_per_fwd"] = 2 * L
        add("gemm_qkv", lambda: ops.gemm(h, wqkv, bqkv, 0), B * 2 * T_TOK * d * 3 * d, M * d * 2 + M * 3 * d * 2)
        add("attention", lambda: ops.attention(qkv, H), B * fl["attn"], M * 4 * d * 2)
        add("gemm_out_resid", lambda: ops.gemm(h, wo, bo, 2, resid=x32), B * 2 * T_TOK * d * d, M * d * 10)
        add("gemm_fc1_gelu", lambda: ops.gemm(h, w1, b1, 1), B * 2 * T_TOK * d * ffn, M * (d + ffn) * 2)
    if astat:
        add("gemm_fc2", lambda: ops.gemm_fulln(f1, w2, bo, 0), B * 2 * T_TOK * d * ffn, M * (ffn * 2 + d * 2))
    if d == 384:
        wt = ops.mlp_pack(w1_f, w2)
        add("mlp_fused(ln+fc1+gelu+fc2)", lambda: ops.mlp_fused(x32, h, wt, u1, c1, bo),
            B * 4 * T_TOK * d * ffn, M * d * 12)
    else:
        add("gemm_fc2_resid", lambda: ops.gemm(f1, w2, bo, 2, resid=x32), B * 2 * T_TOK * d * ffn, M * (ffn * 2 + d * 8))
    return rows


def dora_step(enc_name, per_gpu_batch, dev, world, steps=6, warmup=2):
    """DoRA fine-tuning step as in Signal_vs_Noise/src/train.py:163-168,263-277: whisper encoder with
    DoRA (r=8, alpha=32) on q/k/v, two-detector MLP head, BCEWithLogits, AdamW over the 'lora' + head
    parameters; forward + backward in libgww, head / loss / optimizer in torch; gradients of the
    trainable parameters all-reduced over RCCL in ONE flat bucket when world > 1."""
    import fnmatch
    from gw_whisper_amd import dist as gdist, ops, synth
    from gw_whisper_amd.encoder import WhisperConfig, WhisperEncoder
    from gw_whisper_amd.models import two_channel_ligo_binary_classifier
    from gw_whisper_amd.peft import LoraConfig, get_peft_model
    torch.manual_seed(0)
    sd = synth.named_encoder_state_dict(enc_name, seed=0)
    enc = WhisperEncoder.from_numpy_state_dict(sd, WhisperConfig.named(enc_name), precision="bf16")
    pats = ["layers.*.self_attn.q_proj", "layers.*.self_attn.k_proj", "layers.*.self_attn.v_proj",
            "layers.*.self_attn.o_proj"]
    targets = [n for n, _ in enc.named_modules() if any(fnmatch.fnmatch(n, p) for p in pats)]
    peft = get_peft_model(enc, LoraConfig(use_dora=True, r=8, lora_alpha=32, target_modules=targets))
    for name, p in peft.named_parameters():
        p.requires_grad = "lora" in name
    model = two_channel_ligo_binary_classifier(peft).to(dev)
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(params, lr=1e-4)
    bucket = gdist.FlatGradBucket(params)
    crit = torch.nn.BCEWithLogitsLoss()
    rank = int(os.environ.get("RANK", "0"))
    h1 = ops.logmel(torch.from_numpy(synth.strain_segments(per_gpu_batch, seed=7 + rank)).to(dev))
    l1 = ops.logmel(torch.from_numpy(synth.strain_segments(per_gpu_batch, seed=77 + rank)).to(dev))
    labels = (torch.arange(per_gpu_batch, device=dev) % 2).float()[:, None]
    times = {"fwd": 0.0, "bwd": 0.0, "allreduce": 0.0, "opt": 0.0}
    ev = lambda: torch.cuda.Event(enable_timing=True)
    total = 0.0
    for it in range(warmup + steps):
        e = [ev() for _ in range(5)]
        bucket.zero()
        e[0].record()
        loss = crit(model(h1, l1), labels)
        e[1].record()
        loss.backward()
        e[2].record()
        bucket.all_reduce_mean(world)
        e[3].record()
        opt.step()
        e[4].record()
        torch.cuda.synchronize()
        if os.environ.get("GWW_BENCH_VERBOSE"):
            print(f"[dora_step] it {it}: " + " ".join(f"{k}={e[i].elapsed_time(e[i + 1]):.2f}" for k, i in
                  (("fwd", 0), ("bwd", 1), ("allreduce", 2), ("opt", 3))), file=sys.stderr, flush=True)
        if it >= warmup:
            for k, i in (("fwd", 0), ("bwd", 1), ("allreduce", 2), ("opt", 3)):
                times[k] += e[i].elapsed_time(e[i + 1]) / steps
            total += e[0].elapsed_time(e[4]) / steps
    assert torch.isfinite(loss).all()
    import torch.distributed as tdist
    return {"ms": total, "split_ms": times, "per_gpu_batch": per_gpu_batch, "detectors": 2,
            "allreduce": f"RCCL all_reduce(SUM) of ONE flat fp32 bucket, {(bucket.numel + 1) * 4 / 1e6:.1f} MB, world size {world}"
                         if tdist.is_initialized() else "no process group: skipped",
            "trainable_params": int(bucket.numel), "adapter": f"DoRA r=8 alpha=32 on q,k,v ({len(targets)} modules on {enc_name})",
            "loss": float(loss.detach())}


def config4_composed(dev, mel64, rank):
    """BASELINE configs[3] end to end on one GPU: whisper-base + the 22-class head of Glitch_classification/src/model.py
    behind BOTH front ends -- the log-mel extractor the reference's Glitch code calls (dataset.py:46) and the Q-transform
    adapter BASELINE names (QTransformAdapter defaults, one detector, MLGWSC-1/train.py:78-154) -- strain in, argmax out."""
    from gw_whisper_amd import ops, synth
    from gw_whisper_amd.encoder import WhisperConfig, WhisperEncoder
    from gw_whisper_amd.models import glitch_classifier
    from gw_whisper_amd.qscan import QTransformAdapter
    B = 64
    enc = WhisperEncoder.from_numpy_state_dict(synth.named_encoder_state_dict("base", seed=0), WhisperConfig.named("base"),
                                               precision="bf16")
    model = glitch_classifier(enc, num_classes=22).to(dev).eval()
    wave16k = torch.from_numpy(synth.strain_segments(B, seed=4000 + rank)).to(dev)
    wave2k = torch.from_numpy(synth.strain_segments(B, seed=4100 + rank, n_samples=2048)).to(dev)[:, None, :]
    adapter = QTransformAdapter.train_variant(n_detectors=1).to(dev).eval()
    out = {}
    with torch.no_grad():
        ms_mel = time_kernel(lambda: model(ops.logmel(wave16k)).argmax(1), iters=5, warm=2)
        ms_q = time_kernel(lambda: model(adapter(wave2k)[:, 0]).argmax(1), iters=5, warm=2)
        labels = model(adapter(wave2k)[:, 0]).argmax(1)
    assert labels.shape == (B,)
    for k, ms in (("logmel_frontend", ms_mel), ("q_transform_frontend", ms_q)):
        out[k] = {"batch": B, "ms_per_batch": ms, "segments_per_s_per_gpu": B / ms * 1e3}
    out["workload"] = ("BASELINE configs[3]: strain -> front end -> whisper-base last token -> 22-class head -> argmax, "
                       "B = 64, one GPU; Q front end = QTransformAdapter(128 x 128 Q-scan, CNN 32/64/128) -- parity unpinned "
                       "(DESIGN.md section 2)")
    return out


def config5_composed(dev, rank, enc_names=("tiny", "small"), n_batches=4):
    """BASELINE configs[4] on one GPU: the search loop of MLGWSC-1/inference.py:454-489 -- DeviceSegmentSlicer over a
    seeded two-detector strain segment (0.1 s step), 256-window batches through QTransformAdapter(inference variant) ->
    encoder -> MLP + Softmax head, threshold + clustering on the device -- windows/s per GPU and the projected wall time
    of one month (2.59e7 windows) on 1 and 8 GPUs (replicas over batch-aligned window shards, no collective)."""
    from gw_whisper_amd import inference as inf, synth
    from gw_whisper_amd.encoder import WhisperConfig, WhisperEncoder
    from gw_whisper_amd.qscan import QTransformAdapter
    n_win = 256 * n_batches
    strain = synth.strain_segments(2, seed=5000 + rank, n_samples=2048 + (n_win - 1) * 204)
    month = 30 * 86400 / 0.1
    out = {}
    for name in enc_names:
        d = synth.ENCODER_SIZES[name][0]
        enc = WhisperEncoder.from_numpy_state_dict(synth.named_encoder_state_dict(name, seed=0), WhisperConfig.named(name),
                                                   precision="bf16")
        model = inf.GWWhisperClassifier(enc, n_detectors=2, num_classes=2,
                                        adapter=QTransformAdapter.inference_variant(n_detectors=2)).to(dev).eval()
        sl = inf.DeviceSegmentSlicer(strain, start_time=0.0)
        assert len(sl) == n_win

        def run():
            return inf.evaluate_slices(sl, model, trigger_threshold=0.2, batch_size=256, cluster_threshold=0.35)
        run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        trig, vals, clusters = run()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        wps = n_win / dt
        out[f"whisper_{name}"] = {"windows": n_win, "ms_per_256_window_batch": dt / n_batches * 1e3, "windows_per_s_per_gpu": wps,
                                  "projected_month_hours_1gpu": month / wps / 3600, "projected_month_hours_8gpu": month / wps / 8 / 3600,
                                  "triggers": len(trig), "clusters": int(len(clusters[0]))}
        del model, enc, sl
        torch.cuda.empty_cache()
    out["workload"] = ("BASELINE configs[4]: sliding-window search, 2 detectors, QTransformAdapter(512 x 512 Q-scan, CNN 16/32/64) "
                       "-> encoder -> head -> device threshold + clustering, whitening skipped (--white), host wall time incl. "
                       "the two result copies per segment; reference hard-codes tiny (inference.py:408), BASELINE asks small: both; "
                       "Q-scan parity unpinned (DESIGN.md section 2)")
    return out


def cpu_baseline(enc_name, budget_s=20.0):
    """The encoder forward on the host cores: the torch-CPU restatement of the HF arithmetic (oracle/encoder_torch.py, a
    port pinned by the HF goldens, not the reference itself -- the reference tree does not travel to the GPU box), fp32,
    all the threads torch uses here (stated), on a bounded sample of the same workload sized to ~budget_s seconds."""
    import torch as T
    from gw_whisper_amd import synth
    from oracle import encoder as oenc
    from oracle import encoder_torch as oet
    from oracle import logmel as olm
    sd = synth.named_encoder_state_dict(enc_name, seed=0)
    cfg = oenc.EncCfg.named(enc_name)
    threads = T.get_num_threads()
    chunk = 4
    mel = olm.log_mel(synth.strain_segments(chunk, seed=0))
    oet.encoder_forward(sd, mel, cfg, chunk=chunk)                # warm the thread pool / oneDNN primitives
    t0 = time.perf_counter()
    oet.encoder_forward(sd, mel, cfg, chunk=chunk)
    probe = (time.perf_counter() - t0) / chunk
    n_seg = int(min(256, max(chunk, round(budget_s / probe / chunk) * chunk)))
    mel = olm.log_mel(synth.strain_segments(n_seg, seed=0))
    t0 = time.perf_counter()
    oet.encoder_forward(sd, mel, cfg, chunk=chunk)
    dt = time.perf_counter() - t0
    return {"value": n_seg / dt, "unit": "segments/s", "cores": int(threads), "kind": "port",
            "sample": f"{n_seg} segments x [80,3000] log-mel, whisper-{enc_name} encoder fwd, torch-CPU fp32 restatement "
                      f"(oracle/encoder_torch.py: oneDNN conv, threaded GEMMs, CPU SDPA), {threads} threads of "
                      f"{os.cpu_count()} host CPUs, {dt:.1f} s wall"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--encoder", default="tiny")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-breakdown", action="store_true", help="skip the per-kernel event trace")
    ap.add_argument("--split", type=int, default=0, help="0 (default): one stream; 1: two half batches on two HIP streams "
                    "(paid off before the fused MLP kernel, which owns a whole CU; now within noise)")
    ap.add_argument("--no-train", action="store_true", help="skip the DoRA step timing")
    ap.add_argument("--no-pooled", action="store_true", help="skip the pooled (last-token-only) forward timing")
    ap.add_argument("--train-batch", type=int, default=32, help="per-GPU batch of the DoRA step (reference default 32)")
    ap.add_argument("--isolated", action="store_true", help="also time each kernel class in isolation")
    ap.add_argument("--no-extra", action="store_true", help="skip the whisper-base forward / whisper-small DoRA step extras")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: become the launcher of N child ranks (nothing here has touched the GPU yet)
        sys.exit(self_launch(sys.argv[1:], args.gpus))
    args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist

    class _stdout_to_stderr:
        """RCCL prints its version banner on the process's stdout (fd 1) when a communicator comes up; the driver parses
        stdout as ONE JSON line, so fd 1 points at stderr while the group is created."""
        def __enter__(self):
            sys.stdout.flush()
            self.saved = os.dup(1)
            os.dup2(2, 1)
        def __exit__(self, *exc):
            sys.stdout.flush()
            os.dup2(self.saved, 1)
            os.close(self.saved)

    def _bring_up(**kw):
        with _stdout_to_stderr():
            dist.init_process_group("nccl", device_id=dev, **kw)
            t = torch.zeros(1, device=dev)
            dist.all_reduce(t)          # the communicator itself is created by the first collective
            torch.cuda.synchronize()
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        _bring_up()
    elif not args.no_train and args.precision == "bf16":
        # N = 1: a world-size-1 RCCL group, so that the DoRA step below goes through dist.init's path and
        # FlatGradBucket.all_reduce_mean issues a real ncclAllReduce on the 6.1 MB (tiny) / 10.8 MB (small) bucket --
        # `dora_step.split_ms.allreduce` is then a measured collective, not a skipped branch
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        _bring_up(rank=0, world_size=1)

    from gw_whisper_amd import ops, synth
    from gw_whisper_amd.encoder import WhisperConfig, WhisperEncoder

    d, L, H, ffn = synth.ENCODER_SIZES[args.encoder]
    B = args.batch
    sd = synth.named_encoder_state_dict(args.encoder, seed=0)
    enc = WhisperEncoder.from_numpy_state_dict(sd, WhisperConfig.named(args.encoder), precision=args.precision).to(dev)
    if args.split:
        enc.set_split(True)
    # synthetic whitened-like 1 s strain -> HIP log-mel front end; features stay resident in HBM
    wave = torch.from_numpy(synth.strain_segments(B, seed=1000 + rank)).to(dev)
    mel = ops.logmel(wave)
    fe_call_ms = time_kernel(lambda: ops.logmel(wave), iters=10, warm=2)     # what a caller of ops.logmel pays (host incl.)
    # ... and the device time of the same call (memset + k_logmel_frames_mfma + k_logmel_finalize) replayed from a
    # hipGraph, so that the host-side cost of the Python call does not pace the GPU
    fe_ms = None
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            ops.logmel(wave)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            fe_out = ops.logmel(wave)
        fe_ms = time_kernel(graph.replay, iters=20, warm=2)
        assert torch.equal(fe_out, mel)
        del graph, fe_out
    except Exception as exc:   # capture unavailable: report the call time only
        print(f"[bench] front-end graph capture failed: {exc}", file=sys.stderr)
    # front end #2 (BASELINE configs 4 / 5): Q-scan of B two-detector 1 s windows at 2048 Hz -> [2 B, 128, 128]
    from gw_whisper_amd.qscan import QScan
    qs = QScan(duration=1.0, sample_rate=2048, spectrogram_shape=[128, 128], qrange=[4, 128])
    strain2k = torch.from_numpy(synth.strain_segments(2 * B, seed=2000 + rank, n_samples=2048)).to(dev)
    q_ms = time_kernel(lambda: qs(strain2k), iters=5, warm=1)
    # the whole Q-transform adapter of BASELINE configs 4 / 5 (Q-scan -> the small CNN as three HIP launches, csrc/qadapter_cnn.hip
    # -> pool + affine + FiLM + stack as one HIP kernel), both variants of the reference, B two-detector windows; and that
    # tail kernel alone
    from gw_whisper_amd.qscan import QTransformAdapter, _AdapterTail
    adapter_ms = {}
    with torch.no_grad():
        xw = strain2k.reshape(B, 2, 2048)
        for vname, make in (("train_py_variant(128x128,32/64/128)", QTransformAdapter.train_variant),
                            ("inference_py_variant(512x512,16/32/64)", QTransformAdapter.inference_variant)):
            ad = make(n_detectors=2).to(dev).eval()
            adapter_ms[vname] = time_kernel(lambda: ad(xw), iters=3, warm=1)
            del ad
        ycnn = torch.randn(B, 32, 32, device=dev)
        one, zero = torch.ones(1, device=dev), torch.zeros(1, device=dev)
        gam, bet = torch.ones(2, device=dev), torch.zeros(2, device=dev)
        feat = torch.empty(B, 2, 80, 3000, device=dev)
        tail_ms = time_kernel(lambda: _AdapterTail.apply(ycnn, one, zero, gam, bet, feat, 0), iters=10, warm=2)
        del feat, ycnn, xw
    del strain2k
    torch.cuda.empty_cache()

    def step():
        return enc.forward_raw(mel, want_hidden=True, want_last=True)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    trace = not args.no_breakdown
    with torch.no_grad():
        for _ in range(args.warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step()
        barrier()
        dt = time.perf_counter() - t0
        traced = {}
        if trace:
            # Per-kernel durations for the roofline: a SEPARATE pass of the same K steps on ONE stream
            # (hipEvents on the launch stream around every kernel).  In the timed region above every launch
            # of a two-stream run shares the CUs with the other half batch's kernel, so its duration is not
            # its own; here each launch covers the whole batch and owns the GPU.
            if args.split:
                enc.set_split(False)
            step()
            enc.trace_enable(True)
            for _ in range(args.steps):
                out = step()
            traced = enc.trace_read()
            enc.trace_enable(False)
            if args.split:
                enc.set_split(True)
    assert torch.isfinite(out[1]).all()
    # extra (never `value`): the classifiers of the reference only read token 1499 (src/model.py:25-26); with
    # want_hidden=False the last layer runs on the B pooled rows above its attention.  Reported beside the full
    # forward, which stays the headline metric.
    pooled_ms = None
    if not args.no_pooled:
        with torch.no_grad():
            pooled_ms = time_kernel(lambda: enc.forward_raw(mel, want_hidden=False, want_last=True), iters=5, warm=2)
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    ms_per_step = dt / args.steps * 1e3
    value = world * B * args.steps / dt

    train = None
    if not args.no_train and args.precision == "bf16":
        del out
        enc._ws = None
        torch.cuda.empty_cache()
        train = dora_step(args.encoder, args.train_batch, dev, world)

    # extras (never `value`): BASELINE configs 3 and 4 shapes in the same driver-run record -- whisper-small DoRA step
    # (q, k, v targets, per-GPU batch 32 x 2 detectors, gradients all-reduced over RCCL when world > 1) and the
    # whisper-base forward at B = 64
    extra = None
    if not args.no_extra and args.precision == "bf16" and args.encoder == "tiny":
        torch.cuda.empty_cache()
        extra = {}
        db, Lb, Hb, fb = synth.ENCODER_SIZES["base"]
        enc_b = WhisperEncoder.from_numpy_state_dict(synth.named_encoder_state_dict("base", seed=0),
                                                     WhisperConfig.named("base"), precision="bf16").to(dev)
        with torch.no_grad():
            ms_b = time_kernel(lambda: enc_b.forward_raw(mel[:64], want_hidden=True, want_last=True), iters=5, warm=2)
            ms_bp = time_kernel(lambda: enc_b.forward_raw(mel[:64], want_hidden=False, want_last=True), iters=5, warm=2)
        tf_b = 64 * flops_per_segment(db, Lb, Hb, fb)["total"] / (ms_b * 1e-3) / 1e12
        extra["whisper_base_forward"] = {"batch": 64, "ms_per_batch": ms_b, "segments_per_s_per_gpu": 64 / ms_b * 1e3,
                                         "achieved_tflops_per_gpu": tf_b, "frac_of_bf16_mfma_peak": tf_b / MFMA_BF16_PEAK_TFLOPS,
                                         "pooled_ms_per_batch": ms_bp, "workload": "BASELINE configs[3] encoder (22-class head is torch.nn)"}
        del enc_b
        torch.cuda.empty_cache()
        ds_, Ls_, Hs_, fs_ = synth.ENCODER_SIZES["small"]
        enc_s = WhisperEncoder.from_numpy_state_dict(synth.named_encoder_state_dict("small", seed=0),
                                                     WhisperConfig.named("small"), precision="bf16").to(dev)
        with torch.no_grad():
            ms_s = time_kernel(lambda: enc_s.forward_raw(mel[:64], want_hidden=True, want_last=True), iters=3, warm=1)
        tf_s = 64 * flops_per_segment(ds_, Ls_, Hs_, fs_)["total"] / (ms_s * 1e-3) / 1e12
        extra["whisper_small_forward"] = {"batch": 64, "ms_per_batch": ms_s, "segments_per_s_per_gpu": 64 / ms_s * 1e3,
                                          "achieved_tflops_per_gpu": tf_s, "frac_of_bf16_mfma_peak": tf_s / MFMA_BF16_PEAK_TFLOPS,
                                          "workload": "BASELINE configs[2] / [4] encoder (per-op path: LayerNorm kernel + "
                                                      "k_gemm_bf16_v4 + k_attention_dma_bf16)"}
        del enc_s
        torch.cuda.empty_cache()
        # the parity gate itself: GWW_PREC_F32 (exact fp32 MFMA, 1/16 of the bf16 rate) -- the mode whose logits match
        # the reference to 1e-7; the bf16 headline path is held to 1e-3 on logits / exact labels
        enc_f = WhisperEncoder.from_numpy_state_dict(synth.named_encoder_state_dict("tiny", seed=0),
                                                     WhisperConfig.named("tiny"), precision="fp32").to(dev)
        with torch.no_grad():
            ms_f = time_kernel(lambda: enc_f.forward_raw(mel[:64], want_hidden=True, want_last=True), iters=3, warm=1)
        tf_f = 64 * flops_per_segment(d, L, H, ffn)["total"] / (ms_f * 1e-3) / 1e12
        extra["whisper_tiny_forward_fp32_parity_mode"] = {
            "batch": 64, "ms_per_batch": ms_f, "segments_per_s_per_gpu": 64 / ms_f * 1e3, "achieved_tflops_per_gpu": tf_f,
            "frac_of_fp32_mfma_peak": tf_f / 157.3, "what": "precision='fp32': v_mfma_f32_16x16x4_f32 / 32x32x2 everywhere, "
            "the mode pinned to the reference's logits at 1e-7 (tests/test_gpu_encoder.py)"}
        del enc_f
        torch.cuda.empty_cache()
        extra["whisper_small_dora_step"] = dora_step("small", args.train_batch, dev, world, steps=4, warmup=2)
        extra["whisper_small_dora_step"]["workload"] = "BASELINE configs[2]: whisper-small + DoRA fine-tune, data-parallel"
        torch.cuda.empty_cache()
        extra["config4"] = config4_composed(dev, mel, rank)
        torch.cuda.empty_cache()
        extra["config5"] = config5_composed(dev, rank)
        torch.cuda.empty_cache()

    if rank == 0:
        fl = flops_per_segment(d, L, H, ffn)
        fwd_tflops = B * fl["total"] / (ms_per_step * 1e-3) / 1e12
        line = {
            "metric": "1s-strain segments/sec (encoder fwd)",
            "value": value, "unit": "segments/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if args.precision == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": f"whisper-{args.encoder} encoder fwd, batch {B} log-mel (80x3000) per GPU, "
                                   f"1500 tokens, random-init weights"
                                   + (" (BASELINE configs[1])" if (args.encoder, B) == ("tiny", 256) else ""),
                       "global_batch": B * world, "parallelism": f"dp{world}", "encoder": args.encoder,
                       "streams_per_gpu": 2 if args.split else 1},
            "forward": {"algorithmic_gflop_per_segment": fl["total"] / 1e9, "achieved_tflops_per_gpu": fwd_tflops,
                        "frac_of_bf16_mfma_peak": fwd_tflops / MFMA_BF16_PEAK_TFLOPS},
            "frontend": {"kernel": "logmel (memset + k_logmel_frames_mfma + k_logmel_finalize)",
                         "ms_per_batch": fe_ms, "ms_per_call_incl_host": fe_call_ms,
                         "what": "ms_per_batch = device time of one ops.logmel call replayed from a hipGraph; "
                                 "ms_per_call_incl_host = the eager Python call (allocation of the 245 MB output, ctypes, "
                                 "three enqueues), host-paced",
                         "segments_per_s": B / (fe_ms or fe_call_ms) * 1e3,
                         "algorithmic_gbs": B * (64000 + 960000) / (fe_ms or fe_call_ms) / 1e6,
                         "bound": "two kernels with different roofs: k_logmel_frames_mfma is fp32-MFMA-bound by the dense-DFT "
                                  "form (about 110 us of the call), only k_logmel_finalize (about 39 us for the 245 MB output) "
                                  "is HBM-bound -- per-kernel durations in profiles/*_kernel_stats.md; no single HBM fraction "
                                  "is quoted for the pair"},
            "frontend_qscan": {"kernel": "rDFT GEMM + k_qscan_tiles + k_qscan_interp (parity unpinned, DESIGN.md section 2)",
                               "ms_per_batch": q_ms, "windows_per_s": 2 * B / q_ms * 1e3,
                               "config": f"{2 * B} x 2048 samples, qrange [4, 128], 148 tiles rows, 128 x 128 output"},
            "q_adapter": {"what": f"QTransformAdapter forward on {B} two-detector windows -> [B, 2, 80, 3000] (Q-scan + the CNN's "
                                  "three HIP launches, csrc/qadapter_cnn.hip + gww_qadapter_tail_f32; no library call)", "ms_per_batch": adapter_ms,
                          "tail_kernel_ms_per_detector": tail_ms,
                          "tail_kernel_algorithmic_gbs": B * (80 * 3000 + 32 * 32) * 4 / tail_ms / 1e6,
                          "tail_kernel_frac_of_hbm_peak": B * (80 * 3000 + 32 * 32) * 4 / tail_ms / 1e6 / HBM_PEAK_GBS},
            "pooled_classify": {"what": "encoder.last_token(): same encoder, only last_hidden_state[:, -1] produced "
                                        "(last layer: one query tile of attention, row-wise ops on B rows)",
                                "ms_per_batch": pooled_ms, "segments_per_s_per_gpu": B / pooled_ms * 1e3}
            if pooled_ms else None,
            "dora_step_ms": train["ms"] if train else None,
            "dora_step": train,
            "extra": extra,
        }
        if traced:
            M = B * T_TOK
            es = 2 if args.precision == "bf16" else 4
            # algorithmic FLOPs and HBM bytes of ONE launch of each kernel class
            work = {
                "mel_to_tokens": (0, B * (80 * T_IN * 4 + 80 * (T_IN + 2) * es)),
                "conv1_gelu": (B * fl["conv1"], B * (T_IN + 2) * (80 + d) * es),
                "conv2_gelu_pos": (B * fl["conv2"], B * ((T_IN + 2) * d * es + T_TOK * d * 4)),
                "ln+qkv_proj": (B * 2 * T_TOK * d * 3 * d, M * (d * 4 + 3 * d * es)),
                "attention": (B * fl["attn"], M * 4 * d * es),
                "out_proj": (B * 2 * T_TOK * d * d, M * 2 * d * es),
                "ln+fc1_gelu": (B * 2 * T_TOK * d * ffn, M * (d * 4 + ffn * es)),
                "fc2": (B * 2 * T_TOK * d * ffn, M * (ffn + d) * es),
                # full final LayerNorm: x (+ the pending bf16 delta of the last fc2) in, fp32 out
                "final_layernorm": (0, M * d * (8 + es)),
                "layernorm_rows(B pooled rows)": (0, B * d * (8 + es)),
                "mlp_fused(ln+fc1+gelu+fc2)": (B * 4 * T_TOK * d * ffn, M * d * (4 + 2 + 4 + 2)),
                # + the next layer's LN1 + q/k/v.  What the algorithm needs: x 4 B + delta 2 B in, x_next 4 B + qkv 6 B
                # out = 16 B per element (a second write of the residual stream is NOT algorithmic: it shows up as
                # pmc_hbm_bytes / algorithmic bytes > 1)
                "mlp_fused+next_ln_qkv": (B * (4 * T_TOK * d * ffn + 2 * T_TOK * d * 3 * d), M * d * 16),
                # the last block with the final LayerNorm as its epilogue: x 4 B + ctx 2 B in, last_hidden_state 4 B out
                "mlp_fused+final_layernorm": (B * 4 * T_TOK * d * ffn, M * d * 10),
            }
            if not traced.get("mel_to_tokens", (0, 0))[1]:
                # conv1 reads the fp32 [B, 80, T] features itself (conv1_mel.hip): no token-major copy
                work["conv1_gelu"] = (B * fl["conv1"], B * (80 * T_IN * 4 + (T_IN + 2) * d * es))
            if not traced.get("out_proj", (0, 0))[1] and args.precision == "bf16" and d == 384:
                # out_proj is fused in front of the MLP block (k_mlp_fused<., true>): its FLOPs and bytes belong to that
                # launch -- ctx 2 B in instead of the bf16 delta (2 B), nothing else changes at the HBM boundary
                op = B * 2 * T_TOK * d * d
                for k in ("mlp_fused(ln+fc1+gelu+fc2)", "mlp_fused+next_ln_qkv", "mlp_fused+final_layernorm"):
                    work[k] = (work[k][0] + op, work[k][1])
            rows = []
            for name, (ms, cnt) in traced.items():
                if cnt == 0:
                    continue
                per = ms / cnt
                flops, byts = work[name]
                rows.append({"kernel": name, "launches_per_step": cnt / args.steps, "ms_per_launch": per,
                             "ms_per_step": ms / args.steps, "tflops": flops / per / 1e9 if flops else None,
                             "algorithmic_gbs": byts / per / 1e6})
            line["kernels"] = rows
            # HBM traffic per launch from the PMC passes of the same command (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE,
            # separate runs; profiles/<PMC_TRAFFIC_FILE>).  FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes
            # for 16-B-per-lane reads on gfx950 (an upper bound where a kernel also issues 8-B-per-lane reads).
            pmc, pmc_note = {}, None
            try:
                if args.encoder == "tiny" and B == 256 and args.precision == "bf16":
                    prof = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles",
                                                       PMC_TRAFFIC_FILE)))
                    if prof.get("csrc_hash") == csrc_hash():
                        pmc = prof["kernels"]
                    else:
                        pmc_note = (f"profiles/{PMC_TRAFFIC_FILE} was collected for csrc hash {prof.get('csrc_hash')}, "
                                    f"the kernels now hash to {csrc_hash()}: traffic dropped (stale)")
            except OSError:
                pmc = {}
            line["pmc_traffic_note"] = pmc_note
            for r in rows:
                t = pmc.get(r["kernel"])
                r["pmc_hbm_bytes"] = (t["fetch_bytes_corrected_x2"] + t["write_bytes"]) if t else None
            line["kernels_note"] = ("single-stream pass of the same steps after the timed region: whole-batch launches, "
                                    "each owning the GPU (HIP events on the launch stream)")
            dom = max(rows, key=lambda r: r["ms_per_step"])
            if dom["tflops"]:
                line["roofline"] = {"kernel": dom["kernel"], "bound": "mfma", "achieved": dom["tflops"],
                                    "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                                    "frac": dom["tflops"] / MFMA_BF16_PEAK_TFLOPS, "traffic": dom["pmc_hbm_bytes"],
                                    "ms_per_launch": dom["ms_per_launch"],
                                    "launches_per_step": dom["launches_per_step"]}
            else:
                line["roofline"] = {"kernel": dom["kernel"], "bound": "hbm", "achieved": dom["algorithmic_gbs"],
                                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dom["algorithmic_gbs"] / HBM_PEAK_GBS,
                                    "traffic": dom["pmc_hbm_bytes"], "ms_per_launch": dom["ms_per_launch"],
                                    "launches_per_step": dom["launches_per_step"]}
        if args.isolated and args.precision == "bf16":
            line["kernels_isolated"] = kernel_breakdown(args.encoder, B, dev)
        if not args.no_cpu_baseline and world == 1:     # reported at N = 1 only (rank 0's host cores)
            line["cpu_baseline"] = cpu_baseline(args.encoder)
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        if world > 1:
            dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
