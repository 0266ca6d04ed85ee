"""whisper-base / -small forward at B = 64 and the whisper-tiny / -small DoRA steps: the bench extras alone (ms)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from gw_whisper_amd import synth
from gw_whisper_amd.encoder import WhisperConfig, WhisperEncoder
dev = torch.device("cuda:0")
mel = torch.randn(64, 80, 3000, device=dev)
for name in ("base", "small"):
    enc = WhisperEncoder.from_numpy_state_dict(synth.named_encoder_state_dict(name, seed=0), WhisperConfig.named(name), precision="bf16").to(dev)
    with torch.no_grad():
        ms = bench.time_kernel(lambda: enc.forward_raw(mel, want_hidden=True, want_last=True), iters=8, warm=4)   # (the first measurement of a process runs 20-30 % long with fewer warm-up passes)
    d, L, H, f = synth.ENCODER_SIZES[name]
    tf = 64 * bench.flops_per_segment(d, L, H, f)["total"] / (ms * 1e-3) / 1e12
    print(f"whisper-{name} forward B=64: {ms:.2f} ms  {tf:.0f} TFLOP/s  {tf / bench.MFMA_BF16_PEAK_TFLOPS:.3f} of peak", flush=True)
    del enc
    torch.cuda.empty_cache()
if "--train" in sys.argv:
    for name in ("tiny", "small"):
        r = bench.dora_step(name, 32, dev, 1, steps=4, warmup=2)
        print(f"whisper-{name} DoRA step: {r['ms']:.2f} ms {r['split_ms']}", flush=True)
