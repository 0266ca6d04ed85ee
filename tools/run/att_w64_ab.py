"""Interleaved A/B on the bench shape (B=256, T=1500, H=6), one process (guide rule 24): k_attention_w64_bf16 (64 query
rows per wave, one wave per SIMD; the default) against k_attention_dma_bf16 (GWW_ATT_W64=0, the round-2/3 default).
Also checks the two against each other on the same random input."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gw_whisper_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 7
H = 6
torch.manual_seed(0)
qkv = (torch.randn(B, 1500, 3 * H * 64, device="cuda") * 0.5).bfloat16()
arms = {"dma (3 waves/SIMD)": "0", "w64 (1 wave/SIMD)": "1"}   # needs GWW_LIB=gw_whisper_amd/libgww_lab.so (make LAB=1)
times = {k: [] for k in arms}
outs = {}
def run(name, n=5):
    os.environ["GWW_ATT_W64"] = arms[name]
    outs[name] = ops.attention_log2q(qkv, H)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): ops.attention_log2q(qkv, H)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n
for r in range(rounds):
    for name in arms:
        times[name].append(run(name))
a, b = (outs[k].float() for k in arms)
print("max |w64 - dma| =", float((a - b).abs().max()), " max |ctx| =", float(a.abs().max()))
fl = B * 4 * 1500 * 1500 * 64 * H
for name, t in times.items():
    med = statistics.median(t)
    print(f"{name:20s} median {med:.4f} ms  min {min(t):.4f}  -> {fl / med / 1e9:.0f} TFLOP/s")
