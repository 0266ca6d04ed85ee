"""Interleaved A/B of the bf16 attention kernels on the bench shape (B=256, T=1500, H=6), one process (guide rule 24):
k_attention_bf16 (natural-unit q) against k_attention_l2_bf16 and its GWW_ATT_VAR / GWW_ATT_WAVES variants."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gw_whisper_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 7
H = 6
torch.manual_seed(0)
qkv = (torch.randn(B, 1500, 3 * H * 64, device="cuda") * 0.5).bfloat16()
arms = {"old(natural q)": (None, None, ops.attention)}
for v in (7, 8, 9):
    arms[f"l2 var{v}"] = (str(v), None, ops.attention_log2q)
times = {k: [] for k in arms}
def run(name):
    var, waves, fn = arms[name]
    for k, v in (("GWW_ATT_VAR", var), ("GWW_ATT_WAVES", waves)):
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = v
    fn(qkv, H)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn(qkv, H)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / 5
for r in range(rounds):
    for name in arms:
        times[name].append(run(name))
fl = B * 4 * 1500 * 1500 * 64 * H
for name, t in times.items():
    med = statistics.median(t)
    print(f"{name:18s} median {med:.4f} ms  min {min(t):.4f}  -> {fl / med / 1e9:.0f} TFLOP/s")
