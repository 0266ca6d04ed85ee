"""Interleaved A/B of k_mlp_fused (plain and with the next layer's LN1 + q/k/v appended) at the bench shape, one process:
GWW_MLP_PAIR=1 (one barrier per two weight tiles) against 0 (one per tile)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gw_whisper_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 7
M, d, F, NQ = B * 1500, 384, 1536, 1152
torch.manual_seed(0)
g = lambda *s: torch.randn(*s, device="cuda")
x, dl = g(M, d) * 2, (g(M, d) * 0.5).bfloat16()
w1, b1, w2, b2 = g(F, d) / d ** 0.5, g(F), (g(d, F) / F ** 0.5).bfloat16(), g(d)
wq, bq = g(NQ, d) / d ** 0.5, g(NQ)
ones, zeros = torch.ones(d, device="cuda"), torch.zeros(d, device="cuda")
w1f, u, cb = ops.ln_fold_weights(w1, ones, zeros, b1)
wqf, uq, cq = ops.ln_fold_weights(wq, ones, zeros, bq)
wt0, wt1 = ops.mlp_pack(w1f, w2), ops.mlp_pack(w1f, w2, wqf)
arms = {"plain pair": ("1", lambda: ops.mlp_fused(x, dl, wt0, u, cb, b2)), "plain single": ("0", lambda: ops.mlp_fused(x, dl, wt0, u, cb, b2)),
        "+qkv pair": ("1", lambda: ops.mlp_fused(x, dl, wt1, u, cb, b2, qkv=(uq, cq))),
        "+qkv single": ("0", lambda: ops.mlp_fused(x, dl, wt1, u, cb, b2, qkv=(uq, cq)))}
times = {k: [] for k in arms}
def run(name):
    pair, fn = arms[name]
    os.environ["GWW_MLP_PAIR"] = pair
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / 3
for r in range(rounds):
    for name in arms: times[name].append(run(name))
for name, t in times.items():
    fl = 4 * M * d * F + (2 * M * d * NQ if "+qkv" in name else 0)
    med = statistics.median(t)
    print(f"{name:14s} median {med:.4f} ms  min {min(t):.4f}  -> {fl / med / 1e9:.0f} TFLOP/s")
