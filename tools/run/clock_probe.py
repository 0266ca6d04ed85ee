"""Is k_mlp_fused power-bound?  Same kernel, same per-workgroup work, 32 ... 3000 workgroups: if the chip were not
throttling, the time per ROUND of workgroups (256 per round) would not depend on how many CUs are busy."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gw_whisper_amd import ops
d, F = 384, 1536
torch.manual_seed(0)
g = lambda *s: torch.randn(*s, device="cuda")
w1, b1, w2, b2 = g(F, d) / d ** 0.5, g(F), (g(d, F) / F ** 0.5).bfloat16(), g(d)
w1f, u, cb = ops.ln_fold_weights(w1, torch.ones(d, device="cuda"), torch.zeros(d, device="cuda"), b1)
wt = ops.mlp_pack(w1f, w2)
for wgs in (32, 64, 128, 256, 512, 1024, 3000):
    M = wgs * 128
    x, dl = g(M, d) * 2, (g(M, d) * 0.5).bfloat16()
    fn = lambda: ops.mlp_fused(x, dl, wt, u, cb, b2)
    for _ in range(3): fn()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); [fn() for _ in range(10)]; e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) / 10)
    t = statistics.median(ts)
    rounds = -(-wgs // 256)
    print(f"{wgs:5d} workgroups ({min(wgs,256):3d} CUs busy): {t * 1e3:8.1f} us per launch, {t * 1e3 / rounds:7.1f} us per round of workgroups")
