#!/usr/bin/env python3
"""Diagnostic (run on the GPU box): rebuild mlp_fused.hip with -DGWW_MF_EXP=<mask> (1 no DMA in the loop, 2 no GELU,
4 no fragment reads in the loop, 8 stream folded onto 128 KB -- results are wrong by design, only the time matters), link
against the prebuilt objects of the other sources, and time k_mlp_fused plain / +qkv at the bench shape in a child process."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(ROOT, "gw_whisper_amd", "csrc")
out = os.path.join(ROOT, "gpurun_out", "mlp_exp")
os.makedirs(out, exist_ok=True)
objs = [os.path.join(csrc, "build", f) for f in sorted(os.listdir(os.path.join(csrc, "build"))) if f.endswith(".o") and f != "mlp_fused.o"]
# arguments: comma-separated macro settings per build, e.g.  EXP=0  EXP=2  EXP=0,NORM=0   (-> -DGWW_MF_EXP=0 -DGWW_MF_NORM=0)
masks = sys.argv[1:] or ["EXP=0", "EXP=1", "EXP=2", "EXP=4"]
child = r'''
import os, sys, statistics, torch
sys.path.insert(0, %r)
from gw_whisper_amd import ops
M, d, F, NQ = 256 * 1500, 384, 1536, 1152
torch.manual_seed(0)
g = lambda *s: torch.randn(*s, device="cuda")
x, dl = g(M, d) * 2, (g(M, d) * 0.5).bfloat16()
w1, b1, w2, b2 = g(F, d) / d ** 0.5, g(F), (g(d, F) / F ** 0.5).bfloat16(), g(d)
wq, bq = g(NQ, d) / d ** 0.5, g(NQ)
ones, zeros = torch.ones(d, device="cuda"), torch.zeros(d, device="cuda")
w1f, u, cb = ops.ln_fold_weights(w1, ones, zeros, b1)
wqf, uq, cq = ops.ln_fold_weights(wq, ones, zeros, bq)
wt0, wt1 = ops.mlp_pack(w1f, w2), ops.mlp_pack(w1f, w2, wqf)
def t(fn):
    fn(); ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) / 2)
    return statistics.median(ts)
print("plain %%.4f ms   +qkv %%.4f ms" %% (t(lambda: ops.mlp_fused(x, dl, wt0, u, cb, b2)), t(lambda: ops.mlp_fused(x, dl, wt1, u, cb, b2, qkv=(uq, cq)))))
''' % ROOT
for m in masks:
    tag = m.replace("=", "").replace(",", "_").replace("+", "").replace("-", "")
    defs = [kv[1:] if kv.startswith("+") else f"-DGWW_MF_{kv}" for kv in m.split(",")]   # "+-fflag" passes a compiler flag
    o = os.path.join(out, f"mlp_fused_{tag}.o")
    so = os.path.join(out, f"libgww_{tag}.so")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=fast", "-mllvm",
                    "-pragma-unroll-threshold=4000000", *defs, "-c", os.path.join(csrc, "mlp_fused.hip"), "-o", o], check=True)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + objs + [o], check=True)
    r = subprocess.run([sys.executable, "-c", child], env=dict(os.environ, GWW_LIB=so), capture_output=True, text=True)
    print(f"{m}: {r.stdout.strip()} {r.stderr.strip()[-300:] if r.returncode else ''}", flush=True)
