#!/usr/bin/env python3
"""Diagnostic (run on the GPU box): rebuild mlp_fused.hip with the given macro settings, link against the prebuilt
objects of the other sources, and time k_mlp_fused<0,true> / <1,true> (out_proj in front, +- the q/k/v tail) at the bench
shape with preallocated buffers, one child process per build.
usage: tools/mlp_exp2.py ABL=0 ABL=1 ABL=2,AHEAD=3 ...   (-> -DGWW_MF_ABL=2 -DGWW_MF_AHEAD=3; "+-flag" passes a compiler flag)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(ROOT, "gw_whisper_amd", "csrc")
out = os.path.join(ROOT, "gpurun_out", "mlp_exp")
os.makedirs(out, exist_ok=True)
objs = [os.path.join(csrc, "build", f) for f in sorted(os.listdir(os.path.join(csrc, "build"))) if f.endswith(".o") and f != "mlp_fused.o"]
masks = sys.argv[1:] or ["ABL=0"]
child = r'''
import os, sys, statistics, torch
sys.path.insert(0, %r)
from gw_whisper_amd import ops
from gw_whisper_amd._lib import lib, check
B = int(os.environ.get("GWW_EXP_B", "256"))
M, d, F, NQ = B * 1500, 384, 1536, 1152
torch.manual_seed(0)
g = lambda *s: torch.randn(*s, device="cuda")
x, ctx = g(M, d) * 2, (g(M, d)).bfloat16()
wo, bo = (g(d, d) / d ** 0.5).bfloat16(), g(d)
w1, b1, w2, b2 = g(F, d) / d ** 0.5, g(F), (g(d, F) / F ** 0.5).bfloat16(), g(d)
wq, bq = g(NQ, d) / d ** 0.5, g(NQ)
ones, zeros = torch.ones(d, device="cuda"), torch.zeros(d, device="cuda")
w1f, u, cb = ops.ln_fold_weights(w1, ones, zeros, b1)
wqf, uq, cq = ops.ln_fold_weights(wq, ones, zeros, bq)
Mp = (M + 127) // 128 * 128
st = torch.cuda.current_stream().cuda_stream
def make(qkv):
    nq = NQ if qkv else 0
    wt = torch.empty((d * d + 2 * d * F + nq * d,), dtype=torch.bfloat16, device="cuda")
    check(lib().gww_mlp_pack_op_bf16(wo.data_ptr(), w1f.data_ptr(), w2.data_ptr(), wqf.data_ptr() if qkv else None, wt.data_ptr(), d, F, nq, st))
    x_out = torch.empty_like(x)
    o = torch.empty((Mp, NQ if qkv else d), dtype=torch.bfloat16, device="cuda")
    def fn():
        check(lib().gww_attn_out_mlp_fused_bf16(x.data_ptr(), ctx.data_ptr(), bo.data_ptr(), x_out.data_ptr(), u.data_ptr(), cb.data_ptr(),
              wt.data_ptr(), b2.data_ptr(), None if qkv else o.data_ptr(), M, d, F, uq.data_ptr() if qkv else None,
              cq.data_ptr() if qkv else None, o.data_ptr() if qkv else None, nq, st))
    fn.keep = (wt, x_out, o)
    return fn
def t(fn):
    fn(); fn(); ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); fn(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) / 3)
    return statistics.median(ts), min(ts)
a, b = t(make(False)), t(make(True))
print("op+mlp %%.4f (min %%.4f) ms   op+mlp+qkv %%.4f (min %%.4f) ms" %% (a[0], a[1], b[0], b[1]))
''' % ROOT
for m in masks:
    tag = m.replace("=", "").replace(",", "_").replace("+", "").replace("-", "").replace(".", "_")
    src = "mlp_fused.hip"
    parts = []
    for kv in m.split(","):
        if kv.startswith("SRC="): src = kv[4:]      # another source file of csrc/ (an A/B against a saved earlier version)
        else: parts.append(kv)
    defs = [kv[1:] if kv.startswith("+") else f"-DGWW_MF_{kv}" for kv in parts]
    o = os.path.join(out, f"mlp_fused_{tag}.o")
    so = os.path.join(out, f"libgww_{tag}.so")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=fast", "-mllvm",
                    "-pragma-unroll-threshold=4000000", *defs, "-c", os.path.join(csrc, src), "-o", o], check=True)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + objs + [o], check=True)
    r = subprocess.run([sys.executable, "-c", child], env=dict(os.environ, GWW_LIB=so), capture_output=True, text=True)
    print(f"{m}: {r.stdout.strip()} {r.stderr.strip()[-400:] if r.returncode else ''}", flush=True)
