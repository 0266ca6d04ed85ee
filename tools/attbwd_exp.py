#!/usr/bin/env python3
"""Diagnostic (run on the GPU box): rebuild attention_bwd.hip with the given -D settings and time the attention backward
(rowdot + dq + dkv) at the training shape (B = 64, T = 1500, H = 6), one child process per build.
usage: tools/attbwd_exp.py GWW_ATTBWD_DQ_WAVES=2 GWW_ATTBWD_DQ_WAVES=3"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(ROOT, "gw_whisper_amd", "csrc")
out = os.path.join(ROOT, "gpurun_out", "attbwd_exp"); os.makedirs(out, exist_ok=True)
objs = [os.path.join(csrc, "build", f) for f in sorted(os.listdir(os.path.join(csrc, "build"))) if f.endswith(".o") and f != "attention_bwd.o"]
child = r'''
import os, sys, statistics, torch
sys.path.insert(0, %r)
from gw_whisper_amd import ops
B, T, H = 64, 1500, 6
torch.manual_seed(0)
qkv = (torch.randn(B, T, 3 * H * 64, device="cuda") * 0.5).bfloat16()
ctx, lse = ops.attention_lse(qkv, H) if hasattr(ops, "attention_lse") else (None, None)
dctx = (torch.randn(B, T, H * 64, device="cuda") * 0.1).bfloat16()
fn = lambda: ops.attention_bwd(qkv, ctx, dctx, lse, H)
fn(); fn(); ts = []
for _ in range(7):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); fn(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) / 3)
print("attention backward (incl. allocations of the wrapper) %%.4f ms (min %%.4f)" %% (statistics.median(ts), min(ts)))
''' % ROOT
for n, v in enumerate(sys.argv[1:]):
    o = os.path.join(out, f"attention_bwd_{n}.o"); so = os.path.join(out, f"libgww_ab{n}.so")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=fast",
                    *[f"-D{x}" for x in v.split(",")], "-c", os.path.join(csrc, "attention_bwd.hip"), "-o", o], check=True)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + objs + [o], check=True)
    r = subprocess.run([sys.executable, "-c", child], env=dict(os.environ, GWW_LIB=so), capture_output=True, text=True)
    print(f"{v}: {r.stdout.strip()} {r.stderr.strip()[-300:] if r.returncode else ''}", flush=True)
