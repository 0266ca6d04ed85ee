#!/usr/bin/env python3
"""Diagnostic (run on the GPU box): rebuild attention_w64.hip with the given -DGWW_W64_<KEY>=<value> settings, link against
the prebuilt product objects, and time k_attention_w64_bf16 at the bench shape (B = 256, T = 1500, H = 6), one child
process per build (ablation builds give wrong results by design: only the time matters).
usage: tools/att_w64_exp.py ABL=0 ABL=1 ABL=7 SRC=other.hip,ABL=0 ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(ROOT, "gw_whisper_amd", "csrc")
out = os.path.join(ROOT, "gpurun_out", "att_exp")
os.makedirs(out, exist_ok=True)
objs = [os.path.join(csrc, "build", f) for f in sorted(os.listdir(os.path.join(csrc, "build"))) if f.endswith(".o") and f != "attention_w64.o"]
child = r"""
import os, sys, statistics, torch
sys.path.insert(0, %r)
from gw_whisper_amd import ops
B, H = int(os.environ.get("GWW_EXP_B", "256")), 6
torch.manual_seed(0)
qkv = (torch.randn(B, 1500, 3 * H * 64, device="cuda") * 0.5).bfloat16()
fn = lambda: ops.attention_log2q(qkv, H)
fn(); fn(); ts = []
for _ in range(7):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); fn(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) / 3)
print("%%.4f ms (min %%.4f)" %% (statistics.median(ts), min(ts)))
""" % ROOT
for m in sys.argv[1:] or ["ABL=0"]:
    tag = m.replace("=", "").replace(",", "_").replace(".", "_")
    src, defs = "attention_w64.hip", []
    for kv in m.split(","):
        if kv.startswith("SRC="): src = kv[4:]
        else: defs.append(f"-DGWW_W64_{kv}")
    o, so = os.path.join(out, f"w64_{tag}.o"), os.path.join(out, f"libgww_{tag}.so")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=fast", *defs,
                    "-c", os.path.join(csrc, src), "-o", o], check=True)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + objs + [o], check=True)
    r = subprocess.run([sys.executable, "-c", child], env=dict(os.environ, GWW_LIB=so), capture_output=True, text=True)
    print(f"{m}: {r.stdout.strip()} {r.stderr.strip()[-400:] if r.returncode else ''}", flush=True)
