#!/usr/bin/env python3
"""Diagnostic (run on the GPU box): rebuild attention.hip with the given macro settings, link against the prebuilt objects of the
other sources and time the bf16 attention forward in log2 units (k_attention_dma_bf16) at the bench shape (B = 256, T = 1500,
H = 6), one child process per build; the result is checked against the first build.
usage: tools/attfwd_exp.py NBUF=2 NBUF=3 ...   (-> -DGWW_ATT_NBUF=3)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(ROOT, "gw_whisper_amd", "csrc")
out = os.path.join(ROOT, "gpurun_out", "attfwd_exp")
os.makedirs(out, exist_ok=True)
srcs = open(os.path.join(csrc, "Makefile")).read().split("SRC :=")[1].split("\n")[0].split()
objs = [os.path.join(csrc, "build", f.replace(".hip", ".o")) for f in srcs if f != "attention.hip"] + [os.path.join(csrc, "build", "logmel_host.o")]
child = r'''
import os, sys, statistics, torch
sys.path.insert(0, %r)
from gw_whisper_amd import ops
B, T, H = int(os.environ.get("GWW_EXP_B", "256")), 1500, 6
torch.manual_seed(0)
qkv = (torch.randn(B, T, 3 * H * 64, device="cuda") * 0.7).bfloat16()
fn = lambda: ops.attention_log2q(qkv, H)
r = fn(); fn(); ts = []
for _ in range(7):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); fn(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) / 3)
ref = os.environ.get("GWW_EXP_REF")
msg = ""
if ref and os.path.exists(ref):
    a = torch.load(ref).cuda().float(); d = (r.float() - a).abs().max().item(); msg = " max|d - first build| = %%.3e (max|ref| %%.3f)" %% (d, a.abs().max().item())
elif ref:
    torch.save(r.cpu(), ref)
print("attention forward %%.4f (min %%.4f) ms%%s" %% (statistics.median(ts), min(ts), msg))
''' % ROOT
ref = os.path.join(out, "ref.pt")
if os.path.exists(ref): os.remove(ref)
for m in sys.argv[1:] or ["DMA=1"]:
    tag = m.replace("=", "").replace(",", "_")
    defs = [f"-DGWW_ATT_{kv}" for kv in m.split(",")]
    o, so = os.path.join(out, f"attention_{tag}.o"), os.path.join(out, f"libgww_{tag}.so")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=fast", *defs, "-c",
                    os.path.join(csrc, "attention.hip"), "-o", o], check=True)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + objs + [o], check=True)
    r = subprocess.run([sys.executable, "-c", child], env=dict(os.environ, GWW_LIB=so, GWW_EXP_REF=ref), capture_output=True, text=True)
    print(f"{m}: {r.stdout.strip()} {r.stderr.strip()[-600:] if r.returncode else ''}", flush=True)
