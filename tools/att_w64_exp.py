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
import re
names = re.search(r"^SRC := (.*)$", open(os.path.join(csrc, "Makefile")).read(), re.M).group(1).split() + ["logmel_host.cpp"]
# (the kernel lives in the laboratory build: make -C gw_whisper_amd/csrc LAB=1 first; GWW_ATT_W64=1 selects it)
objs = [os.path.join(csrc, "build_lab", n.rsplit(".", 1)[0] + ".o") for n in names if n != "attention_w64.hip"]
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
if os.environ.get("GWW_EXP_STAMP"):
    import ctypes as C
    from gw_whisper_amd._lib import lib
    L = lib()
    L.gww_debug_stamps_w64.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    buf = (C.c_ulonglong * 8)()
    L.gww_debug_stamps_w64(buf, 1)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    L.gww_debug_stamps_w64(buf, 1)
    names = ["requests, Q loads, set-up", "first tiles: wait + barrier", "first unit's reference", "ring wait + barrier (24 tiles)",
             "the steps (48)", "last unit, drain", "overflow check, epilogue"]
    w = buf[7]; tot = sum(buf[i] for i in range(7))
    print("   stamps: %%d waves, %%.0f ticks per wave" %% (w, tot / w))
    for i in range(7): print("   %%-34s %%9.0f ticks/wave %%5.1f %%%%" %% (names[i], buf[i] / w, 100.0 * buf[i] / tot))
""" % ROOT
for m in sys.argv[1:] or ["ABL=0"]:
    tag = m.replace("=", "").replace(",", "_").replace(".", "_")
    src, defs = "attention_w64.hip", []
    stamp = False
    for kv in m.split(","):
        if kv.startswith("SRC="): src = kv[4:]
        elif kv.startswith("STAMP"): stamp = True; defs.append("-DGWW_W64_STAMP")
        else: defs.append(f"-DGWW_W64_{kv}")
    o, so = os.path.join(out, f"w64_{tag}.o"), os.path.join(out, f"libgww_{tag}.so")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=fast", "-DGWW_LAB", *defs,
                    "-c", os.path.join(csrc, src), "-o", o], check=True)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + objs + [o], check=True)
    r = subprocess.run([sys.executable, "-c", child], env=dict(os.environ, GWW_LIB=so, GWW_ATT_W64="1", **({"GWW_EXP_STAMP": "1"} if stamp else {})), capture_output=True, text=True)
    print(f"{m}: {r.stdout.strip()} {r.stderr.strip()[-400:] if r.returncode else ''}", flush=True)
