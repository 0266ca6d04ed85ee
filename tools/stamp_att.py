#!/usr/bin/env python3
"""Diagnostic: -DGWW_STAMP build, per-phase cycle shares of the bf16 attention kernel."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, "gw_whisper_amd", "csrc")
so = os.path.join(ROOT, "gpurun_out", "libgww_stamp.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
# only attention.hip is rebuilt (with the stamps); the other objects are the prebuilt ones of the production library
objs = [os.path.join(csrc, "build", f) for f in sorted(os.listdir(os.path.join(csrc, "build"))) if f.endswith(".o") and f != "attention.o"]
obj = os.path.join(ROOT, "gpurun_out", "attention_stamp.o")
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=fast", "-DGWW_STAMP",
                "-I", csrc, "-c", os.path.join(csrc, "attention.hip"), "-o", obj], check=True)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + objs + [obj], check=True)
import torch
from gw_whisper_amd import _lib
_lib.LIB_PATH = so
from gw_whisper_amd import ops
lib = _lib.lib()
lib.gww_debug_stamps_att.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
qkv = (torch.randn(B, 1500, 1152, device="cuda") * 0.5).bfloat16()
fn = (lambda: ops.attention_log2q(qkv, 6)) if os.environ.get("GWW_ATT_VAR") else (lambda: ops.attention(qkv, 6))
fn(); torch.cuda.synchronize()
buf = (C.c_ulonglong * 8)()
lib.gww_debug_stamps_att(buf, 1)
for _ in range(3): fn()
torch.cuda.synchronize()
lib.gww_debug_stamps_att(buf, 1)
if os.environ.get("GWW_ATT_VAR") == "9":
    names = ["K-fragment reads + their wait", "ten score MFMAs with the softmax chunks of the tile before", "V-fragment wait + eight P V MFMAs", "LDS-DMA request + ring wait", "barrier", "-", "prologue + last tile + epilogue"]
elif os.environ.get("GWW_ATT_VAR") == "8":
    names = ["M section, waves 0-3 (score + P V MFMAs, fragment reads, LDS-DMA request, ring wait) [x2: per wave of the group]", "V section, waves 0-3 (softmax VALU) [x2]", "M section, waves 4-7 [x2]", "V section, waves 4-7 [x2]", "barrier behind the M section", "barrier behind the V section", "prologue + last P V + epilogue"]
else:
  names = ["prologue+epilogue", "issue next-tile global loads", "S = K Q^T (LDS reads + 8 MFMA)", "softmax (max, exp, cvt)", "O += V^T P (tr reads + 12 MFMA)", "LDS store of next tile", "barrier"]
waves = buf[7]; tot = sum(buf[i] for i in range(7))
print(f"attention B={B}: waves {waves}, cycles/wave {tot / waves:.0f}, per key tile {tot / waves / 24:.0f}")
for i in range(7): print(f"   {names[i]:80s} {buf[i] / waves:9.0f} cyc/wave {100.0 * buf[i] / tot:5.1f} %  ({buf[i] / waves / 24:.0f} per tile)")
