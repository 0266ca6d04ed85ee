#!/usr/bin/env python3
"""Diagnostic: -DGWW_STAMP build, per-phase cycle shares of the bf16 attention kernel."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, "gw_whisper_amd", "csrc")
so = os.path.join(ROOT, "gpurun_out", "libgww_stamp.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
srcs = [f for f in sorted(os.listdir(csrc)) if f.endswith(".hip")]
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=fast",
                "-DGWW_STAMP", "-mllvm", "-pragma-unroll-threshold=4000000", "-shared", "-o", so] + [os.path.join(csrc, f) for f in srcs], check=True)
import torch
from gw_whisper_amd import _lib
_lib.LIB_PATH = so
from gw_whisper_amd import ops
lib = _lib.lib()
lib.gww_debug_stamps_att.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
qkv = (torch.randn(B, 1500, 1152, device="cuda") * 0.5).bfloat16()
fn = lambda: ops.attention(qkv, 6)
fn(); torch.cuda.synchronize()
buf = (C.c_ulonglong * 8)()
lib.gww_debug_stamps_att(buf, 1)
for _ in range(3): fn()
torch.cuda.synchronize()
lib.gww_debug_stamps_att(buf, 1)
names = ["prologue+epilogue", "issue next-tile global loads", "S = K Q^T (LDS reads + 8 MFMA)", "softmax (max, exp, cvt)", "O += V^T P (tr reads + 12 MFMA)", "LDS store of next tile", "barrier"]
waves = buf[7]; tot = sum(buf[i] for i in range(7))
print(f"attention B={B}: waves {waves}, cycles/wave {tot / waves:.0f}, per key tile {tot / waves / 24:.0f}")
for i in range(7): print(f"   {names[i]:36s} {buf[i] / waves:9.0f} cyc/wave {100.0 * buf[i] / tot:5.1f} %  ({buf[i] / waves / 24:.0f} per tile)")
