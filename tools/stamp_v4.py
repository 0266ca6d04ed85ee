#!/usr/bin/env python3
"""Diagnostic (GPU box): build gemm_v4.hip with -DGWW_G4_STAMP (+ optional extra defines) and print where the waves of
k_gemm_bf16_v4 spend their s_memtime ticks, per wave group, phase and section.  usage: tools/stamp_v4.py [N K epi] [-D...]"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, "gw_whisper_amd", "csrc")
out = os.path.join(ROOT, "gpurun_out", "stamp_v4"); os.makedirs(out, exist_ok=True)
args = [a for a in sys.argv[1:] if not a.startswith("-D")]
defs = [a for a in sys.argv[1:] if a.startswith("-D")]
N, K, epi = (int(args[0]), int(args[1]), int(args[2])) if len(args) >= 3 else (2304, 768, 0)
objs = [os.path.join(csrc, "build", f) for f in sorted(os.listdir(os.path.join(csrc, "build"))) if f.endswith(".o") and f != "gemm_v4.o"]
o = os.path.join(out, "gemm_v4_stamp.o"); so = os.path.join(out, "libgww_stamp_v4.so")
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=fast", "-DGWW_G4_STAMP", *defs,
                "-I", csrc, "-c", os.path.join(csrc, "gemm_v4.hip"), "-o", o], check=True)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + objs + [o], check=True)
import torch
from gw_whisper_amd import _lib
_lib.LIB_PATH = so
from gw_whisper_amd import ops
lib = _lib.lib()
lib.gww_debug_stamps_v4.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
M = 96256
a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
b = torch.randn(N, device="cuda"); r = torch.randn(M, N, device="cuda") if epi == 2 else None
fn = lambda: ops.gemm(a, w, b, epilogue=epi, resid=r)
fn(); fn(); torch.cuda.synchronize()
buf = (C.c_ulonglong * 32)()
lib.gww_debug_stamps_v4(buf, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); fn(); fn(); fn(); e1.record(); torch.cuda.synchronize()
lib.gww_debug_stamps_v4(buf, 1)
ms = e0.elapsed_time(e1) / 3
print(f"{defs} N{N} K{K} e{epi}: {ms:.3f} ms {2.0 * M * N * K / ms / 1e9:.0f} TF/s (stamped build)")
ktiles = 3 * (M // 256) * (N // 256) * (K // 64)   # per wave group: every wave runs its block's k-tiles
for g in range(2):
    waves = buf[24 + g]
    tot = sum(buf[g * 12 + i] for i in range(12))
    per_kt = tot / (ktiles * 4)   # 4 waves per group and block
    print(f" group {g}: {per_kt:.0f} ticks per k-tile and wave")
    for p in range(4):
        s = [buf[g * 12 + p * 3 + c] / (ktiles * 4) for c in range(3)]
        print(f"   phase {p}: load section (reads, requests, ring wait, barrier, read wait) {s[0]:6.0f} | 16 MFMAs {s[1]:6.0f} | closing barrier {s[2]:6.0f}")
