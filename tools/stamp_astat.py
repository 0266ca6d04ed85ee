#!/usr/bin/env python3
"""Diagnostic: build libgww_stamp.so with -DGWW_STAMP and print the per-phase cycle shares of the
A-stationary GEMM (prologue pass 1 / pass 2 / ring wait+barrier / MFMA step / epilogue)."""
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
lib.gww_debug_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
M, d, ffn = 256 * 1500, 384, 1536
dev = torch.device("cuda:0")
x = torch.randn(M, d, device=dev)
h = x.bfloat16()
dl = (torch.randn(M, d, device=dev) * 0.3).bfloat16()
lw, lb = torch.ones(d, device=dev), torch.zeros(d, device=dev)
cases = {
    "ln+qkv": lambda: ops.gemm_astat(x, wq_f, None, 0, ln=(uq, cq)),
    "out": lambda: ops.gemm_astat(h, wo, bo, 0),
    "resid+ln+fc1": lambda: ops.gemm_astat(x, w1_f, None, 1, ln=(u1, c1), delta=dl, return_x=True),
}
wqkv = (torch.randn(3 * d, d, device=dev) / d ** 0.5).bfloat16(); bq = torch.randn(3 * d, device=dev)
wo = (torch.randn(d, d, device=dev) / d ** 0.5).bfloat16(); bo = torch.randn(d, device=dev)
w1 = (torch.randn(ffn, d, device=dev) / d ** 0.5).bfloat16(); b1 = torch.randn(ffn, device=dev)
wq_f, uq, cq = ops.ln_fold_weights(wqkv.float(), lw, lb, bq)
w1_f, u1, c1 = ops.ln_fold_weights(w1.float(), lw, lb, b1)
names = ["prologue pass (load/stats/frags)", "stats transfer + drain", "ring wait+barrier", "frag read + MFMA", "epilogue", "-"]
for name, fn in cases.items():
    fn(); torch.cuda.synchronize()
    buf = (C.c_ulonglong * 8)()
    lib.gww_debug_stamps(buf, 1)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    lib.gww_debug_stamps(buf, 1)
    waves = buf[7]
    tot = sum(buf[i] for i in range(5))
    print(f"{name}: waves {waves}, mean cycles/wave {tot / waves:.0f}")
    for i in range(5):
        print(f"   {names[i]:26s} {buf[i] / waves:10.0f} cyc/wave  {100.0 * buf[i] / tot:5.1f} %")
