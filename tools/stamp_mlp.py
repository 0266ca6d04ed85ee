#!/usr/bin/env python3
"""Diagnostic: build libgww_stamp.so with -DGWW_STAMP and print the per-phase cycle shares of the fused MLP kernel."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, "gw_whisper_amd", "csrc")
so = os.path.join(ROOT, "gpurun_out", "libgww_stamp.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
srcs = [f for f in sorted(os.listdir(csrc)) if f.endswith(".hip")]
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=fast",
                "-DGWW_STAMP", "-mllvm", "-pragma-unroll-threshold=4000000"] + os.environ.get("GWW_EXTRA_DEFS", "").split() + ["-shared", "-o", so] + [os.path.join(csrc, f) for f in srcs], check=True)
import torch
from gw_whisper_amd import _lib
_lib.LIB_PATH = so
from gw_whisper_amd import ops
lib = _lib.lib()
lib.gww_debug_stamps_mlp.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
M, d, ffn = B * 1500, 384, 1536
dev = torch.device("cuda:0")
x = torch.randn(M, d, device=dev)
dl = (torch.randn(M, d, device=dev) * 0.3).bfloat16()
lw, lb = torch.ones(d, device=dev), torch.zeros(d, device=dev)
w1 = (torch.randn(ffn, d, device=dev) / d ** 0.5).bfloat16(); b1 = torch.randn(ffn, device=dev)
w2 = (torch.randn(d, ffn, device=dev) / ffn ** 0.5).bfloat16(); b2 = torch.randn(d, device=dev)
w1_f, u1, c1 = ops.ln_fold_weights(w1.float(), lw, lb, b1)
wt = ops.mlp_pack(w1_f, w2)
fn = lambda: ops.mlp_fused(x, dl, wt, u1, c1, b2)
names = ["prologue", "ring wait + barrier", "stage bookkeeping", "epilogue"] + ["-"] * 4 + [f"parity {i // 6} {'fc1' if i % 6 < 3 else 'fc2'} tile {i % 3}" for i in range(12)]
fn(); torch.cuda.synchronize()
buf = (C.c_ulonglong * 24)()
lib.gww_debug_stamps_mlp(buf, 1)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for _ in range(3):
    fn()
ev1.record()
torch.cuda.synchronize()
lib.gww_debug_stamps_mlp(buf, 1)
waves = buf[23]
tot = sum(buf[i] for i in range(20))
print(f"mlp_fused B={B}: {ev0.elapsed_time(ev1) / 3:.3f} ms/launch (stamped build); waves {waves}, mean s_memtime ticks/wave {tot / waves:.0f}")
for i in [0, 1, 2, 3] + list(range(8, 20)):
    print(f"   {names[i]:26s} {buf[i] / waves:10.0f} ticks/wave  {100.0 * buf[i] / tot:5.1f} %")
