#!/usr/bin/env python3
"""Diagnostic: build libgww_stamp.so with -DGWW_STAMP and print the per-phase cycle shares of the fused MLP kernel."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, "gw_whisper_amd", "csrc")
so = os.path.join(ROOT, "gpurun_out", "libgww_stamp.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
# only mlp_fused.hip is rebuilt (with the stamps); the other objects are the prebuilt ones of the production library
objs = [os.path.join(csrc, "build", f) for f in sorted(os.listdir(os.path.join(csrc, "build"))) if f.endswith(".o") and f != "mlp_fused.o"]
obj = os.path.join(ROOT, "gpurun_out", "mlp_fused_stamp.o")
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=fast",
                "-DGWW_STAMP=" + os.environ.get("GWW_STAMP_MODE", "1"), "-mllvm", "-pragma-unroll-threshold=4000000"]
               + os.environ.get("GWW_EXTRA_DEFS", "").split() + ["-c", os.path.join(csrc, "mlp_fused.hip"), "-o", obj], check=True)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + objs + [obj], check=True)
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
if os.environ.get("GWW_STAMP_QKV") == "1":   # the variant with the next layer's LN1 + q / k / v appended
    wq, bq = torch.randn(1152, d, device=dev) / d ** 0.5, torch.randn(1152, device=dev)
    wq_f, uq, cq = ops.ln_fold_weights(wq, lw, lb, bq)
    wtq = ops.mlp_pack(w1_f, w2, wq_f)
    fn = lambda: ops.mlp_fused(x, dl, wtq, u1, c1, b2, qkv=(uq, cq))
if os.environ.get("GWW_STAMP_OP") == "1":   # out_proj fused in front (ctx instead of delta)
    ctxb = (torch.randn(M, d, device=dev)).bfloat16()
    wo = (torch.randn(d, d, device=dev) / d ** 0.5).bfloat16()
    bo = torch.randn(d, device=dev)
    if os.environ.get("GWW_STAMP_QKV") == "1":
        fn = lambda: ops.attn_out_mlp_fused(x, ctxb, wo, bo, w1_f, w2, u1, c1, b2, qkv=(wq_f, uq, cq))
    else:
        fn = lambda: ops.attn_out_mlp_fused(x, ctxb, wo, bo, w1_f, w2, u1, c1, b2)
names = ["prologue", "ring wait + barrier", "stage bookkeeping", "epilogue", "main loop (light mode)", "x_next + LN1 (q/k/v variant)", "q/k/v tail: ring wait + barrier", "q/k/v tail: tile MFMAs"] + ([f"parity {i // 6} {'fc1' if i % 6 < 3 else 'fc2'} tile {i % 3}" for i in range(12)] if os.environ.get("GWW_STAMP_MODE", "1") != "3" else ["prologue: loads + x_new + pack (since kernel start)", "prologue: row statistics", "prologue: normalise + offsets", "prologue: ring wait + barrier", "G1(0): three tiles", "open GELU of chunk 0", "OP: out_proj GEMM (18 tiles)", "OP: seam x_new + normalise + zero O", "-", "-", "-", "-"])
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
print(f"[{os.environ.get('GWW_EXTRA_DEFS', '')}] mlp_fused B={B}: {ev0.elapsed_time(ev1) / 3:.3f} ms/launch (stamped build); waves {waves}, mean s_memtime ticks/wave {tot / waves:.0f}")
print(f"   shader clock held under the kernel: {100.0 * buf[22] / max(buf[21], 1):.0f} MHz (s_memtime / s_memrealtime ticks over each wave's life)")
for i in [0, 1, 2, 3, 4, 5, 6, 7] + list(range(8, 20)):
    if buf[i]:
        print(f"   {names[i]:52s} {buf[i] / waves:10.0f} ticks/wave  {100.0 * buf[i] / tot:5.1f} %")
