#!/usr/bin/env python3
"""Diagnostic (run on the GPU box): rebuild gemm_bf16.hip with the given -DGWW_G3_<NAME>=<v> settings, link against the
prebuilt objects of the other sources and time launch_gemm_bf16 on whisper-small's panel shapes, one child process per
build.  usage: tools/gemm_exp.py ABL=0 ABL=1 ABL=2 ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(ROOT, "gw_whisper_amd", "csrc")
out = os.path.join(ROOT, "gpurun_out", "gemm_exp")
os.makedirs(out, exist_ok=True)
objs = [os.path.join(csrc, "build", f) for f in sorted(os.listdir(os.path.join(csrc, "build")))
        if f.endswith(".o") and f not in ("gemm_bf16.o", "gemm_v4.o")]
child = r'''
import os, sys, statistics, torch
sys.path.insert(0, %r)
from gw_whisper_amd import ops
torch.manual_seed(0)
M = 96000 + 256 - 96000 %% 256 if 96000 %% 256 else 96000
M = 96256
res = []
for (N, K, epi) in ((2304, 768, 0), (768, 768, 2), (3072, 768, 1), (768, 3072, 2), (1536, 512, 0), (2048, 512, 1)):
    a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
    b = torch.randn(N, device="cuda"); r = torch.randn(M, N, device="cuda") if epi == 2 else None
    fn = lambda: ops.gemm(a, w, b, epilogue=epi, resid=r)
    if os.environ.get("GEMM_EXP_BLAS"):
        bb = b.bfloat16()
        fn = lambda: torch.nn.functional.linear(a, w, bb)
    fn(); fn(); ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); fn(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) / 3)
    t = statistics.median(ts)
    res.append("N%%d K%%d e%%d: %%.3f ms %%.0f TF/s" %% (N, K, epi, t, 2.0 * M * N * K / t / 1e9))
    del a, w, b, r
print(" | ".join(res))
''' % ROOT
for m in sys.argv[1:] or ["ABL=0"]:
    tag = m.replace("=", "").replace(",", "_")
    src = os.path.join(csrc, "gemm_bf16.hip")
    src4 = os.path.join(csrc, "gemm_v4.hip")
    defs, env = [], {}
    for kv in m.split(","):
        if kv.startswith("SRC="):   # another form of the source file (kept next to the tool's outputs), same headers
            src = os.path.join(ROOT, kv[4:])
        elif kv.startswith("SRC4="):
            src4 = os.path.join(ROOT, kv[5:])
        elif kv.startswith("ENV:"):   # e.g. ENV:GWW_GEMM_V4=0
            k, v = kv[4:].split("="); env[k] = v
        elif kv.startswith("G4_"):
            defs.append(f"-DGWW_{kv}")
        elif kv == "BLAS":            # yardstick: the same shapes through torch (hipBLASLt), bias only
            env["GEMM_EXP_BLAS"] = "1"
        else:
            defs.append(f"-DGWW_G3_{kv}")
    tag = tag.replace("/", "_").replace(".", "_")
    o = os.path.join(out, f"gemm_bf16_{tag}.o")
    so = os.path.join(out, f"libgww_{tag}.so")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=fast", *defs, "-I", csrc, "-c", src, "-o", o], check=True)
    o4 = os.path.join(out, f"gemm_v4_{tag}.o")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=fast", *defs, "-I", csrc, "-c", src4, "-o", o4], check=True)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + objs + [o, o4], check=True)
    r = subprocess.run([sys.executable, "-c", child], env=dict(os.environ, GWW_LIB=so, **env), capture_output=True, text=True)
    print(f"{m}: {r.stdout.strip()} {r.stderr.strip()[-400:] if r.returncode else ''}", flush=True)
