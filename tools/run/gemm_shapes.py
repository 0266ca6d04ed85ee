"""Time ops.gemm on the layer GEMMs of whisper-base / -small at B = 64 (M = 96 000 -> 96 256 rows), optionally with the column
split forced (GWW_G4_NSPLIT).  usage: gemm_shapes.py [nsplit ...]"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gw_whisper_amd import ops
M = 96256
shapes = [("base qkv", 1536, 512, 0), ("base out", 512, 512, 2), ("base fc1", 2048, 512, 1), ("base fc2", 512, 2048, 2),
          ("small qkv", 2304, 768, 0), ("small out", 768, 768, 2), ("small fc1", 3072, 768, 1), ("small fc2", 768, 3072, 2)]
forced = sys.argv[1:] or [""]
for name, N, K, epi in shapes:
    a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
    b = torch.randn(N, device="cuda"); r = torch.randn(M, N, device="cuda") if epi == 2 else None
    out = []
    for f in forced:
        if f: os.environ["GWW_G4_NSPLIT"] = f
        else: os.environ.pop("GWW_G4_NSPLIT", None)
        fn = lambda: ops.gemm(a, w, b, epilogue=epi, resid=r)
        fn(); fn(); ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); fn(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) / 3)
        t = statistics.median(ts)
        out.append(f"split {f or 'auto'}: {t:.3f} ms {2.0 * M * N * K / t / 1e9:.0f} TF/s")
    print(f"{name:10s} N{N} K{K} e{epi}: " + " | ".join(out), flush=True)
    del a, w, b, r
