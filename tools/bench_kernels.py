#!/usr/bin/env python3
"""Kernel-level timings at the bench shapes (whisper-tiny, B=256 unless --batch) through the C ABI.
Prints one line per kernel: ms, TFLOP/s, GB/s.  Used while tuning; bench.py is the judged number."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import kernel_breakdown, flops_per_segment
from gw_whisper_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--encoder", default="tiny")
args = ap.parse_args()
dev = torch.device("cuda:0")
rows = kernel_breakdown(args.encoder, args.batch, dev)
tot = 0.0
for r in rows:
    tot += r["ms"] * r["launches_per_fwd"]
    tf = f"{r['tflops']:.0f} TF" if r["tflops"] else "   -  "
    print(f"{r['kernel']:16s} {r['ms']:.3f} ms x{r['launches_per_fwd']:2d}  {tf:>8s}  {r['gbs']:.0f} GB/s")
d, L, H, ffn = synth.ENCODER_SIZES[args.encoder]
print(f"sum over layers: {tot:.2f} ms  ({args.batch * flops_per_segment(d, L, H, ffn)['total'] / tot / 1e9:.0f} TF if that were all)")
