#!/usr/bin/env python3
import json, sys
d = json.loads(open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read())
print(f"value {d['value']:.0f} seg/s  ms/step {d['ms_per_step']:.3f}  frac {d['forward']['frac_of_bf16_mfma_peak']:.4f}")
print("  ".join(f"{r['kernel']}={r['ms_per_launch']:.3f}" for r in d.get("kernels", [])))
if d.get("pooled_classify"):
    print(f"pooled classify: {d['pooled_classify']['ms_per_batch']:.3f} ms/batch  {d['pooled_classify']['segments_per_s_per_gpu']:.0f} seg/s/GPU")
if d.get("dora_step"):
    print("dora step", d["dora_step"]["ms"], d["dora_step"]["split_ms"])
