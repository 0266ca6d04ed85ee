#!/usr/bin/env python3
"""Turn a rocprofv3 --kernel-trace --stats output dir into a small summary under profiles/.
usage: tools/prof_summary.py <rocprof_out_dir> <profiles/name.md> [title]"""
import csv, glob, os, sys
src, dst = sys.argv[1], sys.argv[2]
title = sys.argv[3] if len(sys.argv) > 3 else os.path.basename(dst)
stats = sorted(glob.glob(os.path.join(src, "**", "*_kernel_stats.csv"), recursive=True))
assert stats, "no *_kernel_stats.csv under " + src
rows = list(csv.DictReader(open(stats[0])))
with open(dst, "w") as f:
    f.write(f"# {title}\n\nsource: `rocprofv3 --kernel-trace --stats` ({os.path.basename(stats[0])})\n\n")
    f.write("| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|\n")
    for r in rows:
        if float(r["Percentage"]) < 0.01:
            continue
        name = r["Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        f.write(f"| `{name}` | {r['Calls']} | {int(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.1f} | "
                f"{int(r['MinNs'])/1e3:.1f} | {int(r['MaxNs'])/1e3:.1f} | {float(r['Percentage']):.2f} |\n")
print(open(dst).read())
