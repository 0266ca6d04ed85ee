#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc csv (counter_collection.csv): mean per-dispatch counter per kernel."""
import csv, glob, os, sys, collections
src = sys.argv[1]
files = sorted(glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True))
assert files, "no counter_collection.csv under " + src
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, cs in acc.items():
    if not name.startswith("gww"):
        continue
    print(name)
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
