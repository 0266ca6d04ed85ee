#!/usr/bin/env python3
"""Audit a gfx950 .s file for register spills: per kernel `.vgpr_spill_count` against a ceiling, and NO scratch access inside
a loop (hipcc marks every basic block of a loop with `; in Loop:` / `Loop Header`).  Scratch operations are vector-memory
operations: they retire through the same in-order vmcnt queue the fused kernels count by hand, so a spill inside a tile loop
is a correctness hazard for those counts as much as a slowdown; the few spills left sit in the straight-line prologue / seam code.

usage: tools/audit_spills.py file.s <kernel-name-substring> [<mangled-substring>=<max spills> ...]   (default ceiling 0)"""
import re, sys
text = open(sys.argv[1]).read()
want = sys.argv[2] if len(sys.argv) > 2 else ""
ceil = dict(a.split("=") for a in sys.argv[3:])
spills = {m.group(1): int(m.group(2)) for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)", text)}
bad = 0
for k in re.split(r"\n(?=_Z[\w]+:)", text):
    name = k.split(":", 1)[0]
    if want not in name or "s_endpgm" not in k: continue
    in_loop = False; loop_scratch = 0
    for l in k.split("\n"):
        t = l.strip()
        if re.match(r"\.LBB\d+_\d+:", t):
            in_loop = "Loop" in t
        elif t.startswith("scratch_") and in_loop:
            loop_scratch += 1
    n = spills.get(name, 0)
    limit = max([int(v) for s, v in ceil.items() if s in name] or [0])
    ok = n <= limit and loop_scratch == 0
    print(f"{name[:70]}: {n} spilled VGPRs (ceiling {limit}), {loop_scratch} scratch accesses inside loops{'' if ok else '  <-- FAIL'}")
    bad += not ok
sys.exit(1 if bad else 0)
