#!/usr/bin/env python3
"""Audit a gfx950 .s file: no instruction may touch the destination registers of an inline-asm global load before
the hand-written s_waitcnt vmcnt(N) that covers it (hipcc treats an asm load's outputs as written at the end of the
statement and may copy them away under register pressure -- cdna_hip_programming.md section 5.7).
usage: tools/audit_asm_loads.py file.s [kernel-name-substring]"""
import re, sys
text = open(sys.argv[1]).read()
want = sys.argv[2] if len(sys.argv) > 2 else ""
kernels = re.split(r"\n(?=_Z[\w]+:)", text)
def regs_of(tok):
    tok = tok.strip()
    m = re.match(r"([va])\[(\d+):(\d+)\]", tok)
    if m: return {(m.group(1), x) for x in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.match(r"([va])(\d+)$", tok)
    if m: return {(m.group(1), int(m.group(2)))}
    return set()
total = 0
for k in kernels:
    name = k.split(":", 1)[0]
    if want not in name or "s_endpgm" not in k: continue
    lines = k.split("\n")
    inasm = False; pending = []; bad = 0; nasm = 0
    for ln, l in enumerate(lines):
        t = l.strip()
        if t.startswith(";;#ASMSTART"): inasm = True; continue
        if t.startswith(";;#ASMEND"): inasm = False; continue
        if not t or t.startswith(";") or t.endswith(":"): continue
        op = t.split()[0]
        if inasm and op.startswith("global_load_dword"):
            pending.append(("L", regs_of(t.split(None, 1)[1].split(",")[0]))); nasm += 1; continue
        if op.startswith("global_load_lds") or op.startswith("global_store") or (op.startswith("global_load") and not inasm) or op.startswith("scratch_"):
            pending.append(("O", set()))
        m = re.match(r"s_waitcnt.*vmcnt\((\d+)\)", t)
        if m:
            n = int(m.group(1))
            while len(pending) > n: pending.pop(0)
            continue
        used = set()
        for a in t[len(op):].split(","):
            a = a.strip().split(" ")[0] if a.strip() else ""
            used |= regs_of(a)
        for kind, r in pending:
            if kind == "L" and (r & used):
                print(f"HAZARD {name[:60]} line {ln + 1}: {t}"); bad += 1; break
    print(f"{name[:70]}: {nasm} asm loads, {bad} hazards")
    total += bad
sys.exit(1 if total else 0)
