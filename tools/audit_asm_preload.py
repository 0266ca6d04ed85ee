#!/usr/bin/env python3
"""Build-time audit of gemm_v4.hip's fp32-residual kernel: its epilogue requests the NEXT n-tile's residual tile straight
into the accumulator registers by inline-asm global_load_dwordx4 (hipcc does not know the destination is written
asynchronously).  Check in the ISA that, after every such load, the first instruction that names one of its four
destination registers is a v_mfma that reads them as its C operand, that at least one s_waitcnt
vmcnt lies in between, and that no scratch (spill) access exists in the kernel -- a copy, an early read or a spill would
use the registers before the data has landed / would sit uncounted in the hand-counted vmcnt queue.
Scope: the walk follows the LAYOUT path (fall-through and unconditional branches) to the first instruction that names a
destination register.  Taken conditional branches are not explored: a worklist over both successors (tried in round 4)
reports only infeasible paths here -- the kernel's conditions (first / last / has_next) are correlated across blocks, and
the registers are legitimately reused behind the loop -- so a copy on a taken-branch path would go unseen; the scratch
check and the kernel's fp64 / split-invariance tests are the backstop for that case.
usage: audit_asm_preload.py file.s kernel-name-substring"""
import re, sys

def regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()

def audit(lines, scratch_only=False):
    bad = []
    ins, label_at = [], {}
    for i, l in enumerate(lines):
        m = re.match(r"^(\.LBB[\w]+):", l)
        if m: label_at[m.group(1)] = len(ins)
        elif l.startswith("\t") and not l.strip().startswith((";", ".")): ins.append((i, l.strip()))
    n_loads = 0
    for idx, (ln, l) in enumerate(ins):
        if "scratch_" in l:
            bad.append(f"line {ln}: scratch access: {l}")
        if scratch_only or not l.startswith("global_load_dwordx4"): continue
        dst = regs(l.split()[1].rstrip(","))
        n_loads += 1
        waited = False
        pos, steps = idx + 1, 0
        while pos < len(ins) and steps < 20000:
            ln2, l2 = ins[pos]; pos += 1; steps += 1
            if l2.startswith("s_branch"):   # unconditional: follow it (a then-block jumping over its else-block)
                pos = label_at.get(l2.split()[1], pos); continue
            if l2.startswith("s_endpgm"): break
            if l2.startswith("s_waitcnt") and "vmcnt" in l2: waited = True
            toks = re.findall(r"v\[\d+:\d+\]|v\d+", l2)
            used = set()
            for t in toks: used |= regs(t)
            if not (used & dst): continue
            ops = [t.strip() for t in l2.split(None, 1)[1].split(",")]
            if l2.startswith("v_mfma") and regs(ops[3].split()[0]) == dst and waited: break   # (hipcc may rotate the destination)
            bad.append(f"line {ln}: {l}  -> first use line {ln2}: {l2} (waited={waited})")
            break
    return n_loads, bad

if __name__ == "__main__":
    text = open(sys.argv[1]).read().split("\n")
    name = sys.argv[2]
    start = next(i for i, l in enumerate(text) if name in l and re.match(r"^[A-Za-z_][\w$.]*:", l))
    # the kernel's body ends at its .Lfunc_end marker (an s_endpgm may sit in the middle, behind an early-exit branch)
    end = next((i for i in range(start, len(text)) if text[i].startswith(".Lfunc_end")),
               max(i for i in range(start, len(text)) if "s_endpgm" in text[i]) + 1)
    # --scratch-only: a kernel whose global loads are hipcc's own (counted and waited for by the compiler: the conv2 form's
    # position rows) -- only the no-spill requirement applies
    n, bad = audit(text[start:end], scratch_only="--scratch-only" in sys.argv)
    for b in bad: print("AUDIT FAIL:", b)
    print(f"audit_asm_preload: {n} asm loads checked, {len(bad)} problems")
    sys.exit(1 if bad or (n == 0 and "--no-loads" not in sys.argv and "--scratch-only" not in sys.argv) else 0)   # --no-loads: only the scratch check applies
