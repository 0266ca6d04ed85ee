#!/usr/bin/env python3
"""Audit a gfx950 .s file for the MFMA hazards hipcc cannot handle once an MFMA sits inside an inline-asm statement
(k_mlp_fused: the main loop's MFMAs are asm so that their order against the hand-placed GELU slices is the program's):

  * a VALU instruction (v_accvgpr_mov / _write / v_mov / ... -- e.g. a register copy hipcc inserts) that writes a source
    register of an asm MFMA needs two wait states in front of that MFMA;
  * a VALU instruction that reads a register an asm v_mfma_f32_32x32x16 (8 passes) wrote needs 11 wait states behind it
    (an intervening MFMA occupies the pipe for a full 8 passes and counts as such).

Wait states: one per instruction, N + 1 for s_nop N.  usage: tools/audit_asm_mfma.py file.s [kernel-name-substring]"""
import re, sys
text = open(sys.argv[1]).read()
want = sys.argv[2] if len(sys.argv) > 2 else ""
def regs_of(tok):
    tok = tok.strip().split(" ")[0] if tok.strip() else ""
    m = re.match(r"([va])\[(\d+):(\d+)\]", tok)
    if m: return {(m.group(1), x) for x in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.match(r"([va])(\d+)$", tok)
    if m: return {(m.group(1), int(m.group(2)))}
    return set()
total = 0
for k in re.split(r"\n(?=_Z[\w]+:)", text):
    name = k.split(":", 1)[0]
    if want not in name or "s_endpgm" not in k: continue
    ins = []; inasm = False
    for ln, l in enumerate(k.split("\n")):
        t = l.strip()
        if t.startswith(";;#ASMSTART"): inasm = True; continue
        if t.startswith(";;#ASMEND"): inasm = False; continue
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"): continue
        op = t.split()[0]
        args = t[len(op):].split(",")
        ws = int(args[0]) + 1 if op == "s_nop" else (8 if op.startswith("v_mfma") else 1)
        ins.append((ln + 1, op, [regs_of(a) for a in args], inasm, ws, t))
    bad = 0; nasm = 0
    for i, (ln, op, args, asm, ws, t) in enumerate(ins):
        if op.startswith("v_mfma") and asm:
            nasm += 1
            src = set().union(*args[1:]) if len(args) > 1 else set()
            gap = 0
            for j in range(i - 1, max(i - 4, -1), -1):
                pl, pop, pargs, pasm, pws, pt = ins[j]
                if gap >= 2: break
                if pop.startswith("v_") and not pop.startswith("v_mfma") and pargs and (pargs[0] & src):
                    print(f"HAZARD {name[:50]} line {ln}: '{pt}' {gap} wait state(s) in front of asm '{t[:60]}'"); bad += 1
                gap += pws
        elif op.startswith("v_") and not op.startswith("v_mfma"):
            src = set().union(*args[1:]) if len(args) > 1 else set()
            if op.startswith("v_accvgpr_read"): src = args[1] if len(args) > 1 else set()
            gap = 0
            for j in range(i - 1, max(i - 12, -1), -1):
                pl, pop, pargs, pasm, pws, pt = ins[j]
                if gap >= 11: break
                if pop.startswith("v_mfma") and pasm and pargs and (pargs[0] & src):
                    print(f"HAZARD {name[:50]} line {ln}: '{t[:60]}' reads the result of asm MFMA at line {pl} after {gap} wait state(s)"); bad += 1
                gap += pws
    print(f"{name[:70]}: {nasm} asm MFMAs, {bad} hazards")
    total += bad
sys.exit(1 if total else 0)
