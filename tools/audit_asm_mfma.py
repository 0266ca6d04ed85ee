#!/usr/bin/env python3
"""Audit a gfx950 .s file for the MFMA hazards hipcc cannot handle once an MFMA sits inside an inline-asm statement
(k_mlp_fused: the main loop's MFMAs are asm so that their order against the hand-placed GELU slices is the program's):

  * a VALU instruction (v_accvgpr_mov / _write / v_mov / ... -- e.g. a register copy hipcc inserts) that writes a source
    register of an asm MFMA needs two wait states in front of that MFMA;
  * an instruction that reads a register an asm v_mfma_f32_32x32x16 (8 passes) wrote -- a VALU instruction, an LDS /
    global / buffer store (data or address), or another MFMA taking it as SrcA / SrcB (SrcC of the same accumulate chain
    is exempt) -- needs 11 wait states behind it.

Timing model: every instruction issues one wait state after its predecessor (s_nop N: N + 1), and an MFMA issues no
earlier than 8 wait states after the previous MFMA (the matrix pipe is busy for its 8 passes; an MFMA that would have
had to wait anyway adds nothing on top of the ordinary instructions around it -- round 2's model summed 8 per
intervening MFMA and could over-count the distance).  usage: tools/audit_asm_mfma.py file.s [kernel-name-substring]"""
import re, sys
text = open(sys.argv[1]).read()
want = sys.argv[2] if len(sys.argv) > 2 else ""
MFMA_PASSES, NEED_AFTER, NEED_BEFORE = 8, 11, 2
def regs_of(tok):
    tok = tok.strip().split(" ")[0] if tok.strip() else ""
    m = re.match(r"([va])\[(\d+):(\d+)\]", tok)
    if m: return {(m.group(1), x) for x in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.match(r"([va])(\d+)$", tok)
    if m: return {(m.group(1), int(m.group(2)))}
    return set()
STORE = ("ds_write", "ds_store", "global_store", "buffer_store", "flat_store", "scratch_store", "global_atomic", "buffer_atomic")
total = 0
for k in re.split(r"\n(?=_Z[\w]+:)", text):
    name = k.split(":", 1)[0]
    if want not in name or "s_endpgm" not in k: continue
    ins = []; inasm = False; t_issue = 0; last_mfma = -10 ** 9
    for ln, l in enumerate(k.split("\n")):
        t = l.strip()
        if t.startswith(";;#ASMSTART"): inasm = True; continue
        if t.startswith(";;#ASMEND"): inasm = False; continue
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"): continue
        op = t.split()[0]
        args = t[len(op):].split(",")
        if op.startswith("v_mfma"):
            t_issue = max(t_issue, last_mfma + MFMA_PASSES)
            last_mfma = t_issue
        ins.append((ln + 1, op, [regs_of(a) for a in args], inasm, t_issue, t))
        t_issue += int(args[0]) + 1 if op == "s_nop" else 1
    bad = 0; nasm = 0
    for i, (ln, op, args, asm, ti, t) in enumerate(ins):
        is_mfma = op.startswith("v_mfma")
        if is_mfma and asm:
            nasm += 1
            src = set().union(*args[1:]) if len(args) > 1 else set()
            for j in range(i - 1, max(i - 4, -1), -1):
                pl, pop, pargs, pasm, pti, pt = ins[j]
                if ti - pti - 1 >= NEED_BEFORE: break
                if pop.startswith("v_") and not pop.startswith("v_mfma") and pargs and (pargs[0] & src):
                    print(f"HAZARD {name[:50]} line {ln}: '{pt}' {ti - pti - 1} wait state(s) in front of asm '{t[:60]}'"); bad += 1
        # readers of an asm MFMA's result
        if is_mfma:
            dst = args[0] if args else set()
            src = set().union(*args[1:3]) if len(args) > 2 else set()          # SrcA, SrcB
            if len(args) > 3 and args[3] != dst: src |= args[3]                 # a foreign SrcC is an ordinary read
        elif op.startswith("v_accvgpr_read"):
            src = args[1] if len(args) > 1 else set()
        elif op.startswith("v_"):
            src = set().union(*args[1:]) if len(args) > 1 else set()
        elif op.startswith(STORE):
            src = set().union(*args) if args else set()
        else:
            continue
        if not src: continue
        for j in range(i - 1, -1, -1):
            pl, pop, pargs, pasm, pti, pt = ins[j]
            if ti - pti - 1 >= NEED_AFTER: break
            if pop.startswith("v_mfma") and pasm and pargs and (pargs[0] & src):
                print(f"HAZARD {name[:50]} line {ln}: '{t[:60]}' reads the result of asm MFMA at line {pl} after {ti - pti - 1} wait state(s)"); bad += 1
    print(f"{name[:70]}: {nasm} asm MFMAs, {bad} hazards")
    total += bad
sys.exit(1 if total else 0)
