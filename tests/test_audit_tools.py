"""The build's ISA audits (``__graft_entry__.build()`` runs them on the generated mlp_fused.s) on synthetic listings:
they must flag the hazards they exist for and stay quiet on clean code."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HEAD = "_Z6kernelv:\n"
TAIL = "\ts_endpgm\n"


def _run(tool, body, tmp_path):
    f = tmp_path / "k.s"
    f.write_text(HEAD + body + TAIL)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), str(f), "kernel"], capture_output=True, text=True)
    return r.returncode, r.stdout


def _asm(line):
    return "\t;;#ASMSTART\n\t" + line + "\n\t;;#ASMEND\n"


def test_asm_mfma_audit_flags_a_register_copy_in_front_of_an_asm_mfma(tmp_path):
    body = "\tv_accvgpr_mov_b32 a80, a32\n" + _asm("v_mfma_f32_32x32x16_bf16 a[80:95], v[54:57], v[162:165], a[80:95]")
    rc, out = _run("audit_asm_mfma.py", body, tmp_path)
    assert rc == 1 and "HAZARD" in out and "1 hazards" in out


def test_asm_mfma_audit_accepts_two_wait_states(tmp_path):
    body = "\tv_accvgpr_mov_b32 a80, a32\n" + _asm("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[80:95], v[54:57], v[162:165], a[80:95]")
    rc, out = _run("audit_asm_mfma.py", body, tmp_path)
    assert rc == 0 and "1 asm MFMAs, 0 hazards" in out


def test_asm_mfma_audit_flags_an_early_read_of_the_result(tmp_path):
    mf = _asm("v_mfma_f32_32x32x16_bf16 v[48:63], v[168:171], v[64:67], v[48:63]")
    rc, out = _run("audit_asm_mfma.py", mf + "\tv_add_f32_e32 v1, v2, v3\n\tv_mul_f32_e32 v136, v48, v48\n", tmp_path)
    assert rc == 1 and "reads the result of asm MFMA" in out
    rc, out = _run("audit_asm_mfma.py", mf + "\ts_nop 10\n\tv_mul_f32_e32 v136, v48, v48\n", tmp_path)
    assert rc == 0
    # an MFMA in between issues 8 wait states after the first (the pipe is busy) and the reader 3 + 1 behind it
    other = _asm("v_mfma_f32_32x32x16_bf16 v[16:31], v[168:171], v[64:67], v[16:31]")
    rc, out = _run("audit_asm_mfma.py", mf + other + "\ts_nop 2\n\tv_mul_f32_e32 v136, v48, v48\n", tmp_path)
    assert rc == 0
    # ... but it only WAITS for the pipe, it does not add 8 on top of what already went by: A, 3 VALU, B, reader has
    # about 8 wait states, not 3 + 8 (round 2's model passed this listing)
    valu3 = "\tv_add_f32_e32 v1, v2, v3\n" * 3
    rc, out = _run("audit_asm_mfma.py", mf + valu3 + other + "\tv_mul_f32_e32 v136, v48, v48\n", tmp_path)
    assert rc == 1 and "after 8 wait state" in out


def test_asm_mfma_audit_sees_stores_and_mfma_operands_reading_a_result(tmp_path):
    mf = _asm("v_mfma_f32_32x32x16_bf16 v[48:63], v[168:171], v[64:67], v[48:63]")
    for reader in ("ds_write_b64 v200, v[48:49] offset:16", "global_store_dwordx4 v[2:3], v[52:55], off",
                   "buffer_store_dwordx2 v[60:61], v7, s[0:3], 0 offen",
                   "v_mfma_f32_32x32x16_bf16 a[0:15], v[48:51], v[64:67], a[0:15]"):
        rc, out = _run("audit_asm_mfma.py", mf + "\ts_nop 3\n\t" + reader + "\n", tmp_path)
        assert rc == 1 and "reads the result of asm MFMA" in out, reader
        rc, out = _run("audit_asm_mfma.py", mf + "\ts_nop 10\n\t" + reader + "\n", tmp_path)
        assert rc == 0, reader
    # the accumulate chain itself (same registers as SrcC and destination) is not a hazard
    rc, out = _run("audit_asm_mfma.py", mf + mf, tmp_path)
    assert rc == 0


def test_asm_load_audit_flags_a_touch_before_the_covering_wait(tmp_path):
    ld = _asm("global_load_dwordx4 v[10:13], v[2:3], off offset:0")
    rc, out = _run("audit_asm_loads.py", ld + "\tv_mov_b32_e32 v20, v10\n\ts_waitcnt vmcnt(0)\n", tmp_path)
    assert rc == 1 and "HAZARD" in out
    rc, out = _run("audit_asm_loads.py", ld + "\ts_waitcnt vmcnt(0)\n\tv_mov_b32_e32 v20, v10\n", tmp_path)
    assert rc == 0 and "1 asm loads, 0 hazards" in out


def test_spill_audit_flags_a_scratch_access_inside_a_loop_and_a_count_over_its_ceiling(tmp_path):
    meta = ("amdhsa.kernels:\n  - .name:           _Z6kernelv\n    .private_segment_fixed_size: 8\n    .vgpr_count:     512\n"
            "    .vgpr_spill_count: 2\n")
    loop = ".LBB0_1:                                ; =>This Inner Loop Header: Depth=1\n\tscratch_load_dword v1, off, off offset:4\n\ts_cbranch_scc1 .LBB0_1\n"
    flat = "\tscratch_store_dword off, v1, off offset:4\n.LBB0_2:\n\tscratch_load_dword v1, off, off offset:4\n"
    def run(body, *ceil):
        f = tmp_path / "k.s"
        f.write_text(HEAD + body + TAIL + meta)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "audit_spills.py"), str(f), "kernel", *ceil],
                           capture_output=True, text=True)
        return r.returncode, r.stdout
    rc, out = run(flat, "kernel=2")
    assert rc == 0 and "2 spilled VGPRs (ceiling 2), 0 scratch accesses inside loops" in out
    rc, out = run(flat)                       # default ceiling 0
    assert rc == 1 and "FAIL" in out
    rc, out = run(flat + loop, "kernel=2")    # within the ceiling, but inside a loop
    assert rc == 1 and "1 scratch accesses inside loops" in out


# ---- tools/audit_asm_preload.py (gemm_v4.hip: the fp32 residual requested straight into the accumulator registers)
_PRE_LOAD = _asm("global_load_dwordx4 v[116:119], v[100:101], off")
_PRE_WAIT = _asm("s_waitcnt vmcnt(8)")


def test_asm_preload_audit_accepts_load_wait_mfma(tmp_path):
    body = _PRE_LOAD + "\tv_add_f32_e32 v1, v2, v3\n" + _PRE_WAIT + "\tv_mfma_f32_16x16x32_bf16 v[120:123], v[8:11], v[12:15], v[116:119]\n"
    rc, out = _run("audit_asm_preload.py", body, tmp_path)
    assert rc == 0 and "1 asm loads checked, 0 problems" in out


def test_asm_preload_audit_flags_a_copy_an_unwaited_mfma_and_a_spill(tmp_path):
    # a register copy of the destination before the data can have landed
    rc, out = _run("audit_asm_preload.py", _PRE_LOAD + "\tv_mov_b64_e32 v[118:119], v[66:67]\n" + _PRE_WAIT, tmp_path)
    assert rc == 1 and "AUDIT FAIL" in out
    # the accumulating MFMA without any vmcnt wait in between
    rc, out = _run("audit_asm_preload.py", _PRE_LOAD + "\tv_mfma_f32_16x16x32_bf16 v[116:119], v[8:11], v[12:15], v[116:119]\n", tmp_path)
    assert rc == 1 and "waited=False" in out
    # a spill anywhere in the kernel (it would sit uncounted in the hand-counted vmcnt queue)
    body = "\tscratch_store_dwordx2 off, v[8:9], off\n" + _PRE_LOAD + _PRE_WAIT + "\tv_mfma_f32_16x16x32_bf16 v[116:119], v[8:11], v[12:15], v[116:119]\n"
    rc, out = _run("audit_asm_preload.py", body, tmp_path)
    assert rc == 1 and "scratch access" in out


def test_asm_preload_audit_follows_an_unconditional_branch_over_the_else_block(tmp_path):
    # then-block: the loads; else-block (behind s_branch): zeroing moves of the same registers -- not a use of the loaded data
    body = (_PRE_LOAD + "\ts_branch .LBB0_9\n.LBB0_8:\n\tv_mov_b64_e32 v[118:119], v[66:67]\n.LBB0_9:\n" + _PRE_WAIT +
            "\tv_mfma_f32_16x16x32_bf16 v[120:123], v[8:11], v[12:15], v[116:119]\n")
    rc, out = _run("audit_asm_preload.py", body, tmp_path)
    assert rc == 0, out
