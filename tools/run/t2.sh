export TMPDIR=/tmp PYTHONPATH=.
rocprofv3 -L 2>/dev/null | grep -o "SQ_WAIT_[A-Z_]*\|SQ_ACTIVE_INST_[A-Z_]*\|SQ_INSTS_[A-Z_]*\|SQ_WAVE_CYCLES\|SQ_BUSY_CYCLES\|SQ_INST_CYCLES_VMEM[A-Z_]*\|TCC_HIT\b\|TCC_MISS\b\|TCC_REQ\b\|TCP_PENDING_STALL_CYCLES\|TCP_TCC_READ_REQ\b\|TCP_TA_TCP_STATE_READ\|SQ_LDS_BANK_CONFLICT\|SQ_LDS_IDX_ACTIVE\|TA_BUSY[A-Z_]*\|TCP_TCR_TCP_STALL_CYCLES\|TCP_READ_TAGCONFLICT_STALL_CYCLES\|SQ_VALU_MFMA_BUSY_CYCLES" | sort -u | tr '\n' ' ' > gpurun_out/counters.txt
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "TCC_HIT TCC_MISS TCC_REQ" "TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  rm -rf gpurun_out/pmc_a; mkdir -p gpurun_out/pmc_a
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_a -o a -- python3 tools/run/att_only.py > /dev/null 2> gpurun_out/pmc_a.err
  python tools/pmc_summary.py gpurun_out/pmc_a 2>&1 | grep -A12 "k_attention" | head -14
done
cat gpurun_out/counters.txt
