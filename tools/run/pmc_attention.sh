set -o pipefail
export TMPDIR=/tmp
rm -rf gpurun_out/pmc_att; mkdir -p gpurun_out/pmc_att
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/pmc_att/a -o a -- python3 tools/run/att_only.py > gpurun_out/pmc_att/a.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d gpurun_out/pmc_att/b -o b -- python3 tools/run/att_only.py > gpurun_out/pmc_att/b.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU --output-format csv -d gpurun_out/pmc_att/c -o c -- python3 tools/run/att_only.py > gpurun_out/pmc_att/c.log 2>&1 || echo "c failed"
python3 tools/pmc_summary.py gpurun_out/pmc_att > gpurun_out/pmc_att/summary.txt; cat gpurun_out/pmc_att/summary.txt
