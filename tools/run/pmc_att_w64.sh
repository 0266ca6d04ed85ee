#!/bin/bash
# SQ counters of k_attention_w64_bf16 at the bench shape (separate rocprofv3 --pmc passes, no trace domains)
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_att_w64
mkdir -p $O
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --kernel-trace -d $O/$tag -o out --output-format csv -- python3 $R/tools/run/att_only.py 256 > $O/$tag.log 2>&1 || true
done
python3 $R/tools/pmc_summary.py $O > $O/summary.txt 2>&1 || true
cat $O/summary.txt | tail -40
