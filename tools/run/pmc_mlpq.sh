# round 3: SQ counters of k_mlp_fused<1, true> at B = 20 (one lock-step round) and B = 256 (steady state)
set -o pipefail
export TMPDIR=/tmp
rm -rf gpurun_out/pmc_mlpq; mkdir -p gpurun_out/pmc_mlpq
for b in 20 256; do
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/pmc_mlpq/a$b -o a -- python3 tools/run/mlpq_only.py $b > gpurun_out/pmc_mlpq/a$b.log 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_WAIT_INST_VMEM --output-format csv -d gpurun_out/pmc_mlpq/b$b -o b -- python3 tools/run/mlpq_only.py $b > gpurun_out/pmc_mlpq/b$b.log 2>&1 || echo "b failed"
  timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d gpurun_out/pmc_mlpq/g$b -o g -- python3 tools/run/mlpq_only.py $b > gpurun_out/pmc_mlpq/g$b.log 2>&1 || echo "g failed"
  echo "== B = $b"; for x in a b g; do python3 tools/pmc_summary.py gpurun_out/pmc_mlpq/$x$b | grep -A10 "k_mlp_fused<1"; done
done > gpurun_out/pmc_mlpq/summary.txt 2>&1
cat gpurun_out/pmc_mlpq/summary.txt
