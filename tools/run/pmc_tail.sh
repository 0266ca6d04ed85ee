# round 3: SQ counters of the q/k/v tail in isolation (k_mlp_fused<2, false>) at B = 20 (one lock-step round) and B = 256
set -o pipefail
export TMPDIR=/tmp
rm -rf gpurun_out/pmc_tail; mkdir -p gpurun_out/pmc_tail
rocprofv3 -L > gpurun_out/pmc_tail/counters.txt 2>&1 || true
for b in 20 256; do
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/pmc_tail/a$b -o a -- python3 tools/run/lnqkv_only.py $b > gpurun_out/pmc_tail/a$b.log 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_WAIT_INST_VMEM --output-format csv -d gpurun_out/pmc_tail/b$b -o b -- python3 tools/run/lnqkv_only.py $b > gpurun_out/pmc_tail/b$b.log 2>&1 || echo "b failed"
  timeout -k 10 300 rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES --output-format csv -d gpurun_out/pmc_tail/c$b -o c -- python3 tools/run/lnqkv_only.py $b > gpurun_out/pmc_tail/c$b.log 2>&1 || echo "c failed"
  echo "== B = $b"; python3 tools/pmc_summary.py gpurun_out/pmc_tail/a$b; python3 tools/pmc_summary.py gpurun_out/pmc_tail/b$b; python3 tools/pmc_summary.py gpurun_out/pmc_tail/c$b
done > gpurun_out/pmc_tail/summary.txt 2>&1
cat gpurun_out/pmc_tail/summary.txt | grep -v "^$" | tail -80
grep -i "icache\|IFETCH" gpurun_out/pmc_tail/counters.txt | head -20
