#!/bin/bash
# rocprofv3 kernel stats of MLGWSC-style training steps (Q-adapter + DoRA + head through whisper-tiny): is any library convolution left?
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/profm; rm -rf $R/gpurun_out/profm/*
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/profm -o t -- python3 $R/tools/run/mlgwsc_step.py 32 6 > $R/gpurun_out/profm.out 2> $R/gpurun_out/profm.err; echo "rc=$?"
tail -1 $R/gpurun_out/profm.out
python3 $R/tools/prof_summary.py $R/gpurun_out/profm $R/gpurun_out/r04_mlgwsc_step_kernel_stats.md "round 4: rocprofv3 --kernel-trace --stats -- python3 tools/run/mlgwsc_step.py 32 6 (MLGWSC-style step: Q-transform adapter (128 x 128, CNN 32/64/128) + DoRA q,k,v,out_proj + head through whisper-tiny, 32 two-detector windows)" | head -40
echo "library convolution kernels in the trace:"; grep -i -E "miopen|conv|Cijk|gemm" $R/gpurun_out/profm/t_kernel_stats.csv | cut -c1-140 | head -20
