#!/bin/bash
# rocprofv3 kernel stats of 12 DoRA steps (whisper-tiny, 32 x 2 detectors, pooled) -> gpurun_out/r04_train_kernel_stats.md
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/proft; rm -rf $R/gpurun_out/proft/*
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/proft -o t -- python3 $R/tools/run/train_only.py tiny > $R/gpurun_out/proft.out 2> $R/gpurun_out/proft.err; echo "rc=$?"
tail -1 $R/gpurun_out/proft.out
python3 $R/tools/prof_summary.py $R/gpurun_out/proft $R/gpurun_out/r04_train_kernel_stats.md "round 4: rocprofv3 --kernel-trace --stats -- python3 tools/run/train_only.py tiny (12 DoRA steps, whisper-tiny, 32 x 2 detectors, pooled)" | head -24
