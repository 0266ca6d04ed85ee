set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out/proft; rm -rf gpurun_out/proft/*
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/proft -o t -- python3 tools/run/train_only.py tiny > gpurun_out/proft.out 2> gpurun_out/proft.err; echo "rc=$?"
tail -1 gpurun_out/proft.out
python3 tools/prof_summary.py gpurun_out/proft gpurun_out/r03_train_kernel_stats.md "round 3: rocprofv3 --kernel-trace --stats -- python3 tools/run/train_only.py tiny (12 DoRA steps, whisper-tiny, 32 x 2 detectors, pooled)" | head -32
