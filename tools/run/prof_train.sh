set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out/proft; rm -rf gpurun_out/proft/*
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/proft -o t -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-breakdown > gpurun_out/proft_bench.json 2> gpurun_out/proft.err; echo "rc=$?"
