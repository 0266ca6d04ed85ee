set -o pipefail
mkdir -p gpurun_out/prof gpurun_out/prof0
export TMPDIR=/tmp
timeout -k 10 600 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"; python tools/show_bench.py gpurun_out/bench.json
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof0 -o r01c -- python3 bench.py --split 0 --steps 10 --warmup 3 --no-cpu-baseline --no-train > gpurun_out/prof0_bench.json 2> gpurun_out/prof0.err; echo "prof0 rc=$?"
