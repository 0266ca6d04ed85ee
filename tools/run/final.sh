set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
rm -rf gpurun_out/prof/*
timeout -k 10 600 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"; python tools/show_bench.py gpurun_out/bench.json
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r01d -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train --no-pooled > gpurun_out/prof_bench.json 2> gpurun_out/prof.err; echo "prof rc=$?"
python tools/show_bench.py gpurun_out/prof_bench.json
