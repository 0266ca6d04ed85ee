set -o pipefail
timeout -k 10 300 python bench.py --no-cpu-baseline --no-train > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"; python tools/show_bench.py gpurun_out/bench.json
