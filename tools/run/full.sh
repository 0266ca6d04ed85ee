set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -q -m gpu -x 2>&1 | grep -v "^$" > gpurun_out/t1.log; grep -n "FAILED\|passed\|failed\|rror" gpurun_out/t1.log | head -20
timeout -k 10 600 python bench.py --no-cpu-baseline --no-train > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"
python tools/show_bench.py gpurun_out/bench.json
GWW_GENERIC_PATH=16 timeout -k 10 600 python bench.py --no-cpu-baseline --no-train > gpurun_out/bench_nofuse.json 2> gpurun_out/bench_nofuse.err; python tools/show_bench.py gpurun_out/bench_nofuse.json
