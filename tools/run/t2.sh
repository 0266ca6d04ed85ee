set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_training.py -q -m gpu -k "attention or training_step" 2>&1 | grep -v "^$" > gpurun_out/t1.log; grep -n "FAILED\|passed\|failed" gpurun_out/t1.log | head
timeout -k 10 300 python tools/bench_kernels.py 2>&1 | grep attention
timeout -k 10 300 python bench.py --no-cpu-baseline --no-train > gpurun_out/bench.json 2> gpurun_out/bench.err; python tools/show_bench.py gpurun_out/bench.json
