set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_encoder.py -q -m gpu -x -k "mlp_ or config1 or named_sizes" 2>&1 | tail -4
timeout -k 10 300 python bench.py --no-cpu-baseline --no-train > gpurun_out/bench.json 2> gpurun_out/bench.err; python tools/show_bench.py gpurun_out/bench.json
