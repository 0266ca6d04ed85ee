set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "mlp_" 2>&1 | tail -12
