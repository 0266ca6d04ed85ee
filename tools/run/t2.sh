set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_training.py -q -m gpu -x -k "harness" 2>&1 | tail -15
