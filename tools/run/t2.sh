set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_training.py -q -m gpu -x -s 2>&1 | grep -v "^$" > gpurun_out/t1.log; grep -n "FAILED\|passed\|failed\|d loss\|rror" gpurun_out/t1.log | head -20
