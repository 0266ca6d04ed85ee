set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_inference.py -q -m gpu -x 2>&1 | grep -v "^$" > gpurun_out/t1.log; tail -25 gpurun_out/t1.log
