set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_qscan.py -q -m gpu -x -s 2>&1 | grep -v "^$" > gpurun_out/t1.log; tail -30 gpurun_out/t1.log
