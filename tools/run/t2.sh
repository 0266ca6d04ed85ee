set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "mlp_" 2>&1 | grep -v "^$" > gpurun_out/t1.log; grep -n "FAILED\|passed\|failed\|Mismatch\|Max abs\|rror" gpurun_out/t1.log | head -40
timeout -k 10 300 python tools/bench_kernels.py 2>&1 | grep mlp_fused
timeout -k 10 300 python tools/stamp_mlp.py 2>&1 | tail -17
