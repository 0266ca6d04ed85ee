set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_training.py tests/test_gpu_encoder.py -q -m gpu -x -k "attention or training_step or config1 or finite" 2>&1 | tail -4
for w in 4 8; do echo "waves $w: $(GWW_ATT_WAVES=$w timeout -k 10 300 python tools/bench_kernels.py 2>&1 | grep attention)"; done
timeout -k 10 600 python tools/stamp_att.py 2>&1 | tail -8
