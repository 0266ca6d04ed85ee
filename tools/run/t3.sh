set -o pipefail

timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_encoder.py tests/test_gpu_training.py -q -m gpu -x -k "attention or encoder or training_step" > gpurun_out/t3_test.log 2>&1 ; rc=$?; tail -3 gpurun_out/t3_test.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --no-cpu-baseline --no-train --steps 10 --warmup 3 > gpurun_out/bench3.json 2> gpurun_out/bench3.err && python tools/show_bench.py gpurun_out/bench3.json
