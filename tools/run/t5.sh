set -o pipefail
export GWW_ATT_PIPE=1
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "attention" > gpurun_out/t5_test.log 2>&1 ; rc=$?; tail -3 gpurun_out/t5_test.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --no-cpu-baseline --no-train --no-pooled --steps 10 --warmup 3 > gpurun_out/bench5.json 2> gpurun_out/bench5.err && python tools/show_bench.py gpurun_out/bench5.json
