set -o pipefail
timeout -k 10 300 python -m pytest tests/test_gpu_training.py -q -m gpu -x -k "dora_parameter" > gpurun_out/t2_test.log 2>&1 ; rc=$?; tail -3 gpurun_out/t2_test.log; [ $rc -eq 0 ] || exit $rc
for nb in 256 128; do echo "blocks=$nb"; GWW_DORA_BLOCKS=$nb PYTHONPATH=. timeout -k 10 120 python tools/run/dg2.py || exit 1; done
