set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -q -m gpu -x > gpurun_out/t1.log 2>&1; rc=$?; tail -5 gpurun_out/t1.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; rc=$?; tail -4 gpurun_out/smoke.log; [ $rc -eq 0 ] || exit $rc
bash tools/run/final.sh
