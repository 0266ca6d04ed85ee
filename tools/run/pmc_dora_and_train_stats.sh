set -o pipefail
export TMPDIR=/tmp
rm -rf gpurun_out/proft gpurun_out/pmc_dg; mkdir -p gpurun_out/proft gpurun_out/pmc_dg
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/proft -o t -- python3 tools/run/train_only.py tiny > gpurun_out/proft.out 2> gpurun_out/proft.err || exit 1
tail -1 gpurun_out/proft.out
PYTHONPATH=. timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_dg/r -o r -- python3 tools/run/dg2.py > gpurun_out/pmc_dg/r.log 2>&1 || exit 1
PYTHONPATH=. timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_dg/w -o w -- python3 tools/run/dg2.py > gpurun_out/pmc_dg/w.log 2>&1 || exit 1
python3 tools/pmc_summary.py gpurun_out/pmc_dg > gpurun_out/pmc_dg/summary.txt; cat gpurun_out/pmc_dg/summary.txt
