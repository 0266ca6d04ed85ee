set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc_r gpurun_out/pmc_w
rm -rf gpurun_out/pmc_r/* gpurun_out/pmc_w/*
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_r -o r -- python3 bench.py --split 0 --steps 2 --warmup 1 --no-cpu-baseline --no-train --no-breakdown > gpurun_out/pmc_r.json 2> gpurun_out/pmc_r.err; echo "rc=$?"
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -o w -- python3 bench.py --split 0 --steps 2 --warmup 1 --no-cpu-baseline --no-train --no-breakdown > gpurun_out/pmc_w.json 2> gpurun_out/pmc_w.err; echo "rc=$?"
