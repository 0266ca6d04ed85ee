set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out/prof gpurun_out/prof0 gpurun_out/pmc_r gpurun_out/pmc_w
rm -rf gpurun_out/prof/* gpurun_out/prof0/* gpurun_out/pmc_r/* gpurun_out/pmc_w/*
timeout -k 10 600 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"; python tools/show_bench.py gpurun_out/bench.json
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r01d -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train > gpurun_out/prof_bench.json 2> gpurun_out/prof.err; echo "prof rc=$?"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof0 -o r01e -- python3 bench.py --split 0 --steps 10 --warmup 3 --no-cpu-baseline --no-train > gpurun_out/prof0_bench.json 2> gpurun_out/prof0.err; echo "prof0 rc=$?"
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_r -o r -- python3 bench.py --split 0 --steps 2 --warmup 1 --no-cpu-baseline --no-train --no-breakdown > gpurun_out/pmc_r.json 2> gpurun_out/pmc_r.err; echo "rc=$?"
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -o w -- python3 bench.py --split 0 --steps 2 --warmup 1 --no-cpu-baseline --no-train --no-breakdown > gpurun_out/pmc_w.json 2> gpurun_out/pmc_w.err; echo "rc=$?"
