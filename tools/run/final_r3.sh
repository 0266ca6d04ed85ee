# round 3: default bench, the same command under rocprofv3 --kernel-trace --stats, and the two PMC traffic passes
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_r3 gpurun_out/pmc_r3
rm -rf gpurun_out/prof_r3/* gpurun_out/pmc_r3/*
timeout -k 10 700 python bench.py > gpurun_out/r3_bench_final.json 2> gpurun_out/r3_bench_final.err; echo "bench rc=$?"; python tools/show_bench.py gpurun_out/r3_bench_final.json
B="--steps 10 --warmup 3 --no-cpu-baseline --no-train --no-pooled --no-extra"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r3 -o r03 -- python3 bench.py $B > gpurun_out/r3_prof_bench.json 2> gpurun_out/r3_prof.err; echo "prof rc=$?"
python tools/prof_summary.py gpurun_out/prof_r3 gpurun_out/r03_final_kernel_stats.md "round 3: rocprofv3 --kernel-trace --stats -- python3 bench.py $B" > /dev/null && echo stats ok
P="--steps 4 --warmup 1 --no-cpu-baseline --no-train --no-pooled --no-extra --no-breakdown"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_r3/f -o f -- python3 bench.py $P > gpurun_out/pmc_r3/f.json 2> gpurun_out/pmc_r3/f.err; echo "pmc fetch rc=$?"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_r3/w -o w -- python3 bench.py $P > gpurun_out/pmc_r3/w.json 2> gpurun_out/pmc_r3/w.err; echo "pmc write rc=$?"
python tools/pmc_traffic.py gpurun_out/pmc_r3/f gpurun_out/pmc_r3/w gpurun_out/r03_pmc_traffic > /dev/null && echo traffic ok
