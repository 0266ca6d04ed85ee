# round 4: default bench, the same command under rocprofv3 --kernel-trace --stats, the two PMC traffic passes, SQ counters of the dominant kernel
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/prof_r4 gpurun_out/pmc_r4
rm -rf gpurun_out/prof_r4/* gpurun_out/pmc_r4/*
timeout -k 10 700 python bench.py > gpurun_out/r4_bench_final.json 2> gpurun_out/r4_bench_final.err; echo "bench rc=$?"; python tools/show_bench.py gpurun_out/r4_bench_final.json
B="--steps 10 --warmup 3 --no-cpu-baseline --no-train --no-pooled --no-extra"
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r4 -o r04 -- python3 $R/bench.py $B > $R/gpurun_out/r4_prof_bench.json 2> $R/gpurun_out/r4_prof.err; echo "prof rc=$?"
python3 $R/tools/prof_summary.py $R/gpurun_out/prof_r4 $R/gpurun_out/r04_final_kernel_stats.md "round 4: rocprofv3 --kernel-trace --stats -- python3 bench.py $B" > /dev/null && echo stats ok
P="--steps 4 --warmup 1 --no-cpu-baseline --no-train --no-pooled --no-extra --no-breakdown"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_r4/f -o f -- python3 $R/bench.py $P > $R/gpurun_out/pmc_r4/f.json 2> $R/gpurun_out/pmc_r4/f.err; echo "pmc fetch rc=$?"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_r4/w -o w -- python3 $R/bench.py $P > $R/gpurun_out/pmc_r4/w.json 2> $R/gpurun_out/pmc_r4/w.err; echo "pmc write rc=$?"
python3 $R/tools/pmc_traffic.py $R/gpurun_out/pmc_r4/f $R/gpurun_out/pmc_r4/w $R/gpurun_out/r04_pmc_traffic > /dev/null && echo traffic ok
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_r4/s -o s -- python3 $R/bench.py $P > $R/gpurun_out/pmc_r4/s.json 2> $R/gpurun_out/pmc_r4/s.err; echo "pmc sq rc=$?"
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_r4/s > $R/gpurun_out/r04_pmc_sq_raw.txt 2>&1 && echo sq ok
