#!/bin/bash
# round 3: whole -m gpu suite, smoke(), then the default bench (outputs under gpurun_out/)
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/r3_gputests.txt 2>&1; rc=$?
tail -5 gpurun_out/r3_gputests.txt
[ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3_smoke.txt 2>&1 || { tail -5 gpurun_out/r3_smoke.txt; exit 1; }
tail -2 gpurun_out/r3_smoke.txt
python bench.py > gpurun_out/r3_bench.json 2> gpurun_out/r3_bench.err || { tail -5 gpurun_out/r3_bench.err; exit 1; }
python tools/show_bench.py gpurun_out/r3_bench.json 2>/dev/null | head -60
