set -o pipefail
timeout -k 10 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
timeout -k 10 600 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"; python tools/show_bench.py gpurun_out/bench.json
python -c "
import json; d=json.load(open('gpurun_out/bench.json')); print(d['roofline']); print(d['config'])"
