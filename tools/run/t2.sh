set -o pipefail
timeout -k 10 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -3
timeout -k 10 300 python bench.py --no-cpu-baseline --no-breakdown --steps 3 --warmup 1 > gpurun_out/bench.json 2> gpurun_out/bench.err; python -c "
import json; d=json.load(open('gpurun_out/bench.json')); print(d['dora_step'])"
