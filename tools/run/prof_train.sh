set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out/proft; rm -rf gpurun_out/proft/*
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/proft -o t -- python3 tools/run/train_only.py small > gpurun_out/proft.out 2> gpurun_out/proft.err; echo "rc=$?"
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/proft/**/t_kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
with open('gpurun_out/proft_stats.txt', 'w') as o:
    for r in rows[:40]:
        o.write(f"{r['Name'][:90]:90s} calls {r['Calls']:>6s} total_us {float(r['TotalDurationNs'])/1e3:12.1f} avg_us {float(r['AverageNs'])/1e3:10.1f} pct {r['Percentage']}\n")
print(open('gpurun_out/proft_stats.txt').read())
PY
