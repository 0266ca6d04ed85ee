# kernel traces: whisper-base forward, whisper-small forward, whisper-small DoRA step (-> gpurun_out/prof_*.txt)
set -o pipefail
export TMPDIR=/tmp
for job in "fwd_only.py base" "fwd_only.py small" "train_only.py small"; do
  tag=$(echo $job | tr ' .' '__')
  rm -rf gpurun_out/prof_$tag; mkdir -p gpurun_out/prof_$tag
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o t -- python3 tools/run/$job > gpurun_out/prof_$tag.out 2> gpurun_out/prof_$tag.err || { echo "FAILED $job"; tail -3 gpurun_out/prof_$tag.err; exit 1; }
  python3 - "$tag" <<'PY'
import csv, glob, sys
tag = sys.argv[1]
f = glob.glob(f'gpurun_out/prof_{tag}/**/t_kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
with open(f'gpurun_out/prof_{tag}_stats.txt', 'w') as o:
    for r in rows[:22]:
        o.write(f"{r['Name'][:80]:80s} calls {r['Calls']:>5s} total_us {float(r['TotalDurationNs'])/1e3:11.1f} avg_us {float(r['AverageNs'])/1e3:9.1f} pct {r['Percentage']}\n")
print("====", tag); print(open(f'gpurun_out/prof_{tag}_stats.txt').read())
PY
done
