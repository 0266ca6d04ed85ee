set -o pipefail
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_encoder.py -q -m gpu -x -k "logmel or config1 or golden" > gpurun_out/lm_test.log 2>&1 ; rc=$?; tail -2 gpurun_out/lm_test.log; [ $rc -eq 0 ] || exit $rc
for v in 0 1; do
if [ $v = 1 ]; then export GWW_LOGMEL_VALU=1; fi
PYTHONPATH=. timeout -k 10 120 python - <<'PY'
import torch, os
from gw_whisper_amd import ops, synth
w = torch.from_numpy(synth.strain_segments(256, seed=1)).cuda()
ops.logmel(w); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): ops.logmel(w)
e1.record(); torch.cuda.synchronize()
print(f"logmel 256 segments ({'VALU' if os.environ.get('GWW_LOGMEL_VALU') else 'MFMA'} kernel): {e0.elapsed_time(e1)/20*1000:.0f} us")
PY
done
