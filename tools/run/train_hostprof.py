"""Host-side profile of the DoRA training step (whisper-tiny, 32 x 2 detectors): cProfile over 20 steps, top cumulative entries.
The GPU idles while the host prepares a step (the loop syncs every step, as the reference's does), so host time in the forward
is step time."""
import cProfile, io, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
dev = torch.device("cuda:0")
bench.dora_step("tiny", 32, dev, 1, steps=4, warmup=2)   # warm everything
pr = cProfile.Profile()
pr.enable()
r = bench.dora_step("tiny", 32, dev, 1, steps=20, warmup=0)
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
print(r["ms"], r["split_ms"])
