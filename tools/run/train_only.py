"""DoRA training steps only (whisper-tiny, 32 x 2 detectors), for a rocprofv3 kernel trace."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
dev = torch.device("cuda:0")
import sys as _s
name = _s.argv[1] if len(_s.argv) > 1 else "tiny"
r = bench.dora_step(name, 32, dev, 1, steps=4 if name != "tiny" else 10, warmup=2)
print(r)
