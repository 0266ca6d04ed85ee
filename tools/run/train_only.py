"""DoRA training steps only (whisper-tiny, 32 x 2 detectors), for a rocprofv3 kernel trace."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
dev = torch.device("cuda:0")
r = bench.dora_step("tiny", 32, dev, 1, steps=10, warmup=2)
print(r)
