"""Time the bf16 attention kernel on the bench shape (B = 256, T = 1500, H = 6): median of 7 x 5 launches."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gw_whisper_amd import ops
torch.manual_seed(0)
qkv = (torch.randn(256, 1500, 1152, device="cuda") * 0.5).bfloat16()
ops.attention_log2q(qkv, 6); ts = []
for _ in range(7):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ops.attention_log2q(qkv, 6)
    e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) / 5)
print("attention %.4f ms (median of 7 x 5)" % statistics.median(ts))
