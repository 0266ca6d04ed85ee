"""Both bf16 attention kernels on the bench shape (B=256, T=1500, H=6), for rocprofv3 counter passes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gw_whisper_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
qkv = (torch.randn(B, 1500, 1152, device="cuda") * 0.5).bfloat16()
for _ in range(3):
    ops.attention(qkv, 6)
    ops.attention_log2q(qkv, 6)
torch.cuda.synchronize()
print("done")
