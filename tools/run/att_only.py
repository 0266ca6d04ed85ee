import torch
from gw_whisper_amd import ops
a = (torch.randn(256, 1500, 1152, device="cuda") * 0.5).bfloat16()
for _ in range(3): ops.attention(a, 6)
torch.cuda.synchronize()
