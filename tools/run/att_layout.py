import torch, time
from gw_whisper_amd import ops
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
a = (torch.randn(256, 1500, 1152, device="cuda") * 0.5).bfloat16()
b = (torch.randn(1536, 1500, 192, device="cuda") * 0.5).bfloat16()
print("row-major 6 heads interleaved (stride 2304 B):", t(lambda: ops.attention(a, 6)), "ms")
print("one head per batch item        (stride  384 B):", t(lambda: ops.attention(b, 1)), "ms")
