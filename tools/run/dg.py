import torch, numpy as np, sys
from gw_whisper_amd import ops, synth
M, d = 96000, 384
g = torch.Generator().manual_seed(0)
x = torch.randn(M, d, generator=g).cuda().bfloat16(); dy = (torch.randn(M, d, generator=g) * 0.3).cuda().bfloat16(); y = torch.randn(M, d, generator=g).cuda().bfloat16()
W0 = (np.random.default_rng(0).standard_normal((d, d)) / np.sqrt(d)).astype(np.float32)
A, B, m = synth.dora_adapter(d, d, 8, W0, seed=4)
c = lambda a: torch.from_numpy(np.asarray(a, np.float32)).cuda()
A, B, m = c(A), c(B), c(m); n = torch.ones(d).cuda(); b = torch.zeros(d).cuda()
f = lambda: ops.dora_grads(x, dy, y, b, 1.0, 4.0, A, B, m, n)
f(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): f()
e1.record(); torch.cuda.synchronize()
print(f"dora_grads M={M} d={d}: {e0.elapsed_time(e1)/10*1000:.0f} us per call (incl. 3 memsets)")
