import torch, numpy as np, sys
from gw_whisper_amd import ops, synth
M, d = 96000, 384
g = torch.Generator().manual_seed(0)
x = torch.randn(M, d, generator=g).cuda().bfloat16(); dy = (torch.randn(M, 3 * d, generator=g) * 0.3).cuda().bfloat16(); y = torch.randn(M, 3 * d, generator=g).cuda().bfloat16()
c = lambda a: torch.from_numpy(np.asarray(a, np.float32)).cuda()
As, Bs, ms = [], [], []
for p in range(3):
    W0 = (np.random.default_rng(p).standard_normal((d, d)) / np.sqrt(d)).astype(np.float32)
    A, B, m = synth.dora_adapter(d, d, 8, W0, seed=4 + p)
    As.append(c(A)); Bs.append(c(B)); ms.append(c(m))
n = [torch.ones(d).cuda()] * 3; b = [torch.zeros(d).cuda()] * 3
f = lambda: ops.dora_grads_multi(x, dy, y, [0, d, 2 * d], b, [0.125, 1.0, 1.0], [4.0] * 3, As, Bs, ms, n)
f(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): f()
e1.record(); torch.cuda.synchronize()
print(f"dora_grads_multi M={M} d={d} np=3: {e0.elapsed_time(e1)/10*1000:.0f} us per call (incl. 9 memsets)", flush=True)
