"""k_mlp_fused<1, true> in isolation (out_proj + MLP + next LN1 / q,k,v) on B segments, a few launches with preallocated
operands (for rocprofv3 --pmc passes; B from argv, default 256)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gw_whisper_amd import ops
from gw_whisper_amd._lib import lib, check
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
M, d, F, NQ = B * 1500, 384, 1536, 1152
torch.manual_seed(0)
g = lambda *s: torch.randn(*s, device="cuda")
x, ctx = g(M, d) * 2, g(M, d).bfloat16()
wo, bo = (g(d, d) / d ** 0.5).bfloat16(), g(d)
w1, b1, w2, b2 = g(F, d) / d ** 0.5, g(F), (g(d, F) / F ** 0.5).bfloat16(), g(d)
wq, bq = g(NQ, d) / d ** 0.5, g(NQ)
ones, zeros = torch.ones(d, device="cuda"), torch.zeros(d, device="cuda")
w1f, u, cb = ops.ln_fold_weights(w1, ones, zeros, b1)
wqf, uq, cq = ops.ln_fold_weights(wq, ones, zeros, bq)
st = torch.cuda.current_stream().cuda_stream
wt = torch.empty((d * d + 2 * d * F + NQ * d,), dtype=torch.bfloat16, device="cuda")
check(lib().gww_mlp_pack_op_bf16(wo.data_ptr(), w1f.data_ptr(), w2.data_ptr(), wqf.data_ptr(), wt.data_ptr(), d, F, NQ, st))
x_out = torch.empty_like(x)
o = torch.empty(((M + 127) // 128 * 128, NQ), dtype=torch.bfloat16, device="cuda")
for _ in range(4):
    check(lib().gww_attn_out_mlp_fused_bf16(x.data_ptr(), ctx.data_ptr(), bo.data_ptr(), x_out.data_ptr(), u.data_ptr(), cb.data_ptr(),
                                            wt.data_ptr(), b2.data_ptr(), None, M, d, F, uq.data_ptr(), cq.data_ptr(), o.data_ptr(), NQ, st))
torch.cuda.synchronize()
