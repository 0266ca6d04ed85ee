"""The q/k/v tail in isolation: k_mlp_fused<2, false> (panel prologue + 54-tile q/k/v GEMM) on B segments, a few launches
(for rocprofv3 --pmc passes; B from argv, default 256)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gw_whisper_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
M, d, NQ = B * 1500, 384, 1152
torch.manual_seed(0)
x = torch.randn(M, d, device="cuda") * 2
wq, bq = torch.randn(NQ, d, device="cuda") / d ** 0.5, torch.randn(NQ, device="cuda")
wqf, uq, cq = ops.ln_fold_weights(wq, torch.ones(d, device="cuda"), torch.zeros(d, device="cuda"), bq)
wt = ops.mlp_pack(None, None, wqf) if hasattr(ops, "mlp_pack") else None
for _ in range(4):
    ops.lnqkv_fused(x, wt, uq, cq)
torch.cuda.synchronize()
