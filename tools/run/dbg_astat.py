import numpy as np, torch as T
from gw_whisper_amd import ops
from oracle import encoder as oenc
def run(M, N, K, with_delta, seed=0):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal((M, K)) * 2 + 0.3).astype(np.float32)
    dl = (rng.standard_normal((M, K)) * 0.5).astype(np.float32)
    dl = T.from_numpy(dl).bfloat16().float().numpy()
    lw = (1 + 0.1 * rng.standard_normal(K)).astype(np.float32)
    lb = (0.1 * rng.standard_normal(K)).astype(np.float32)
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    xn = x + dl if with_delta else x
    ref = oenc.layer_norm(xn.astype(np.float64), lw, lb) @ w.astype(np.float64).T + bias
    wf, u, cb = ops.ln_fold_weights(T.from_numpy(w).cuda(), T.from_numpy(lw).cuda(), T.from_numpy(lb).cuda(), T.from_numpy(bias).cuda())
    for rep in range(2):
        c, x_new = ops.gemm_astat(T.from_numpy(x).cuda(), wf, None, epilogue=0, ln=(u, cb),
                                  delta=T.from_numpy(dl).cuda().bfloat16() if with_delta else None, return_x=True)
        got = c.float().cpu().numpy()
        err = np.abs(got - ref)
        bad = err > 0.05
        rows = np.unique(np.nonzero(bad)[0]); cols = np.unique(np.nonzero(bad)[1])
        print(f"M={M} N={N} K={K} delta={with_delta} rep={rep}: max {err.max():.3f} bad {bad.mean()*100:.1f}%  rows {len(rows)} [{rows[:3]}..{rows[-3:] if len(rows) else ''}] cols {len(cols)} [{cols[:3]}..{cols[-3:] if len(cols) else ''}]")
for M in (256, 1500, 3000):
    for N in (512, 1024, 1536, 2048):
        for dlt in (False, True):
            run(M, N, 512, dlt)
run(3000, 2048, 384, True)
