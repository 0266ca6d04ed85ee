"""Diagnostic: per-tensor accuracy of the DoRA gradients of the bf16 HIP backward against finite differences of the
fp32 HIP forward (GWW_PREC_F32, 1e-7 against HF), whisper-tiny, direction = the tensor's own gradient."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch as T
from gw_whisper_amd import synth
from gw_whisper_amd.encoder import WhisperConfig, WhisperEncoder
from gw_whisper_amd.peft import LoraConfig, get_peft_model
from oracle import logmel as olm
name = sys.argv[1] if len(sys.argv) > 1 else "tiny"
projs = ("q_proj", "k_proj", "v_proj", "out_proj") if len(sys.argv) > 2 and sys.argv[2] == "qkvo" else ("q_proj", "k_proj", "v_proj")
d, L, H, F = synth.ENCODER_SIZES[name]
sd = synth.encoder_state_dict(d, L, H, F, seed=3)
mel = T.from_numpy(olm.log_mel(synth.strain_segments(2, seed=33))).cuda()
enc = WhisperEncoder.from_numpy_state_dict(sd, WhisperConfig(d, L, H, F), precision="bf16")
targets = [f"layers.{i}.self_attn.{p}" for i in range(L) for p in projs]
peft = get_peft_model(enc, LoraConfig(use_dora=True, r=8, lora_alpha=32, target_modules=targets)).cuda()
with T.no_grad():
    for j, n in enumerate(targets):
        lin = peft.base_model.model.get_submodule(n)
        A, Bm, m = synth.dora_adapter(d, d, 8, sd[n + ".weight"], seed=70 + j)
        lin.lora_A["default"].weight.copy_(T.from_numpy(A)); lin.lora_B["default"].weight.copy_(T.from_numpy(Bm))
        lin.lora_magnitude_vector["default"].weight.copy_(T.from_numpy(m))
wl = T.from_numpy(np.random.default_rng(0).standard_normal((2, d))).cuda().float()
mode = os.environ.get("MODE", "last_token")
last = peft.last_token(mel) if mode == "last_token" else peft(mel).last_hidden_state[:, -1, :]
(last * wl).sum().backward()
grads = {n: p.grad.clone() for n, p in peft.named_parameters() if p.grad is not None}

def loss32():
    # fp32 forward with the weight norm FROZEN at its current value (peft detaches it): emulate by computing the loss with
    # the merged weights the library builds; the norm moves with A / B, so the FD below is taken with the norm
    # recomputed -- its derivative is what peft detaches.  To freeze it we rescale m by n_new / n_old.
    enc.precision = "fp32"
    with T.no_grad():
        out = peft.last_token(mel)
    enc.precision = "bf16"
    return float((out.double() * wl.double()).sum())

def norms():
    return {n: T.linalg.norm(peft.base_model.model.get_submodule(n).base_layer.weight.double() + 4.0 *
                             peft.base_model.model.get_submodule(n).lora_B["default"].weight.double() @
                             peft.base_model.model.get_submodule(n).lora_A["default"].weight.double(), dim=1) for n in targets}
n0 = norms()
params = dict(peft.named_parameters())
eps = float(os.environ.get("EPS", "2e-3"))
print(f"{name} {'+'.join(projs)} mode={mode} eps={eps}")
for pn, g in grads.items():
    tgt = pn.split(".lora_")[0].replace("base_model.model.", "")
    p = params[pn]
    mag = params[pn.split(".lora_")[0] + ".lora_magnitude_vector.default.weight"]
    v = g / (g.pow(2).mean().sqrt() + 1e-30)
    vals = []
    for sgn in (+1, -1):
        with T.no_grad():
            p.add_(sgn * eps * v)
            keep = mag.detach().clone()
            if "magnitude" not in pn:      # detached norm: y = (m / n0) W' -> keep m / n fixed by scaling m with n_new / n0
                mag.mul_((norms()[tgt] / n0[tgt]).float())
            vals.append(loss32())
            mag.copy_(keep)
            p.sub_(sgn * eps * v)
    fd = (vals[0] - vals[1]) / (2 * eps)
    an = float((g.double() * v.double()).sum())
    print(f"{pn.replace('base_model.model.', ''):62s} an {an:12.4f} fd {fd:12.4f} rel {abs(an - fd) / (abs(fd) + 1e-12):7.4f}  |g|max {g.abs().max().item():.3e}")

# ---- optional: element-wise finite differences of ONE tensor (FULL=<substring of its name>)
full = os.environ.get("FULL")
if full:
    pn = [n for n in grads if full in n][0]
    tgt = pn.split(".lora_")[0].replace("base_model.model.", "")
    p, g = params[pn], grads[pn]
    mag = params[pn.split(".lora_")[0] + ".lora_magnitude_vector.default.weight"]
    fdg = T.zeros_like(g, dtype=T.float64)
    flat = p.data.view(-1)
    e2 = float(os.environ.get("EPS2", "1e-2"))
    for i in range(flat.numel()):
        vals = []
        for sgn in (+1, -1):
            with T.no_grad():
                flat[i] += sgn * e2
                keep = mag.detach().clone()
                if "magnitude" not in pn:
                    mag.mul_((norms()[tgt] / n0[tgt]).float())
                vals.append(loss32())
                mag.copy_(keep)
                flat[i] -= sgn * e2
        fdg.view(-1)[i] = (vals[0] - vals[1]) / (2 * e2)
    gd = g.double()
    cos = float((gd * fdg).sum() / (gd.norm() * fdg.norm()))
    scale = float((gd * fdg).sum() / (fdg * fdg).sum())
    print(f"FULL {pn}: cosine {cos:.5f}  least-squares scale g_hip / g_fd {scale:.4f}  |g_hip - g_fd| / |g_fd| {float((gd - fdg).norm() / fdg.norm()):.4f}")
    r = (gd - fdg)
    if r.dim() == 2:
        print("  relative residual per row   :", [round(float(r[i].norm() / fdg[i].norm()), 3) for i in range(min(r.shape[0], 8))])
        cols = r.norm(dim=0) / (fdg.norm(dim=0) + 1e-30)
        print("  relative residual per column: min %.3f median %.3f max %.3f" % (float(cols.min()), float(cols.median()), float(cols.max())))
        print("  g_hip[0,:6]", gd[0, :6].tolist()); print("  g_fd [0,:6]", fdg[0, :6].tolist())
