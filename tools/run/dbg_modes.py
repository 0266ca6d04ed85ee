"""Debug aid: per-target difference between the hidden and the pooled form of the DoRA step (whisper-tiny, q/k/v/out_proj)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch as T
from gw_whisper_amd import synth
from gw_whisper_amd.encoder import WhisperConfig, WhisperEncoder
from gw_whisper_amd.peft import LoraConfig, get_peft_model
from oracle import logmel as olm
d, L, H, F = synth.ENCODER_SIZES["tiny"]
sd = synth.encoder_state_dict(d, L, H, F, seed=3)
mel = olm.log_mel(synth.strain_segments(2, seed=33))
enc = WhisperEncoder.from_numpy_state_dict(sd, WhisperConfig(d, L, H, F), precision="bf16")
projs = ("q_proj", "k_proj", "v_proj", "out_proj")
targets = [f"layers.{i}.self_attn.{p}" for i in range(L) for p in projs]
peft = get_peft_model(enc, LoraConfig(use_dora=True, r=8, lora_alpha=32, target_modules=targets)).cuda()
with T.no_grad():
    for j, name in enumerate(targets):
        lin = peft.base_model.model.get_submodule(name)
        A, Bm, m = synth.dora_adapter(d, d, 8, sd[name + ".weight"], seed=70 + j)
        lin.lora_A["default"].weight.copy_(T.from_numpy(A)); lin.lora_B["default"].weight.copy_(T.from_numpy(Bm))
        lin.lora_magnitude_vector["default"].weight.copy_(T.from_numpy(m))
wloss = T.from_numpy(np.random.default_rng(0).standard_normal((2, d))).cuda().float()
G = {}
for mode in ("hidden", "last_token"):
    for p in peft.parameters(): p.grad = None
    last = peft.last_token(T.from_numpy(mel).cuda()) if mode == "last_token" else peft(T.from_numpy(mel).cuda()).last_hidden_state[:, -1, :]
    (last * wloss).sum().backward()
    G[mode] = {n: p.grad.double().cpu().numpy().copy() for n, p in peft.named_parameters() if p.grad is not None}
for n in G["hidden"]:
    a, b = G["hidden"][n], G["last_token"][n]
    rel = np.abs(a - b).max() / (np.abs(b).max() + 1e-30)
    if rel > 1e-3: print(f"{n:70s} max|hidden-pooled|/max|pooled| = {rel:.4f}  |pooled|max {np.abs(b).max():.4e}")
print("done")
