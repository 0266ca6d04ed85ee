import os, sys, subprocess, numpy as np
if len(sys.argv) > 1:
    import torch as T
    from gw_whisper_amd import synth
    from gw_whisper_amd.encoder import WhisperConfig, WhisperEncoder
    from oracle import logmel as olm
    d, L, H, F = synth.ENCODER_SIZES["base"]
    sd = synth.encoder_state_dict(d, L, H, F, seed=5)
    mel = T.from_numpy(olm.log_mel(synth.strain_segments(1, seed=9))).cuda()
    enc = WhisperEncoder.from_numpy_state_dict(sd, WhisperConfig.named("base"), precision="fp32").cuda()
    with T.no_grad():
        ref = enc(mel).last_hidden_state
        enc.precision = "bf16"
        out = enc(mel).last_hidden_state
    print("mask", os.environ.get("GWW_GENERIC_PATH"), "max", (out - ref).abs().max().item(), "mean", (out - ref).abs().mean().item())
else:
    for m in ("0", "1", "2", "4", "7"):
        subprocess.run([sys.executable, __file__, "x"], env=dict(os.environ, GWW_GENERIC_PATH=m))
