"""Encoder forward only (whisper-base / -small at B = 64), for a rocprofv3 kernel trace."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gw_whisper_amd import synth
from gw_whisper_amd.encoder import WhisperConfig, WhisperEncoder
name = sys.argv[1] if len(sys.argv) > 1 else "base"
dev = torch.device("cuda:0")
enc = WhisperEncoder.from_numpy_state_dict(synth.named_encoder_state_dict(name, seed=0), WhisperConfig.named(name), precision="bf16").to(dev)
mel = torch.randn(64, 80, 3000, device=dev)
with torch.no_grad():
    for _ in range(6):
        enc.forward_raw(mel, want_hidden=True, want_last=True)
torch.cuda.synchronize()
