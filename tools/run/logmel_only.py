"""log-mel front end only (256 x 1 s segments), for a rocprofv3 kernel trace."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gw_whisper_amd import ops, synth
w = torch.from_numpy(synth.strain_segments(256, seed=1)).cuda()
for _ in range(10):
    ops.logmel(w)
torch.cuda.synchronize()
print("done")
