"""Q-scan front end only (512 x 2048 samples), for a rocprofv3 kernel trace."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gw_whisper_amd import synth
from gw_whisper_amd.qscan import QScan
qs = QScan(duration=1.0, sample_rate=2048, spectrogram_shape=[128, 128], qrange=[4, 128])
x = torch.from_numpy(synth.strain_segments(512, seed=3, n_samples=2048)).cuda()
for _ in range(6):
    qs(x)
torch.cuda.synchronize()
print("done")
