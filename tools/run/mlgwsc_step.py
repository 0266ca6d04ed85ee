"""One MLGWSC-style training step (MLGWSC-1/train.py:494-504: Adam over the Q-transform adapter + the DoRA adapters + the
head, through the frozen whisper-tiny encoder) repeated a few times, for rocprofv3 --kernel-trace --stats: the trace must
show no MIOpen / aten convolution kernel.  usage: mlgwsc_step.py [batch] [steps]"""
import fnmatch, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gw_whisper_amd import synth
from gw_whisper_amd.encoder import WhisperConfig, WhisperEncoder
from gw_whisper_amd.inference import GWWhisperClassifier, RegBCELoss
from gw_whisper_amd.peft import LoraConfig, get_peft_model
from gw_whisper_amd.qscan import QTransformAdapter
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
torch.manual_seed(0)
dev = torch.device("cuda")
enc = WhisperEncoder.from_numpy_state_dict(synth.named_encoder_state_dict("tiny", seed=0), WhisperConfig.named("tiny"), precision="bf16")
pats = ["layers.*.self_attn.q_proj", "layers.*.self_attn.k_proj", "layers.*.self_attn.v_proj", "layers.*.self_attn.out_proj"]
targets = [n for n, _ in enc.named_modules() if any(fnmatch.fnmatch(n, p) for p in pats)]
peft = get_peft_model(enc, LoraConfig(use_dora=True, r=8, lora_alpha=32, target_modules=targets))
for name, p in peft.named_parameters():
    p.requires_grad = "lora" in name
model = GWWhisperClassifier(peft, n_detectors=2, num_classes=2, adapter=QTransformAdapter.train_variant(n_detectors=2)).to(dev)
params = [p for p in model.parameters() if p.requires_grad]
opt = torch.optim.Adam(params, lr=1e-4)
crit = RegBCELoss(dim=2)
x = torch.from_numpy(synth.strain_segments(2 * B, seed=3, n_samples=2048)).to(dev).reshape(B, 2, 2048)
y = torch.nn.functional.one_hot(torch.arange(B, device=dev) % 2, 2).float()
t = []
for it in range(steps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    loss = crit(model(x), y)
    loss.backward()
    opt.step()
    torch.cuda.synchronize(); t.append((time.perf_counter() - t0) * 1e3)
print({"batch": B, "trainable": sum(p.numel() for p in params), "loss": float(loss), "ms_per_step": sorted(t)[len(t) // 2], "all_ms": [round(v, 1) for v in t]})
