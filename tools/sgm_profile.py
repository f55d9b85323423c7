"""Per-launch timings of one sgm UNetModel forward at BASELINE config 3 shapes (CFG batch 2, 64x64 latent).
Usage (GPU box): python tools/sgm_profile.py gpurun_out/ops_sgm.csv"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import gpu_random_state_dict  # noqa: E402
from neurons_amd.sgm import NativeSGMUNet, SGMUNetConfig, sgm_state_dict_schema  # noqa: E402

dev = torch.device("cuda", 0)
cfg = SGMUNetConfig()
net = NativeSGMUNet(cfg).to(dev)
net.load_state_dict({k: v.cpu() for k, v in gpu_random_state_dict(sgm_state_dict_schema(cfg), 3, dev).items()})
x = torch.randn(2, 4, 64, 64, device=dev)
ctx = torch.randn(2, 256, 1664, device=dev)
y = torch.randn(2, 1024, device=dev)
for _ in range(2):
    net(x, torch.tensor([500.0, 500.0]), context=ctx, y=y)
torch.cuda.synchronize()
os.environ["NR_PROFILE_CSV"] = sys.argv[1]
for _ in range(2):
    p = net.profile_last()
print({k: (round(v["ms"], 3), v["launches"]) for k, v in p.items()})
