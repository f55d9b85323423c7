"""Driver for rocprofv3 --pmc passes over the L = 1024, d = 40 spatial self-attention (BASELINE config 2, 32x32 level)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurons_amd import ops  # noqa: E402

qkv = torch.randn(32, 1024, 960, device="cuda").to(torch.bfloat16)
for _ in range(6):
    out = ops.attention_self(qkv, 8)
torch.cuda.synchronize()
