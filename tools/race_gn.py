"""Stress: small-image GroupNorm on one stream while other kernels run on a second stream; the GN result must not change."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurons_amd import ops  # noqa: E402

dev = "cuda"
torch.manual_seed(0)
x = (torch.randn(16, 8, 8, 64, device=dev) * 2 + 0.5).to(torch.bfloat16)
g, b = torch.randn(64, device=dev), torch.randn(64, device=dev)
x2 = (torch.randn(16, 8, 8, 64, device=dev)).to(torch.bfloat16)
a = torch.randn(1024, 64, device=dev).to(torch.bfloat16)
w = (torch.randn(64, 64, device=dev) * 0.1).to(torch.bfloat16)
xc = torch.randn(16, 8, 8, 64, device=dev).to(torch.bfloat16)
wc = (torch.randn(64, 3, 3, 64, device=dev) * 0.05).to(torch.bfloat16)
ref = ops.groupnorm(x, g, b, groups=32, eps=1e-5, silu=True)
torch.cuda.synchronize()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for other in ("none", "gn", "gemm", "conv", "attn"):
    bad = 0
    for it in range(300):
        with torch.cuda.stream(s2):
            for _ in range(4):
                if other == "gn":
                    ops.groupnorm(x2, g, b, groups=32, eps=1e-5, silu=True)
                elif other == "gemm":
                    ops.gemm(a, w)
                elif other == "conv":
                    ops.conv3x3(xc, wc)
                elif other == "attn":
                    ops.attention_self(torch.randn(16, 64, 192, device=dev).to(torch.bfloat16), 8)
        with torch.cuda.stream(s1):
            outs = [ops.groupnorm(x, g, b, groups=32, eps=1e-5, silu=True) for _ in range(4)]
        torch.cuda.synchronize()
        bad += sum(0 if torch.equal(o, ref) else 1 for o in outs)
    print(f"concurrent with {other:5s}: {bad} / 1200 GroupNorm results differ", flush=True)
