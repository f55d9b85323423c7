"""Tiny driver for rocprofv3 --pmc passes: runs a few igemm shapes N times each (see profiles/README.md)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurons_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
which = sys.argv[1] if len(sys.argv) > 1 else "all"


def conv(nimg, H, W, Cin, N):
    x = torch.randn(nimg, H, W, Cin, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, 3, 3, Cin, device=dev) * (9 * Cin) ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device=dev)
    for _ in range(5):
        ops.conv3x3(x, w, b)


def lin(M, N, K):
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device=dev)
    r = torch.randn(M, N, device=dev).to(torch.bfloat16)
    for _ in range(5):
        ops.gemm(a, w, b, r)


conv(32, 32, 32, 960, 320)      # M=32768 N=320 K=8640  (best case, ~890 TF/s)
lin(8192, 1920, 640)            # mid 1x1
lin(32768, 320, 320)            # memory-bound 1x1
lin(2048, 1280, 1280)           # small grid
torch.cuda.synchronize()
