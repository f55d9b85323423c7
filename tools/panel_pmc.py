"""Driver for rocprofv3 --pmc passes over the register-panel kernel of lin160.hip: four launches of each headline shape (GEGLU / q|k|v at C = 640 and 1280)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurons_amd import ops  # noqa: E402

dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
for M, K, N, geglu in ((8192, 640, 5120, True), (8192, 640, 1920, False), (2048, 1280, 10240, True), (2048, 1280, 3840, False)):
    a = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
    w = torch.randn(N, K, generator=g, device=dev) * K ** -0.5
    gamma, beta, bias = torch.ones(K, device=dev), torch.zeros(K, device=dev), torch.zeros(N, device=dev)
    for _ in range(4):
        ops.ln_gemm(a, w, gamma, beta, bias=bias, geglu=geglu)
torch.cuda.synchronize()
