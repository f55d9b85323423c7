"""Accuracy of the LayerNorm-folded wide projection kernels on heavy-tailed rows: register-panel form of lin160.hip vs the tiled igemm (NR_LIN160_PANEL=0 in a
second process) against an fp64 reference of LayerNorm -> Linear (-> GEGLU).  Usage: python tools/panel_numerics.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurons_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
for name, mean, outl in (("plain", 0.2, 0.0), ("offset rows", 3.0, 0.0), ("outlier channels", 0.2, 40.0), ("offset + outliers", 3.0, 40.0)):
    for M, K, N, geglu in ((2048, 640, 5120, True), (2048, 640, 1920, False)):
        g = torch.Generator(device=dev).manual_seed(1)
        a = torch.randn(M, K, generator=g, device=dev) * 1.3 + mean
        if outl:
            idx = torch.randperm(K, generator=g, device=dev)[:6]
            a[:, idx] *= outl
        a = a.to(torch.bfloat16)
        w = torch.randn(N, K, generator=g, device=dev) * K ** -0.5
        bias = 0.1 * torch.randn(N, generator=g, device=dev)
        gamma = 1.0 + 0.2 * torch.randn(K, generator=g, device=dev)
        beta = 0.1 * torch.randn(K, generator=g, device=dev)
        h = torch.nn.functional.linear(torch.nn.functional.layer_norm(a.double(), (K,), gamma.double(), beta.double(), 1e-5), w.double(), bias.double())
        ref = h[:, :N // 2] * torch.nn.functional.gelu(h[:, N // 2:]) if geglu else h
        out = ops.ln_gemm(a, w, gamma, beta, bias=bias, geglu=geglu).double()
        rel = ((out - ref).norm() / ref.norm()).item()
        # the un-folded path for scale: LayerNorm in fp32, rounded to bf16, then the plain GEMM
        n16 = torch.nn.functional.layer_norm(a.float(), (K,), gamma, beta, 1e-5).to(torch.bfloat16)
        h2 = torch.nn.functional.linear(n16.double(), w.to(torch.bfloat16).double(), bias.double())
        ref2 = h2[:, :N // 2] * torch.nn.functional.gelu(h2[:, N // 2:]) if geglu else h2
        rel2 = ((ref2.to(torch.bfloat16).double() - ref).norm() / ref.norm()).item()
        print(f"{name:18s} M={M} N={N} geglu={int(geglu)} PANEL={os.environ.get('NR_LIN160_PANEL', '1')}: rel-L2 vs fp64 {rel:.3e}   (un-folded bf16 composition: {rel2:.3e})")
