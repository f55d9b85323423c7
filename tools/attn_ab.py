"""Timing + accuracy of the block-shared self-attention kernel (attention.hip) at the d = 40 shapes of BASELINE configs 2 / 5, for an A/B of
the softmax denominator: default = pad column of V holds 1.0 and the matrix pipe accumulates the row sum; NR_ATTN_ROWSUM=adds = fp32
v_add_f32 chain (the round-3 form).  The env switch is read once per process: run the script once per arm.
Usage (GPU box): python tools/attn_ab.py ; NR_ATTN_ROWSUM=adds python tools/attn_ab.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurons_amd import ops  # noqa: E402

arm = "adds" if os.environ.get("NR_ATTN_ROWSUM", "")[:1] == "a" else "ones-column"
for nimg, L in ((32, 1024), (8, 4096)):
    g = torch.Generator(device="cuda").manual_seed(L)
    qkv = (torch.randn(nimg, L, 960, generator=g, device="cuda") * 1.5).to(torch.bfloat16)
    out = ops.attention_self(qkv, 8)
    # fp64 reference on the first image
    q, k, v = (t.reshape(L, 8, 40).permute(1, 0, 2).double() for t in qkv[0].split(320, dim=-1))
    ref = (torch.softmax(q @ k.transpose(1, 2) * 40 ** -0.5, -1) @ v).permute(1, 0, 2).reshape(L, 320)
    err = (out[0].double() - ref).norm() / ref.norm()
    for _ in range(5):
        ops.attention_self(qkv, 8)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(50):
        ops.attention_self(qkv, 8)
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) / 50 * 1e3
    flops = 4.0 * nimg * 8 * L * L * 40
    print(f"{arm:12s} nimg={nimg} L={L} d=40: {us:8.1f} us  {flops / us / 1e6:6.0f} TFLOP/s  rel-L2 vs fp64 {err:.3e}", flush=True)
