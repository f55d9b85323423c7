"""GroupNorm(+SiLU) timings at the shapes of BASELINE config 2 (HIP events, 50 launches each).
Usage (GPU box): python tools/gn_bench.py           (slab kernel)   |   NR_GN_SLAB=0 python tools/gn_bench.py   (stats + apply passes)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurons_amd import ops  # noqa: E402

SHAPES = [(32, 1024, 320, 0), (32, 1024, 640, 320), (32, 1024, 320, 320), (32, 256, 640, 0), (32, 256, 320, 0), (32, 256, 1280, 640),
          (32, 256, 1280, 0), (32, 256, 640, 640), (32, 64, 1280, 0), (32, 64, 1280, 1280), (32, 64, 640, 0), (32, 16, 1280, 0), (32, 16, 1280, 1280)]
for nimg, hw, c0, c1 in SHAPES:
    x0 = torch.randn(nimg, hw, c0, device="cuda").to(torch.bfloat16)
    x1 = torch.randn(nimg, hw, c1, device="cuda").to(torch.bfloat16) if c1 else None
    C = c0 + c1
    g, b = torch.randn(C, device="cuda"), torch.randn(C, device="cuda")
    fn = lambda: ops.groupnorm(x0.view(nimg, hw, 1, c0), g, b, groups=32, eps=1e-5, silu=True, x1=None if x1 is None else x1.view(nimg, hw, 1, c1))
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(50):
        fn()
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) / 50 * 1e3
    mb = nimg * hw * C * 2 * 2 / 1e6
    print(f"gn nimg={nimg} hw={hw} C={c0}+{c1}: {us:7.1f} us  {mb / us * 1e-3 * 1e3:7.0f} GB/s (read+write once)", flush=True)
