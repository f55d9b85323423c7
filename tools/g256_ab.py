"""A/B of the 256-row role-alternating tile kernel (gemm256.hip) against the 128-row igemm kernel on the long-K shapes of BASELINE config 2:
same inputs, NR_IGEMM256=0 vs 2 (x split-K overrides), max |diff| between the two outputs and the time of each.
Usage (GPU box): python tools/g256_ab.py > gpurun_out/g256_ab.txt"""
import os
import sys

import torch

os.environ.setdefault("NR_LIB_VARIANT", "exp")      # make -C neurons_amd/csrc experiments
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurons_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
SHAPES = [
    ("conv", (32, 32, 32), 320, 320, True), ("conv", (32, 32, 32), 320, 640, False), ("conv", (32, 32, 32), 320, 960, False),
    ("conv", (32, 16, 16), 640, 640, True), ("conv", (32, 16, 16), 640, 320, False), ("conv", (32, 16, 16), 640, 1280, False),
    ("conv", (32, 16, 16), 640, 1920, False), ("conv", (32, 16, 16), 640, 960, False),
    ("conv", (32, 8, 8), 1280, 1280, True), ("conv", (32, 8, 8), 1280, 640, False), ("conv", (32, 8, 8), 1280, 2560, False),
    ("conv", (32, 8, 8), 1280, 1920, False), ("conv", (32, 4, 4), 1280, 1280, True), ("conv", (32, 4, 4), 1280, 2560, False),
    ("lin", 32768, 320, 1600, True), ("lin", 8192, 640, 3200, True), ("lin", 2048, 1280, 6400, True), ("lin", 512, 1280, 6400, True),
    ("lin", 32768, 320, 1280, True), ("lin", 8192, 640, 2560, True), ("lin", 2048, 1280, 5120, True),
    ("lin", 8192, 1920, 640, False), ("lin", 2048, 3840, 1280, False), ("lin", 2048, 1280, 1280, True), ("lin", 8192, 640, 640, True),
    ("lin", 32768, 960, 320, False),
]


def bench(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


torch.manual_seed(0)
for sh in SHAPES:
    kind = sh[0]
    if kind == "lin":
        _, M, N, K, res = sh
        a = torch.randn(M, K, device=dev).to(torch.bfloat16)
        w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
        b = torch.randn(N, device=dev)
        r = torch.randn(M, N, device=dev).to(torch.bfloat16) if res else None
        fn = lambda: ops.gemm(a, w, b, r)
        flops = 2.0 * M * N * K
        name = f"lin  M={M} N={N} K={K} res={int(res)}"
    else:
        _, (nimg, H, W), N, Cin, res = sh
        x = torch.randn(nimg, H, W, Cin, device=dev).to(torch.bfloat16)
        w = (torch.randn(N, 3, 3, Cin, device=dev) * (9 * Cin) ** -0.5).to(torch.bfloat16)
        b = torch.randn(N, device=dev)
        r = torch.randn(nimg, H, W, N, device=dev).to(torch.bfloat16) if res else None
        fn = lambda: ops.conv3x3(x, w, b, res=r)
        flops = 2.0 * nimg * H * W * N * 9 * Cin
        name = f"conv M={nimg*H*W} N={N} K={9*Cin} res={int(res)}"
    os.environ["NR_IGEMM256"] = "0"
    ref = fn().float()
    t0 = bench(fn)
    line = f"{name:40s} 128-row {t0*1e3:7.1f}us {flops/t0/1e9:5.0f}TF |"
    for sk in ("", "1", "2", "4", "8"):
        os.environ["NR_IGEMM256"] = "2"
        if sk:
            os.environ["NR_IGEMM256_SPLITK"] = sk
        else:
            os.environ.pop("NR_IGEMM256_SPLITK", None)
        try:
            out = fn().float()
            err = (out - ref).abs().max().item()
            t1 = bench(fn)
            line += f" sk={sk or 'auto'} {t1*1e3:6.1f}us {flops/t1/1e9:4.0f}TF err={err:.3g} |"
        except Exception as ex:
            line += f" sk={sk or 'auto'} FAIL {type(ex).__name__} |"
    os.environ.pop("NR_IGEMM256_SPLITK", None)
    print(line, flush=True)
