"""Known-good ceiling for the U-Net's 1x1 GEMM shapes: torch (hipBLASLt / rocBLAS) bf16 matmul on the same GPU next to
the igemm kernel (measurement only; the product never calls a BLAS).  Usage: python tools/blas_ceiling.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurons_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
SHAPES = [(32768, 320, 320), (32768, 960, 320), (32768, 2560, 320), (32768, 320, 1280), (8192, 640, 640), (8192, 1920, 640),
          (8192, 5120, 640), (8192, 640, 2560), (2048, 1280, 1280), (2048, 3840, 1280), (2048, 10240, 1280), (2048, 1280, 5120),
          (512, 1280, 1280), (512, 3840, 1280), (512, 10240, 1280), (512, 1280, 5120)]


def bench(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


for M, N, K in SHAPES:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device=dev)
    bb = b.to(torch.bfloat16)
    t_nr = bench(lambda: ops.gemm(a, w, b, None))
    t_mm = bench(lambda: torch.matmul(a, w.t()))
    t_lin = bench(lambda: torch.nn.functional.linear(a, w, bb))
    fl = 2.0 * M * N * K
    print(f"M={M:6d} N={N:6d} K={K:5d}  igemm {t_nr*1e3:7.1f} us {fl/t_nr/1e9:6.0f} TF | torch.matmul {t_mm*1e3:7.1f} us {fl/t_mm/1e9:6.0f} TF | "
          f"F.linear+bias {t_lin*1e3:7.1f} us {fl/t_lin/1e9:6.0f} TF", flush=True)
