"""A/B of the producer / consumer tile kernel (experiments/gemmws.hip, NR_IGEMM_WS) against the tiled igemm: tap-inner 3x3 convs and plain Linears of
BASELINE config 2 (and the SparseCtrl-group batch), same inputs, max |diff| of the outputs, event-timed raw op calls.
Usage (GPU box): make -C neurons_amd/csrc experiments && python tools/ws_ab.py"""
import os
import sys

import torch

os.environ.setdefault("NR_LIB_VARIANT", "exp")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurons_amd import ops  # noqa: E402

lib = ops._lib.load()
dev = torch.device("cuda", 0)


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


def ab(name, run, out, flops):
    os.environ["NR_IGEMM_WS"] = "0"
    run()
    ref = out.float().clone()
    t0 = bench(run)
    line = f"{name:44s} igemm {t0*1e3:7.1f}us {flops/t0/1e9:5.0f}TF |"
    for (ncons, nprod, ns) in ((4, 4, 3), (4, 8, 3), (8, 4, 3), (8, 8, 3), (8, 8, 4)):
        os.environ["NR_IGEMM_WS"] = "2"
        os.environ["NR_IGEMM_WS_NCONS"] = str(ncons)
        os.environ["NR_IGEMM_WS_NPROD"] = str(nprod)
        os.environ["NR_IGEMM_WS_NS"] = str(ns)
        out.zero_()
        run()
        err = (out.float() - ref).abs().max().item()
        t1 = bench(run)
        line += f" c{ncons}p{nprod}s{ns} {t1*1e3:6.1f}us {flops/t1/1e9:4.0f}TF x{t0/t1:4.2f} e={err:.2g} |"
    print(line, flush=True)


torch.manual_seed(0)
for (nimg, H, W, N, Cin, res) in [(32, 32, 32, 320, 320, True), (32, 16, 16, 640, 640, True), (32, 8, 8, 1280, 1280, True), (32, 4, 4, 1280, 1280, True),
                                  (160, 32, 32, 320, 320, False), (160, 16, 16, 640, 640, True), (32, 32, 32, 256, 640, False), (31, 16, 16, 640, 320, True)]:
    x = torch.randn(nimg, H, W, Cin, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, 3, 3, Cin, device=dev) * (9 * Cin) ** -0.5).to(torch.bfloat16)
    wt = w.reshape(N, 9, Cin // 64, 64).permute(0, 2, 1, 3).contiguous()
    b = torch.randn(N, device=dev)
    r = torch.randn(nimg, H, W, N, device=dev).to(torch.bfloat16) if res else None
    out = torch.empty(nimg, H, W, N, dtype=torch.bfloat16, device=dev)

    def run(x=x, wt=wt, b=b, r=r, out=out, Cin=Cin, nimg=nimg, H=H, W=W, N=N):
        ops._lib.check(lib.nr_op_conv3x3_tap_inner(ops._stream(), ops._ptr(x), Cin, nimg, H, W, ops._ptr(wt), ops._ptr(b), None, 1, ops._ptr(r), ops._ptr(out), N))
    ab(f"conv M={nimg*H*W} N={N} K={9*Cin} res={int(res)}", run, out, 2.0 * nimg * H * W * N * 9 * Cin)
for (M, N, K, res) in [(8192, 640, 3200, True), (2048, 1280, 6400, True), (32768, 320, 1600, True), (8192, 1920, 640, False), (8192, 640, 640, True),
                       (2048, 1280, 1280, True), (8000, 608, 1280, True)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device=dev)
    r = torch.randn(M, N, device=dev).to(torch.bfloat16) if res else None
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)

    def run(a=a, w=w, b=b, r=r, out=out, M=M, N=N, K=K):
        ops._lib.check(lib.nr_op_gemm(ops._stream(), ops._ptr(a), K, ops._ptr(w), ops._ptr(b), ops._ptr(r), N, ops._ptr(out), N, M, N, K, 0))
    ab(f"lin  M={M} N={N} K={K} res={int(res)}", run, out, 2.0 * M * N * K)
