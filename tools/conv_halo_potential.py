"""k-loop ablations of the tiled igemm on 3x3 convs and Linears (what bounds it) and what a halo-tile 3x3 conv could gain: the tap-inner convs of BASELINE config 2 timed with the product library and with a build whose
activation tile is fetched for one tap in nine only (make -C neurons_amd/csrc experiments ABLATE_A=1 -> NR_LIB_VARIANT=exp; results of that
build are wrong by construction).  Each library runs in its own child process (NR_LIB_VARIANT is read at import).
Usage (GPU box): python tools/conv_halo_potential.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, torch
sys.path.insert(0, %r)
from neurons_amd import ops
dev = torch.device("cuda", 0)
def bench(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
for (nimg, H, W, N, Cin) in [(32, 32, 32, 320, 320), (32, 16, 16, 640, 640), (32, 8, 8, 1280, 1280), (160, 32, 32, 320, 320), (160, 16, 16, 640, 640)]:
    x = torch.randn(nimg, H, W, Cin, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, 3, 3, Cin, device=dev) * (9 * Cin) ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device=dev)
    r = torch.randn(nimg, H, W, N, device=dev).to(torch.bfloat16)
    wt = w.reshape(N, 9, Cin // 64, 64).permute(0, 2, 1, 3).contiguous()
    lib = ops._lib.load()
    out = torch.empty(nimg, H, W, N, dtype=torch.bfloat16, device=dev)
    def run():
        ops._lib.check(lib.nr_op_conv3x3_tap_inner(ops._stream(), ops._ptr(x), Cin, nimg, H, W, ops._ptr(wt), ops._ptr(b), None, 1, ops._ptr(r), ops._ptr(out), N))
    t = bench(run)
    fl = 2.0 * nimg * H * W * N * 9 * Cin
    print(f"conv M={nimg*H*W} N={N} K={9*Cin}: {t*1e3:7.1f} us {fl/t/1e9:6.0f} TF", flush=True)
for (M, N, K) in [(8192, 1920, 640), (8192, 640, 3200), (2048, 1280, 6400), (32768, 320, 1600)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device=dev)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    lib = ops._lib.load()
    def run():
        ops._lib.check(lib.nr_op_gemm(ops._stream(), ops._ptr(a), K, ops._ptr(w), ops._ptr(b), None, N, ops._ptr(out), N, M, N, K, 0))
    t = bench(run)
    fl = 2.0 * M * N * K
    print(f"lin  M={M} N={N} K={K}: {t*1e3:7.1f} us {fl/t/1e9:6.0f} TF", flush=True)
''' % ROOT
VARIANTS = os.environ.get("ABL_VARIANTS", ",abla,abl2,abl4,abl6,abl8,abl10").split(",")
# abla: A tile for one tap in nine; abl2: no LDS-DMA inside the k-loop; abl4: fragment reads in the first iteration only; abl6: both;
# abl8: no MFMAs; abl10: no MFMAs, no DMA   (make -C neurons_amd/csrc experiments ABLATE=<n>, copied to libneurons_amd_abl<n>.so)
for variant in VARIANTS:
    print(f"== library variant '{variant or 'product'}'", flush=True)
    env = dict(os.environ, NR_LIB_VARIANT=variant)
    subprocess.run([sys.executable, "-c", CHILD], env=env, check=False)
