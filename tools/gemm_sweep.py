"""Tile / split-K / pipeline-depth sweep of the igemm kernel on the shapes of BASELINE config 2.
NR_IGEMM_FORCE="bm,bn,splitk,stages,order" overrides the heuristic per launch (test/tuning hook in gemm.hip).
Usage (GPU box): python tools/gemm_sweep.py > gpurun_out/gemm_sweep.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurons_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
# (kind, M or (nimg,H,W), N, K/Cin, res)
SHAPES = [
    ("lin", 32768, 320, 320, True), ("lin", 32768, 960, 320, False), ("lin", 32768, 320, 1280, True),
    ("lin", 8192, 640, 640, True), ("lin", 8192, 1920, 640, False), ("lin", 8192, 640, 2560, True),
    ("lin", 2048, 1280, 1280, True), ("lin", 2048, 3840, 1280, False), ("lin", 2048, 1280, 5120, True),
    ("lin", 512, 1280, 1280, True), ("lin", 512, 1280, 5120, True),
    ("conv", (32, 32, 32), 320, 320, True), ("conv", (32, 16, 16), 640, 640, True), ("conv", (32, 8, 8), 1280, 1280, True),
    ("conv", (32, 8, 8), 1280, 2560, False), ("conv", (32, 4, 4), 1280, 1280, True), ("conv", (32, 4, 4), 1280, 2560, False),
    ("conv", (32, 16, 16), 640, 1920, False), ("conv", (32, 32, 32), 320, 960, False),
]
CONFIGS = ["-1,-1,-1,2,-1,4", "128,128,1,2,-1,4", "128,128,1,2,-1,8", "128,64,1,2,-1,4", "128,64,1,2,-1,8", "64,64,1,2,-1,4",
           "128,160,1,2,-1,4", "256,128,1,2,-1,8", "128,128,1,3,-1,8", "128,64,1,3,-1,8", "64,64,1,3,-1,4",
           "128,128,2,2,-1,8", "128,64,2,2,-1,4", "128,64,4,2,-1,4", "128,64,2,2,-1,8", "128,64,4,2,-1,8",
           "64,64,2,2,-1,4", "64,64,4,2,-1,4", "64,64,8,2,-1,4", "256,128,2,2,-1,8", "256,128,4,2,-1,8"]

CONFIGS += ["128,128,1,4,-1,8", "128,64,1,4,-1,8", "128,64,1,4,-1,4", "64,64,1,4,-1,4", "256,128,1,3,-1,8", "128,160,1,3,-1,4", "128,128,1,3,-1,4",
            "128,64,2,3,-1,8", "128,64,4,3,-1,8", "64,64,4,3,-1,4", "64,64,2,4,-1,4", "64,64,4,4,-1,4", "128,64,2,4,-1,4", "128,64,4,4,-1,4",
            "256,128,2,3,-1,8", "128,160,2,3,-1,4", "128,160,4,3,-1,4", "128,160,8,3,-1,4", "128,160,4,2,-1,4", "128,160,8,2,-1,4"]
if os.environ.get("SWEEP_SET") == "ff":   # GEGLU projections and the folded net.2 + proj_out GEMMs (K = 5C)
    SHAPES = [("geglu", 32768, 2560, 320, False), ("geglu", 8192, 5120, 640, False), ("geglu", 2048, 10240, 1280, False), ("geglu", 512, 10240, 1280, False),
              ("lin", 32768, 320, 1600, True), ("lin", 8192, 640, 3200, True), ("lin", 2048, 1280, 6400, True), ("lin", 512, 1280, 6400, True)]
if os.environ.get("SWEEP_SET") == "order":   # tile-order sweep (5th field: 0 n-fastest, 1 m-fastest, G >= 2 grouped) on the shapes whose weights exceed one L2
    SHAPES = [("geglu", 8192, 5120, 640, False), ("lin", 8192, 1920, 640, False), ("lin", 8192, 640, 3200, True), ("lin", 8192, 640, 640, True),
              ("geglu", 2048, 10240, 1280, False), ("lin", 2048, 3840, 1280, False), ("lin", 2048, 1280, 1280, True), ("lin", 2048, 1280, 6400, True),
              ("geglu", 32768, 2560, 320, False), ("lin", 32768, 320, 1600, True),
              ("conv", (32, 32, 32), 320, 320, True), ("conv", (32, 16, 16), 640, 640, True), ("conv", (32, 8, 8), 1280, 1280, True),
              ("conv", (32, 16, 16), 640, 1920, False), ("conv", (32, 32, 32), 320, 960, False)]
    CONFIGS = ["-1,-1,-1,-1,-1,-1"] + [f"-1,-1,-1,-1,{o},-1" for o in (0, 1, 2, 4, 8, 16)]
if os.environ.get("SWEEP_CONFIGS"):      # e.g. SWEEP_CONFIGS="256,128,1,2,-1,4;256,160,1,2,-1,4"
    CONFIGS = ["-1,-1,-1,2,-1,4"] + os.environ["SWEEP_CONFIGS"].split(";")
if os.environ.get("SWEEP_EXTRA"):        # extra shapes for the VAE: "conv:16,128,128,256,256,0;lin:8192,640,2560,1"
    SHAPES = []
    for item in os.environ["SWEEP_EXTRA"].split(";"):
        kind, vals = item.split(":")
        v = [int(x) for x in vals.split(",")]
        SHAPES.append(("lin", v[0], v[1], v[2], bool(v[3])) if kind == "lin" else ("conv", (v[0], v[1], v[2]), v[3], v[4], bool(v[5])))


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


for sh in SHAPES:
    kind = sh[0]
    if kind in ("lin", "geglu"):
        _, M, N, K, res = sh
        a = torch.randn(M, K, device=dev).to(torch.bfloat16)
        w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
        b = torch.randn(N, device=dev)
        r = torch.randn(M, N, device=dev).to(torch.bfloat16) if res else None
        gg = kind == "geglu"
        fn = lambda: ops.gemm(a, w, b, r, geglu=gg)
        flops = 2.0 * M * N * K
        name = f"{kind:5s}M={M} N={N} K={K} res={int(res)}"
    else:
        _, (nimg, H, W), N, Cin, res = sh
        x = torch.randn(nimg, H, W, Cin, device=dev).to(torch.bfloat16)
        w = (torch.randn(N, 3, 3, Cin, device=dev) * (9 * Cin) ** -0.5).to(torch.bfloat16)
        b = torch.randn(N, device=dev)
        r = torch.randn(nimg, H, W, N, device=dev).to(torch.bfloat16) if res else None
        fn = lambda: ops.conv3x3(x, w, b, res=r)
        flops = 2.0 * nimg * H * W * N * 9 * Cin
        name = f"conv M={nimg*H*W} N={N} K={9*Cin} res={int(res)}"
    results = []
    for cfg in CONFIGS:
        if cfg.startswith("128,160") and N % 160 != 0:
            continue
        if cfg.startswith("256,128") and N % 128 != 0:
            continue
        os.environ["NR_IGEMM_FORCE"] = cfg
        try:
            ms = bench(fn)
        except Exception as ex:  # unsupported combination
            continue
        results.append((ms, cfg))
    os.environ.pop("NR_IGEMM_FORCE", None)
    base = bench(fn)
    results.sort()
    best = ", ".join(f"[{c}] {ms*1e3:.1f}us {flops/ms/1e9:.0f}TF" for ms, c in results[:8])
    print(f"{name:42s} heuristic {base*1e3:7.1f}us {flops/base/1e9:5.0f}TF | best: {best}", flush=True)
