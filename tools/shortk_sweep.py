"""Round 5: plan sweep of the short-K Linears (K = 640 / 1280) as the engine replays them -- a chain of identical launches in ONE hipGraph, weights
L2 / Infinity-Cache hot -- over tile / ring-depth / wave-count candidates of the tiled igemm (NR_IGEMM_FORCE), with the asm LDS-DMA form enabled
from 4 k-tiles (NR_IGEMM_ADMA_MINK, read once per process: run the tool once per value).  us per launch, boundary included.
Usage (GPU box): NR_IGEMM_ADMA_MINK=4 python tools/shortk_sweep.py > gpurun_out/r05_shortk_sweep_adma4.txt"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neurons_amd import ops  # noqa: E402

SHAPES = [(8192, 640, 640, 0), (8192, 1920, 640, 0), (8192, 5120, 640, 1), (8192, 640, 3200, 0), (2048, 1280, 1280, 0), (2048, 3840, 1280, 0),
          (2048, 10240, 1280, 1), (2048, 1280, 6400, 0), (512, 1280, 1280, 0), (512, 3840, 1280, 0)]
CANDS = ["", "128,160,1,2,-1,4", "128,160,1,3,-1,4", "128,160,1,4,-1,4", "128,128,1,2,-1,8", "128,128,1,3,-1,8", "128,128,1,4,-1,8", "128,128,1,2,-1,4",
         "128,128,1,4,-1,4", "128,64,1,2,-1,8", "128,64,1,3,-1,8", "128,64,1,4,-1,8", "128,64,1,4,-1,4", "64,64,1,2,-1,4", "64,64,1,4,-1,4", "64,64,1,6,-1,4",
         "64,32,1,4,-1,4", "64,32,1,8,-1,4", "256,128,1,2,-1,8", "256,160,1,2,-1,4"]
CHAIN, REPS = 48, 20


def chain_us(a, w, geglu):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        keep = []
        with torch.cuda.graph(g):
            for _ in range(CHAIN):
                keep.append(ops.gemm(a, w, geglu=bool(geglu)))
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(REPS):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / REPS / CHAIN


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    ops.g8p_mode(0)          # the tiled igemm only: this sweep is about its plan space
    gen = torch.Generator(device=dev).manual_seed(0)
    print(f"NR_IGEMM_ADMA_MINK={os.environ.get('NR_IGEMM_ADMA_MINK', '(default 24)')}; chain of {CHAIN} launches per graph, {REPS} replays; us per launch")
    for (M, N, K, geglu) in SHAPES:
        w = (torch.randn(N, K, generator=gen, device=dev) * 0.03).to(torch.bfloat16)
        a = torch.randn(M, K, generator=gen, device=dev).to(torch.bfloat16)
        res = []
        for c in CANDS:
            if c.startswith("128,160") and (N % 160 or geglu):
                continue
            if c.startswith("256,160") and (N % 160 or geglu):
                continue
            if c.startswith("64,32") and geglu:
                continue
            if c:
                os.environ["NR_IGEMM_FORCE"] = c
            else:
                os.environ.pop("NR_IGEMM_FORCE", None)
            try:
                res.append((chain_us(a, w, geglu), c or "heuristic"))
            except Exception as e:          # a plan the launcher refuses
                res.append((float("inf"), (c or "heuristic") + " (" + str(e)[:40] + ")"))
        os.environ.pop("NR_IGEMM_FORCE", None)
        base = [r for r in res if r[1] == "heuristic"][0][0]
        res.sort()
        gf = 2.0 * M * N * K / 1e9
        print(f"M={M} N={N} K={K} geglu={geglu} ({gf:.1f} GFLOP): heuristic {base:.2f} us ({gf / base / 1e3:.0f} TF) | " +
              ", ".join(f"[{c}] {u:.2f}" for u, c in res[:6]), flush=True)


if __name__ == "__main__":
    main()
