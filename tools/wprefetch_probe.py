"""Probe (round 5): how much of a mid-size / small-M Linear's time is the FIRST-TOUCH latency of its weights?

Per denoising step every layer's weights come from HBM (2.55 GB of U-Net weights >> the 256 MiB Infinity Cache), so every k-tile of a weight
panel is an HBM miss for the first workgroup that touches it, and the tiled igemm keeps only 1-3 k-tiles in flight.  This tool times a chain
of IDENTICAL GEMM launches (one hipGraph, as the engine replays them) in three regimes:
  hot      : the same weight buffer every launch (L2 / Infinity-Cache resident: the latency floor)
  cold     : a ring of distinct weight buffers larger than the Infinity Cache (what the engine sees)
  cold+pf  : cold, plus a side branch of the graph that touches the weights of launch i+2 while launch i runs (one element per 128-byte
             line: pulls the lines through the fabric into the Infinity Cache / the toucher's L2)
Usage (GPU box): python tools/wprefetch_probe.py > gpurun_out/r05_wprefetch_probe.txt"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neurons_amd import ops  # noqa: E402

SHAPES = [(512, 1280, 1280), (512, 3840, 1280), (512, 10240, 1280), (2048, 1280, 1280), (2048, 3840, 1280), (8192, 640, 640), (8192, 1920, 640),
          (8192, 5120, 640), (2048, 10240, 1280)]
CHAIN = 64          # launches per graph
REPS = 20


def touch(w):
    # one bf16 per 128-byte line; the reduction result is discarded (a graph node of its own on the side stream)
    return w.view(-1, 64)[:, 0].float().sum()


def build(a, ws, mode):
    cur = torch.cuda.current_stream()
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    keep = []
    with torch.cuda.graph(g):
        cur = torch.cuda.current_stream()
        for i in range(CHAIN):
            if mode == "cold+pf":
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    keep.append(touch(ws[(i + 2) % len(ws)]))
            w = ws[0] if mode == "hot" else ws[i % len(ws)]
            keep.append(ops.gemm(a, w))
        if mode == "cold+pf":
            cur.wait_stream(side)
    return g, keep


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    gen = torch.Generator(device=dev).manual_seed(0)
    print(f"chain of {CHAIN} identical GEMM launches in one hipGraph, {REPS} replays; us per launch")
    for (M, N, K) in SHAPES:
        nbuf = max(CHAIN, int(600e6 / (N * K * 2)) + 1)          # ring > 2 x Infinity Cache
        nbuf = min(nbuf, 512)
        ws = [(torch.randn(N, K, generator=gen, device=dev) * 0.03).to(torch.bfloat16) for _ in range(nbuf)]
        a = (torch.randn(M, K, generator=gen, device=dev)).to(torch.bfloat16)
        line = f"M={M:5d} N={N:5d} K={K:5d} (W {N * K * 2 / 1e6:5.1f} MB, ring {nbuf}):"
        for mode in ("hot", "cold", "cold+pf"):
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                g, keep = build(a, ws, mode)
                for _ in range(3):
                    g.replay()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(REPS):
                    g.replay()
                e1.record()
                torch.cuda.synchronize()
            line += f"  {mode} {1e3 * e0.elapsed_time(e1) / REPS / CHAIN:7.2f}"
            del g, keep
        print(line, flush=True)
        del ws, a
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
