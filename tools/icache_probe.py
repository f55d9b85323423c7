"""Probe (round 5): does a launch pay for COLD CODE?  The engine's graphs alternate between ~10 kernel instantiations of 8-13 KB each; an isolated
chain of one instantiation re-runs hot code.  For one GEMM shape this times, per launch and in one hipGraph each: a chain of every single plan
(tile / ring / wave variants of the tiled igemm = different kernel instantiations doing the same arithmetic), and a chain that cycles through all
of them.  If instruction fetch mattered, the cycling chain would cost more than the mean of the homogeneous ones.
Usage (GPU box): python tools/icache_probe.py > gpurun_out/r05_icache_probe.txt"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neurons_amd import ops  # noqa: E402

PLANS = ["64,32,1,4,-1,4", "64,64,1,2,-1,4", "64,64,1,4,-1,4", "128,64,1,2,-1,4", "128,64,1,2,-1,8", "128,128,1,2,-1,4", "128,128,1,2,-1,8", "128,160,1,2,-1,4"]
CHAIN, REPS = 64, 20


def chain_us(a, w, plans):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        keep = []
        with torch.cuda.graph(g):
            for i in range(CHAIN):
                os.environ["NR_IGEMM_FORCE"] = plans[i % len(plans)]
                keep.append(ops.gemm(a, w))
        os.environ.pop("NR_IGEMM_FORCE", None)
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
    ops.g8p_mode(0)
    gen = torch.Generator(device=dev).manual_seed(0)
    for (M, N, K) in [(2048, 1280, 1280), (512, 1280, 1280), (8192, 640, 640)]:
        w = (torch.randn(N, K, generator=gen, device=dev) * 0.03).to(torch.bfloat16)
        a = torch.randn(M, K, generator=gen, device=dev).to(torch.bfloat16)
        single = [chain_us(a, w, [p]) for p in PLANS]
        mixed = chain_us(a, w, PLANS)
        mixed2 = chain_us(a, w, PLANS[::-1])
        print(f"M={M} N={N} K={K}: homogeneous chains " + " ".join(f"{u:.2f}" for u in single) + f" | mean {sum(single) / len(single):.2f} us | "
              f"cycling through all {len(PLANS)} instantiations {mixed:.2f} / {mixed2:.2f} us per launch", flush=True)


if __name__ == "__main__":
    main()
