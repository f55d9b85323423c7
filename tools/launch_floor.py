"""What a dependent chain of TRIVIAL kernels costs per kernel on this box: eager launches vs one captured hipGraph replayed (the engine's
mode).  The kernel is the library's own CFG + DDIM update on 256 elements (one workgroup).  Sets the floor under the ~550 launches of a
denoising step: launches x this number is time no kernel optimisation can recover.   python tools/launch_floor.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurons_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
eps = torch.randn(2, 256, device=dev)
x = torch.randn(1, 256, device=dev)
N = 1000


def chain():
    y = x
    for _ in range(N):
        y = ops.cfg_ddim_step(eps, y, 8.5, 0.9, 0.95)
    return y


chain()
torch.cuda.synchronize()
t0 = time.perf_counter()
chain()
torch.cuda.synchronize()
eager = (time.perf_counter() - t0) / N * 1e6
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    chain()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        out = chain()
torch.cuda.synchronize()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    g.replay()
e1.record()
torch.cuda.synchronize()
print(f"dependent chain of {N} one-workgroup kernels: eager {eager:.2f} us per kernel (host-bound), hipGraph replay {e0.elapsed_time(e1) * 1e3 / 10 / N:.2f} us per kernel")
