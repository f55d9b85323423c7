"""Driver for rocprofv3 --pmc passes over the row-panel kernel: a few launches of the three 32x32-level shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurons_amd import _lib, ops  # noqa: E402

dev = "cuda"
M, K = 32768, 320
g = torch.Generator(device=dev).manual_seed(0)
a = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
for N, geglu in ((320, 0), (960, 0), (2560, 1)):
    w = (torch.randn(N, K, generator=g, device=dev) * K ** -0.5).to(torch.bfloat16)
    c = w.float().sum(1).contiguous()
    b = torch.randn(N, device=dev)
    out = torch.empty(M, N // 2 if geglu else N, dtype=torch.bfloat16, device=dev)
    for _ in range(4):
        _lib.check(lib.nr_op_ln_gemm(st, a.data_ptr(), K, w.data_ptr(), c.data_ptr(), b.data_ptr(), 1e-5, None, 0, out.data_ptr(), out.shape[1],
                                     M, N, K, geglu, 0))
torch.cuda.synchronize()
