"""Standalone timing of the fused FeedForward + proj_out kernel (ffpanel.hip) against the two launches it replaces (LayerNorm-folded GEGLU
projection on the row-panel kernel, then a K = 4C GEMM for net.2: the folded K = 5C two-source form only exists inside the engine, so the
second launch here is the slightly cheaper plain net.2 + residual).  Usage (GPU box): python tools/ff_bench.py [M]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurons_amd import _lib, ops  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
C = 320
g = torch.Generator(device="cuda").manual_seed(0)
t = (torch.randn(M, C, generator=g, device="cuda") * 1.2 + 0.2).to(torch.bfloat16)
x = torch.randn(M, C, generator=g, device="cuda").to(torch.bfloat16)
gamma = 1.0 + 0.2 * torch.randn(C, generator=g, device="cuda")
beta = 0.1 * torch.randn(C, generator=g, device="cuda")
w1 = torch.randn(8 * C, C, generator=g, device="cuda") * C ** -0.5
b1 = 0.1 * torch.randn(8 * C, generator=g, device="cuda")
w2 = torch.randn(C, 4 * C, generator=g, device="cuda") * (4 * C) ** -0.5
b2 = 0.1 * torch.randn(C, generator=g, device="cuda")
wpo = torch.randn(C, C, generator=g, device="cuda") * C ** -0.5
bpo = 0.1 * torch.randn(C, generator=g, device="cuda")

# host-side folding once (what ops.ff_fused does per call)
wf = w1.float()
ws = (wf * gamma[None]).to(torch.bfloat16)
c = ws.float().sum(1)
b = (wf.double() @ beta.double()).float() + b1
wsp, _ = ops.geglu_permute(ws, None)
cp, bp = ops.geglu_permute(c[:, None], b)
wsp, cp, bp = wsp.contiguous(), cp.reshape(-1).contiguous(), bp.contiguous()
wc = torch.cat([wpo, wpo @ w2], 1).to(torch.bfloat16).contiguous()
bc = (bpo + wpo @ b2).contiguous()
w2b = w2.to(torch.bfloat16).contiguous()
out = torch.empty(M, C, dtype=torch.bfloat16, device="cuda")
hid = torch.empty(M, 4 * C, dtype=torch.bfloat16, device="cuda")
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream


w1p, b1p = ops.geglu_permute(w1.to(torch.bfloat16), b1.float())
gam, bet = gamma.contiguous(), beta.contiguous()


def fused_pack():
    _lib.check(lib.nr_op_ff_fused(st, t.data_ptr(), x.data_ptr(), out.data_ptr(), M, C, w1p.data_ptr(), gam.data_ptr(), bet.data_ptr(), b1p.data_ptr(),
                                  wc.data_ptr(), bc.data_ptr(), 1e-5))


def fused():      # stage stream already packed by fused_pack()
    _lib.check(lib.nr_op_ff_fused(st, t.data_ptr(), x.data_ptr(), out.data_ptr(), M, C, None, gam.data_ptr(), bet.data_ptr(), b1p.data_ptr(),
                                  wc.data_ptr(), bc.data_ptr(), 1e-5))


def two_launch():
    _lib.check(lib.nr_op_ln_gemm(st, t.data_ptr(), C, wsp.data_ptr(), cp.data_ptr(), bp.data_ptr(), 1e-5, None, 0, hid.data_ptr(), 4 * C, M, 8 * C, C, 1, 0))
    _lib.check(lib.nr_op_gemm(st, hid.data_ptr(), 4 * C, w2b.data_ptr(), b2.data_ptr(), t.data_ptr(), C, out.data_ptr(), C, M, C, 4 * C, 0))


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
    return s.elapsed_time(e) / iters * 1e3


flops = 2.0 * M * C * 13 * C
fused_pack()
outs = {}
for waves in (4, 8):
    ops.ff_waves(waves)
    us = bench(fused)
    outs[waves] = out.clone()
    print(f"M={M}: fused FF + proj_out, {waves} waves/WG       {us:8.1f} us  {flops / us / 1e6:7.0f} TFLOP/s (13 C^2 per row)", flush=True)
print("4-wave and 8-wave results bit-identical:", torch.equal(outs[4], outs[8]), flush=True)
us = bench(two_launch)
print(f"M={M}: GEGLU + net.2 (two launches)          {us:8.1f} us  {flops / us / 1e6:7.0f} TFLOP/s (13 C^2 per row)", flush=True)
