"""Standalone timing of the fused temporal-attention block (tattn.hip) against the three launches it replaces (LayerNorm + PE folded
q|k|v projection, strided 16 x 16 attention core, to_out + residual).  Usage (GPU box): python tools/tattn_bench.py [nbatch] [hw]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurons_amd import _lib, ops  # noqa: E402

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 2
hw = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
C, F = 320, 16
M = nb * F * hw
g = torch.Generator(device="cuda").manual_seed(0)
t = (torch.randn(M, C, generator=g, device="cuda") * 1.1).to(torch.bfloat16)
gamma = 1.0 + 0.2 * torch.randn(C, generator=g, device="cuda")
beta = 0.1 * torch.randn(C, generator=g, device="cuda")
wq, wk, wv, wo = (torch.randn(C, C, generator=g, device="cuda") * C ** -0.5 for _ in range(4))
bo = 0.1 * torch.randn(C, generator=g, device="cuda")
pe = ops.temporal_pe_table(F, C, t.device)
wqkv = torch.cat([wq, wk, wv])
rv = (pe.double() @ wqkv.double().t()).float().contiguous()
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
wob = wo.to(torch.bfloat16).contiguous()
qkv = torch.empty(M, 3 * C, dtype=torch.bfloat16, device="cuda")
att = torch.empty(M, C, dtype=torch.bfloat16, device="cuda")
t2 = t.clone()


def three():
    q = ops.gemm_ex(t, wqkv, None, ln=(gamma, beta), rowvec=rv, rowvec_div=hw, rowvec_mod=F)
    _lib.check(lib.nr_op_attention(st, 2, q.data_ptr(), None, att.data_ptr(), nb * F, hw, hw, C, 8, F, 1))
    _lib.check(lib.nr_op_gemm(st, att.data_ptr(), C, wob.data_ptr(), bo.data_ptr(), t.data_ptr(), C, t2.data_ptr(), C, M, C, C, 0))


ops.tattn_fused(t.clone(), nb, hw, gamma, beta, wq, wk, wv, wo, bo)
tt = t.clone()


gbt = (beta[None] + pe).contiguous()
gam, bof = gamma.contiguous(), bo.contiguous()


def fused():      # stream packed by the call above; no host-side work per call
    _lib.check(lib.nr_op_tattn_fused(st, tt.data_ptr(), nb, hw, None, None, None, None, gam.data_ptr(), gbt.data_ptr(), bof.data_ptr(), 1e-5))


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


flops = 2.0 * M * C * 4 * C
for name, fn in (("fused block (one launch)", fused), ("q|k|v + attention + to_out (three launches, host-side folding included)", three)):
    us = bench(fn)
    print(f"M={M}: {name:72s} {us:8.1f} us  {flops / us / 1e6:7.0f} TFLOP/s", flush=True)
