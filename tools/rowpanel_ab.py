"""A/B of the K = 320 Linears of the 32x32 level: run once with NR_ROWPANEL=0 (tiled igemm, + separate LayerNorm kernel where the
engine used one) and once with NR_ROWPANEL=1 (row-panel kernel, LayerNorm folded).  Prints us per call and TFLOP/s.
  NR_ROWPANEL=0 python tools/rowpanel_ab.py ; NR_ROWPANEL=1 python tools/rowpanel_ab.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurons_amd import ops  # noqa: E402


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    rp = os.environ.get("NR_ROWPANEL", "1") != "0"
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    K = 320
    for M in (32768, 131072):
        a = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
        gamma, beta = torch.ones(K, device=dev), torch.zeros(K, device=dev)
        for name, N, ln, geglu, res in (("square+res", 320, False, False, True), ("square", 320, False, False, False), ("q (ln)", 320, True, False, False),
                                        ("qkv (ln)", 960, True, False, False), ("qkv", 960, False, False, False),
                                        ("ff1 geglu (ln)", 2560, True, True, False), ("ff1 geglu", 2560, False, True, False)):
            w = (torch.randn(N, K, generator=g, device=dev) * K ** -0.5)
            wb = w.to(torch.bfloat16)
            bias = torch.randn(N, device=dev)
            r = torch.randn(M, N, generator=g, device=dev).to(torch.bfloat16) if res else None
            if geglu and not ln:
                wp, bp = ops.geglu_permute(wb, bias)
                fn = lambda: ops.gemm(a, wp, bp, geglu=True)
            elif geglu:
                wp, bp = ops.geglu_permute(wb, bias)
                if rp:
                    # folded LN on the permuted weight
                    ws = (wp.float() * gamma[None]).to(torch.bfloat16).contiguous()
                    c = ws.float().sum(1).contiguous()
                    from neurons_amd import _lib
                    out = torch.empty(M, N // 2, dtype=torch.bfloat16, device=dev)
                    lib = _lib.load()
                    st = torch.cuda.current_stream().cuda_stream
                    fn = lambda: _lib.check(lib.nr_op_ln_gemm(st, a.data_ptr(), K, ws.data_ptr(), c.data_ptr(), bp.data_ptr(), 1e-5, None, 0,
                                                              out.data_ptr(), N // 2, M, N, K, 1, 0))
                else:
                    fn = lambda: ops.gemm(ops.layernorm(a, gamma, beta), wp, bp, geglu=True)
            elif ln:
                if rp or N < 3 * K:
                    fn = lambda: ops.gemm_ex(a, w, bias, ln=(gamma, beta))      # host folding included in the timing: subtract below
                    ws = (w * gamma[None]).to(torch.bfloat16).contiguous()
                    c = ws.float().sum(1).contiguous()
                    b2 = bias.clone()
                    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
                    from neurons_amd import _lib
                    lib = _lib.load()
                    st = torch.cuda.current_stream().cuda_stream
                    fn = lambda: _lib.check(lib.nr_op_ln_gemm(st, a.data_ptr(), K, ws.data_ptr(), c.data_ptr(), b2.data_ptr(), 1e-5, None, 0,
                                                              out.data_ptr(), N, M, N, K, 0, 0))
                else:
                    fn = lambda: ops.gemm(ops.layernorm(a, gamma, beta), wb, bias)
            else:
                fn = lambda: ops.gemm(a, wb, bias, r)
            us = timeit(fn)
            fl = 2.0 * M * N * K
            print(f"rowpanel={int(rp)} M={M:6d} N={N:5d} {name:16s} {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
