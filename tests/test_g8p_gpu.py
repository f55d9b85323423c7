"""The 256-row ping-pong GEMM kernel (gemm8p.hip) against a torch fp32 reference AND against the tiled igemm (gemm.hip) on the same
inputs: both kernels accumulate every output over k in the same order (k-tiles ascending, two 32-wide MFMA steps per tile), so wherever the
tiled kernel runs without split-K the two results must be bit-identical; with split-K the fp32 summation order differs and they agree to a
bf16 ulp.  Shapes cover: M tails, N tails (N = 96 / 320 / 640 on 128- / 320- / 256-wide tiles), the two-source operand (K = c0 + c1),
GEGLU, bias + row vector + scale + activation + residual epilogues, tap-inner 3x3 convs with image borders, odd k-tile counts, nk = 4..45.
Tolerance vs fp32 torch (identical bf16 inputs): max err <= 2e-2 * max|ref| (the bar of tests/test_ops_gpu.py)."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))


def _bf(*shape, scale=1.0):
    return (torch.randn(*shape, device="cuda") * scale).to(torch.bfloat16).contiguous()


def _cmp(name, got, ref, tol=2e-2):
    err = (got.float() - ref).abs().max().item()
    scale = ref.abs().max().item()
    print(f"[{name}] max_err={err:.4e} (ref max {scale:.3e})")
    assert torch.isfinite(got.float()).all()
    assert err <= tol * scale, (name, err, scale)


def _both(fn):
    """fn() with the tiled igemm (mode 0) and with the ping-pong kernel forced (mode 2); restores the heuristic"""
    from neurons_amd import ops
    try:
        ops.g8p_mode(0)
        old = fn().clone()
        ops.g8p_mode(2)
        new = fn().clone()
        new2 = fn()
        assert torch.equal(new, new2), "ping-pong kernel: two runs differ"
        ops.g8p_phases(4)                  # the four-phase schedule is the same arithmetic in the same order as the (default) two-phase one
        assert torch.equal(new, fn()), "ping-pong kernel: 4 phases per k-tile differ from 2"
    finally:
        ops.g8p_phases(2)
        ops.g8p_mode(1)
    return old, new


def _vs_old(name, old, new, ref):
    d = (old.float() - new.float()).abs().max().item()
    print(f"[{name}] bit-identical to the tiled igemm: {torch.equal(old, new)} (max |diff| {d:.3e})")
    assert d <= 2.0 ** -6 * ref.abs().max().item(), (name, d)      # split-K arm of the tiled kernel: fp32 order differs, <= ~2 bf16 ulp of the range


@pytest.mark.parametrize("M,N,K", [(16384, 320, 320), (8192, 640, 640), (4096 + 40, 1280, 640), (2048, 96, 1920), (1024, 2560, 1280),
                                   (300, 320, 256), (163840 // 8, 320, 2880), (512, 1280, 2880 + 64)])
def test_gemm_bias_res(cuda, M, N, K):
    from neurons_amd import ops
    torch.manual_seed(M + N + K)
    a, w, res = _bf(M, K), _bf(N, K, scale=K ** -0.5), _bf(M, N)
    bias = torch.randn(N, device=cuda)
    old, new = _both(lambda: ops.gemm(a, w, bias, res))
    ref = a.float() @ w.float().t() + bias + res.float()
    _cmp(f"g8p gemm {M}x{N}x{K}", new, ref)
    _vs_old(f"g8p gemm {M}x{N}x{K}", old, new, ref)


@pytest.mark.parametrize("M,C", [(8192, 640), (2048 + 16, 320)])
def test_gemm_geglu(cuda, M, C):
    from neurons_amd import ops
    torch.manual_seed(5)
    a = _bf(M, C)
    w, b = _bf(8 * C, C, scale=C ** -0.5), torch.randn(8 * C, device=cuda)
    wp, bp = ops.geglu_permute(w, b)
    old, new = _both(lambda: ops.gemm(a, wp, bp, geglu=True))
    h = a.float() @ w.float().t() + b
    v, g = h.chunk(2, dim=-1)
    ref = v * torch.nn.functional.gelu(g)
    _cmp(f"g8p geglu {M}x{C}", new, ref)
    _vs_old(f"g8p geglu {M}x{C}", old, new, ref)


def test_gemm_ex_rowvec_scale_act_res(cuda):
    from neurons_amd import ops
    torch.manual_seed(6)
    M, N, K, F, hw = 16 * 256, 960, 320, 16, 256
    a, w, res = _bf(M, K), _bf(N, K, scale=K ** -0.5), _bf(M, N)
    bias = torch.randn(N, device=cuda)
    rv = torch.randn(F, N, device=cuda)
    old, new = _both(lambda: ops.gemm_ex(a, w, bias, rowvec=rv, rowvec_div=hw, rowvec_mod=F, res=res, act=1, out_scale=0.5))
    h = (a.float() @ w.float().t() + bias + rv[(torch.arange(M, device=cuda) // hw) % F]) * 0.5
    ref = h * torch.sigmoid(1.702 * h) + res.float()
    _cmp("g8p gemm_ex rowvec+scale+quick_gelu+res", new, ref)
    _vs_old("g8p gemm_ex", old, new, ref)


@pytest.mark.parametrize("M,c0,c1,N", [(4096, 320, 1280, 320), (2048 + 8, 640, 2560, 640), (1024, 64, 192, 128)])
def test_two_source_operand(cuda, M, c0, c1, N):
    from neurons_amd import ops
    torch.manual_seed(7)
    a0, a1 = _bf(M, c0), _bf(M, c1)
    w = _bf(N, c0 + c1, scale=(c0 + c1) ** -0.5)
    bias, res = torch.randn(N, device=cuda), _bf(M, N)
    old, new = _both(lambda: ops.gemm2(a0, a1, w, bias, res))
    ref = torch.cat([a0, a1], 1).float() @ w.float().t() + bias + res.float()
    _cmp(f"g8p two-source {M}x{N}x({c0}+{c1})", new, ref)
    _vs_old(f"g8p two-source {M}", old, new, ref)


@pytest.mark.parametrize("nimg,H,W,Cin,Cout", [(4, 32, 32, 320, 320), (16, 16, 16, 640, 640), (8, 8, 8, 1280, 1280), (3, 6, 10, 64, 128),
                                               (2, 5, 7, 128, 96), (1, 1, 1, 256, 64), (40, 32, 32, 320, 320)])
def test_conv3x3_tap_inner(cuda, nimg, H, W, Cin, Cout):
    from neurons_amd import ops
    import torch.nn.functional as Fn
    torch.manual_seed(8)
    x = _bf(nimg, H, W, Cin)
    w = _bf(Cout, 3, 3, Cin, scale=(9 * Cin) ** -0.5)
    bias = torch.randn(Cout, device=cuda)
    temb = torch.randn(nimg, Cout, device=cuda)
    res = _bf(nimg, H, W, Cout)
    old, new = _both(lambda: ops.conv3x3(x, w, bias, rowvec=temb, rowvec_div=H * W, res=res, tap_inner=True))
    y = Fn.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), bias, padding=1).permute(0, 2, 3, 1)
    ref = y + temb[:, None, None, :] + res.float()
    _cmp(f"g8p conv3x3 tap-inner {nimg}x{H}x{W} {Cin}->{Cout}", new, ref)
    _vs_old(f"g8p conv {nimg}x{H}x{W}", old, new, ref)


def test_many_repetitions_are_bit_reproducible(cuda):
    """race screen: a wrong wait / barrier placement shows up as rare wrong tiles that come and go with timing; 200 launches of two shapes
    (short and long K) interleaved with an unrelated streaming kernel must all be identical"""
    from neurons_amd import ops
    torch.manual_seed(9)
    a, w = _bf(16384, 640), _bf(1280, 640, scale=640 ** -0.5)
    x = _bf(8, 16, 16, 640)
    wc = _bf(640, 3, 3, 640, scale=(9 * 640) ** -0.5)
    big = torch.randn(64 << 20, device=cuda)
    try:
        ops.g8p_mode(2)
        r1, r2 = ops.gemm(a, w).clone(), ops.conv3x3(x, wc, tap_inner=True).clone()
        for phases in (4, 2):
            ops.g8p_phases(phases)
            for i in range(100):
                if i % 3 == 0:
                    big.mul_(1.0001)
                assert torch.equal(ops.gemm(a, w), r1), f"gemm repetition {i} ({phases} phases)"
                assert torch.equal(ops.conv3x3(x, wc, tap_inner=True), r2), f"conv repetition {i} ({phases} phases)"
    finally:
        ops.g8p_phases(2)
        ops.g8p_mode(1)
