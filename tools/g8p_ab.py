"""A/B of the 256-row ping-pong kernel (gemm8p.hip, mode 2) against the tiled igemm (gemm.hip, mode 0) on the big launch shapes:
SparseCtrl groups of BASELINE config 2 (G = 5: 10 samples), config 4 (8 clips), config 5 (32 f x 64x64, B = 1 and 4), the VAE decoder.
Interleaved rounds in ONE process (cdna_hip_programming.md 5.4 rule 24), random operands, median of 7 rounds of 10 launches each.
    python tools/g8p_ab.py [--quick]"""
import os
import sys
import statistics

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurons_amd import ops  # noqa: E402


def bf(*s, scale=1.0):
    return (torch.randn(*s, device="cuda") * scale).to(torch.bfloat16).contiguous()


def time_fn(fn, n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3          # us


def ab(name, fn, flop):
    arms = ((0, 4), (2, 4), (2, 2))          # (mode, phases per k-tile)
    t = {a: [] for a in arms}
    for mode, ph in arms:
        ops.g8p_mode(mode)
        ops.g8p_phases(ph)
        fn()
    torch.cuda.synchronize()
    for _ in range(7):
        for mode, ph in arms:
            ops.g8p_mode(mode)
            ops.g8p_phases(ph)
            t[(mode, ph)].append(time_fn(fn))
    ops.g8p_mode(1)
    ops.g8p_phases(2)
    a, b, c = (statistics.median(t[k]) for k in arms)
    print(f"{name:52s} tiled {a:7.1f} us {flop / a / 1e6:6.0f} TF | ping-pong 4ph {b:7.1f} us {flop / b / 1e6:6.0f} TF x{a / b:4.2f} | 2ph {c:7.1f} us {flop / c / 1e6:6.0f} TF x{a / c:4.2f}",
          flush=True)


def main():
    quick = "--quick" in sys.argv
    torch.manual_seed(0)
    convs = [  # (nimg, H, Cin, Cout)
        ("C2 SparseCtrl group conv 32x32 320->320", 160, 32, 320, 320), ("C2 ctrl group conv 16x16 640->640", 160, 16, 640, 640),
        ("C2 ctrl group conv 8x8 1280->1280", 160, 8, 1280, 1280), ("C4 U-Net conv 32x32 320->320 (8 clips)", 256, 32, 320, 320),
        ("C4 U-Net conv 16x16 640->640", 256, 16, 640, 640), ("C4 U-Net conv 8x8 1280->1280", 256, 8, 1280, 1280),
        ("C5 U-Net conv 64x64 320->320 (B=1)", 64, 64, 320, 320), ("C5 conv 32x32 640->640", 64, 32, 640, 640),
        ("C5 conv 16x16 1280->1280", 64, 16, 1280, 1280), ("VAE conv 128x128 256->256 (16 frames)", 16, 128, 256, 256),
        ("VAE conv 256x256 128->128", 16, 256, 128, 128), ("VAE conv 64x64 512->512", 16, 64, 512, 512),
        ("C2 U-Net conv 32x32 320->320 (B=1)", 32, 32, 320, 320), ("C2 U-Net conv 32x32 640->640", 32, 32, 640, 640),
    ]
    gemms = [  # (M, N, K, geglu)
        ("C2 ctrl group GEGLU 16x16 M=40960 N=5120 K=640", 40960, 5120, 640, True), ("C2 ctrl group qkv 16x16 N=1920 K=640", 40960, 1920, 640, False),
        ("C2 ctrl group N=K=640", 40960, 640, 640, False), ("C2 ctrl group 8x8 GEGLU N=10240 K=1280", 10240, 10240, 1280, True),
        ("C2 ctrl group 8x8 N=K=1280", 10240, 1280, 1280, False), ("C4 GEGLU 16x16 M=65536", 65536, 5120, 640, True),
        ("C4 N=K=640 M=65536", 65536, 640, 640, False), ("C5 N=K=640 M=262144", 262144, 640, 640, False),
        ("C5 GEGLU 32x32 M=262144 N=5120 K=640", 262144, 5120, 640, True), ("C2 U-Net 16x16 GEGLU M=8192", 8192, 5120, 640, True),
    ]
    if quick:
        convs, gemms = convs[:3], gemms[:3]
    for name, nimg, H, Cin, Cout in convs:
        x, w = bf(nimg, H, H, Cin), bf(Cout, 3, 3, Cin, scale=(9 * Cin) ** -0.5)
        bias = torch.randn(Cout, device="cuda")
        ab(name, lambda: ops.conv3x3(x, w, bias, tap_inner=True), 2.0 * nimg * H * H * Cout * 9 * Cin)
        del x, w
    for name, M, N, K, geglu in gemms:
        a, w = bf(M, K), bf(N, K, scale=K ** -0.5)
        bias = torch.randn(N, device="cuda")
        ab(name, lambda: ops.gemm(a, w, bias, geglu=geglu), 2.0 * M * N * K)
        del a, w


if __name__ == "__main__":
    main()
