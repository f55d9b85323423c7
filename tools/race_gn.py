"""Standalone reproducer for the two-stream nondeterminism of round 2: the small-image GroupNorm kernel
(gn_fused_small_kernel<8>, norm.hip) ALONE on one stream while another kernel runs on a second stream; the GroupNorm result must not
change.  Run once per library variant:
    python tools/race_gn.py                      # product build (packed fp32 VALU ops off, csrc/Makefile NOPK)
    NR_LIB_VARIANT=pk python tools/race_gn.py    # same sources WITH v_pk_{add,mul,fma}_f32 (make -C neurons_amd/csrc pk)
Every differing element is reported as (image, pixel, channel, lane of the wave, low/high element of the packed pair)."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurons_amd import ops  # noqa: E402

dev = "cuda"
torch.manual_seed(0)
NIMG, HW, C = 16, 64, 64                     # the tiny-network shape of the failing tests: 32 groups of 2 channels, 8 x 8 pixels
x = (torch.randn(NIMG, 8, 8, C, device=dev) * 2 + 0.5).to(torch.bfloat16)
g, b = torch.randn(C, device=dev), torch.randn(C, device=dev)
x2 = (torch.randn(NIMG, 8, 8, C, device=dev)).to(torch.bfloat16)
a = torch.randn(1024, 64, device=dev).to(torch.bfloat16)
w = (torch.randn(64, 64, device=dev) * 0.1).to(torch.bfloat16)
xc = torch.randn(NIMG, 8, 8, C, device=dev).to(torch.bfloat16)
wc = (torch.randn(64, 3, 3, 64, device=dev) * 0.05).to(torch.bfloat16)
qkv = torch.randn(NIMG, 64, 192, device=dev).to(torch.bfloat16)
ref = ops.groupnorm(x, g, b, groups=32, eps=1e-5, silu=True)
torch.cuda.synchronize()
# fp32 torch reference of the same op: which of the two (reference run / concurrent run) is the wrong one
tref = torch.nn.functional.silu(torch.nn.functional.group_norm(x.float().permute(0, 3, 1, 2), 32, g, b, 1e-5)).permute(0, 2, 3, 1)
print("variant:", os.environ.get("NR_LIB_VARIANT", "product (no packed fp32)"), "| solo run vs fp32 torch: max err",
      (ref.float() - tref).abs().max().item(), flush=True)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
ITERS = int(os.environ.get("RACE_ITERS", "300"))
# "big_*": the other stream's kernel fills the whole chip (65 536 workgroups), so the GroupNorm's waves SHARE their SIMDs with it
xbig = torch.randn(2048, 8, 8, C, device=dev).to(torch.bfloat16)
abig = torch.randn(1 << 18, 64, device=dev).to(torch.bfloat16)
for other in ("none", "gn", "gemm", "conv", "attn", "big_gn_silu", "big_gn_nosilu", "big_gemm"):
    bad = 0
    lanes, halves, examples = collections.Counter(), collections.Counter(), []
    for it in range(ITERS):
        with torch.cuda.stream(s2):
            for _ in range(4):
                if other == "gn":
                    ops.groupnorm(x2, g, b, groups=32, eps=1e-5, silu=True)
                elif other == "gemm":
                    ops.gemm(a, w)
                elif other == "conv":
                    ops.conv3x3(xc, wc)
                elif other == "attn":
                    ops.attention_self(qkv, 8)
            if other == "big_gn_silu":
                ops.groupnorm(xbig, g, b, groups=32, eps=1e-5, silu=True)
            elif other == "big_gn_nosilu":
                ops.groupnorm(xbig, g, b, groups=32, eps=1e-5, silu=False)
            elif other == "big_gemm":
                ops.gemm(abig, w)
        with torch.cuda.stream(s1):
            outs = [ops.groupnorm(x, g, b, groups=32, eps=1e-5, silu=True) for _ in range(4)]
        torch.cuda.synchronize()
        for o in outs:
            if not torch.equal(o, ref):
                bad += 1
                d = (o != ref).reshape(NIMG, HW, C).nonzero()
                for img, px, ch in d.tolist():
                    lanes[px] += 1            # one workgroup per (image, group): thread = pixel, so lane = pixel
                    halves["low" if ch % 2 == 0 else "high"] += 1
                    if len(examples) < 6:
                        examples.append((img, px, ch, float(ref.reshape(NIMG, HW, C)[img, px, ch]), float(o.reshape(NIMG, HW, C)[img, px, ch]),
                                         float(tref.reshape(NIMG, HW, C)[img, px, ch])))
    print(f"concurrent with {other:5s}: {bad} / {4 * ITERS} GroupNorm results differ", flush=True)
    if bad:
        rows = collections.Counter(l // 16 for l in lanes.elements())
        print(f"    differing elements by 16-lane row of the wave {dict(sorted(rows.items()))}, by element of the channel pair {dict(halves)}")
        print("    (image, pixel = lane, channel, solo result, concurrent result, fp32 torch):", examples, flush=True)
