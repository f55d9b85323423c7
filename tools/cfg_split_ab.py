"""Experiment: the two CFG halves of the U-Net evaluation (uncond / cond sample of one clip: independent until the CFG combine,
pipeline_neuroclips.py:435,478-480) as TWO concurrent batch-1 evaluations on two handles / two streams, against the one batch-2 evaluation.
At B = 1 most of the U-Net's launches below the 32x32 level are latency-bound (<= 1 round of tiles): two half-size launches in flight can
fill each other's ramps and tails.  Prints ms per U-Net evaluation for both schedules (same box, interleaved rounds)."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurons_amd import _lib, NativeUNet3D  # noqa: E402
from neurons_amd.synth import gpu_random_state_dict  # noqa: E402
from neurons_amd.unet3d import UNet3DConfig, state_dict_schema  # noqa: E402

dev = torch.device("cuda", 0)
ucfg = UNet3DConfig()
sd = {k: v.cpu() for k, v in gpu_random_state_dict(state_dict_schema(ucfg, _lib.NR_KIND_UNET3D), 1, dev).items()}
F, L = 16, 32
x = torch.randn(2, 4, F, L, L, device=dev)
ctx = torch.randn(2, 77, 768, device=dev)
full = NativeUNet3D(ucfg).to(dev)
full.load_state_dict(sd)
halves = [NativeUNet3D(ucfg).to(dev) for _ in range(2)]
for h in halves:
    h.load_state_dict(sd)
del sd
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def run_full():
    return full(x, 500, encoder_hidden_states=ctx).sample


def run_split():
    outs = [None, None]
    cur = torch.cuda.current_stream()
    for i in range(2):
        streams[i].wait_stream(cur)
        with torch.cuda.stream(streams[i]):
            outs[i] = halves[i](x[i:i + 1], 500, encoder_hidden_states=ctx[i:i + 1]).sample
    for s in streams:
        cur.wait_stream(s)
    return torch.cat(outs)


a, b = run_full(), run_split()
torch.cuda.synchronize()
d = (a - b).abs().max().item()
print(f"max |full - split| = {d:.3e} (ref max {a.abs().max().item():.3e})")


def t(fn, n=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


tf, ts = [], []
for _ in range(5):
    tf.append(t(run_full))
    ts.append(t(run_split))
print(f"U-Net evaluation, CFG batch 2 in one launch sequence: {statistics.median(tf):.3f} ms; two concurrent batch-1 sequences: {statistics.median(ts):.3f} ms "
      f"(x{statistics.median(tf) / statistics.median(ts):.3f})")
