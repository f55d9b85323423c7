"""One full-size U-Net evaluation (CFG batch 2, 16 frames, 32x32 latent) on "real-checkpoint regime" weights (neurons_amd.synth.stress_state_dict)
at several stress levels: HIP engine vs the fp32 oracle on the same GPU, with torch's bf16 autocast evaluation of the SAME oracle as a yardstick
of what a standard reduced-precision PyTorch path loses in that regime.  Usage: python tools/stress_probe.py [level ...]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neurons_amd import _lib, NativeUNet3D  # noqa: E402
from neurons_amd.synth import STRESS_LEVELS, gpu_random_state_dict, stress_state_dict  # noqa: E402
from neurons_amd.unet3d import UNet3DConfig, state_dict_schema  # noqa: E402
from oracle import animatediff_oracle as O  # noqa: E402

LEVELS = {"none": None, **STRESS_LEVELS,
          "qk3": dict(gain_outliers=(1.0, 1.0), qk_scale=3.0, temb_scale=1.0),
          "gain50": dict(gain_outliers=(20.0, 50.0), qk_scale=1.0, temb_scale=1.0),
          "temb100": dict(gain_outliers=(1.0, 1.0), qk_scale=1.0, temb_scale=100.0)}


def rel(a, b):
    return ((a.float() - b.float()).pow(2).mean().sqrt() / b.float().pow(2).mean().sqrt()).item()


def main():
    dev = torch.device("cuda", 0)
    cfg = UNet3DConfig()
    oc = O.OracleConfig.from_native(cfg)
    g = torch.Generator(device=dev).manual_seed(0)
    sample = torch.randn(2, 4, 16, 32, 32, generator=g, device=dev)
    ctx = torch.randn(2, 77, cfg.cross_attention_dim, generator=g, device=dev)
    for name in (sys.argv[1:] or list(LEVELS)):
        sd = gpu_random_state_dict(state_dict_schema(cfg, _lib.NR_KIND_UNET3D), 1, dev)
        if LEVELS[name]:
            stress_state_dict(sd, 7, **LEVELS[name])
        t0 = time.time()
        with torch.no_grad():
            want = O.unet3d_forward(sd, oc, sample, 481, ctx)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                yard = O.unet3d_forward(sd, oc, sample, 481, ctx).float()
            with torch.autocast("cuda", dtype=torch.float16):
                yard16 = O.unet3d_forward(sd, oc, sample, 481, ctx).float()
        net = NativeUNet3D(cfg).to(dev)
        net.load_state_dict({k: v.cpu() for k, v in sd.items()})
        got = net(sample, 481, encoder_hidden_states=ctx).sample
        torch.cuda.synchronize()
        print(f"[stress {name}] eps rms {want.pow(2).mean().sqrt().item():.3e}  engine rel-L2 {rel(got, want):.3e}  torch-autocast-bf16 rel-L2 {rel(yard, want):.3e}"
              f"  engine vs autocast-bf16 {rel(got, yard):.3e}  torch-autocast-fp16 rel-L2 {rel(yard16, want):.3e}"
              f"  finite={bool(torch.isfinite(got).all())}  ({time.time() - t0:.0f} s)", flush=True)
        del net, sd, want, yard, yard16, got
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
