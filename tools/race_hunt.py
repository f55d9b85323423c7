"""Timing-dependent mismatch hunt: B clips in one call vs B independent calls, repeated, under several engine settings."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from neurons_amd import _lib, DDIMScheduler, NativeSparseCtrl, NativeUNet3D, NeuroclipsPipeline  # noqa: E402
from neurons_amd.synth import randn  # noqa: E402
from neurons_amd.unet3d import random_state_dict  # noqa: E402
from tiny_configs import tiny_ctrl_config, tiny_unet_config  # noqa: E402


def psnr(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    mse = ((a - b) ** 2).mean().item()
    rng = (b.max() - b.min()).item()
    return 10 * np.log10(rng * rng / (mse + 1e-20))


def run(mode, reps=6):
    ucfg, ccfg = tiny_unet_config(), tiny_ctrl_config()
    unet = NativeUNet3D(ucfg).to("cuda")
    unet.load_state_dict(random_state_dict(ucfg, _lib.NR_KIND_UNET3D, seed=11))
    ctrl = NativeSparseCtrl(ccfg).to("cuda")
    ctrl.load_state_dict(random_state_dict(ccfg, _lib.NR_KIND_SPARSECTRL, seed=12))
    if "eager" in mode:
        unet.enable_graph(False)
        ctrl.enable_graph(False)
    sched = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="linear", steps_offset=1, clip_sample=False)
    pipe = NeuroclipsPipeline(None, None, None, unet, sched, ctrl).to("cuda")
    if "nooverlap" in mode:
        pipe.overlap_controlnet = False
    B = 2
    lat = randn("b.lat", (B, 4, 8, 8, 8), 1).cuda()
    noise = randn("b.noise", (B, 4, 8, 8, 8), 2)
    ctx_u, ctx_t = randn("b.ctxu", (B, 77, 64), 3), randn("b.ctxt", (B, 77, 64), 4)
    cimg = (randn("b.cimg", (B, 4, 1, 8, 8), 5) * 0.18215).cuda()
    kw = dict(video_length=8, height=64, width=64, num_inference_steps=3, guidance_scale=8.5, controlnet_image_index=[0],
              low_strength=0.3, output_type="latent")
    res = []
    outs = {}
    for r in range(reps):
        both = pipe([""] * B, latents=lat, noise=noise, text_embeddings=torch.cat([ctx_u, ctx_t]).cuda(), controlnet_images=cimg, **kw).videos
        ones = [pipe("", latents=lat[i:i + 1], noise=noise[i:i + 1], text_embeddings=torch.cat([ctx_u[i:i + 1], ctx_t[i:i + 1]]).cuda(),
                     controlnet_images=cimg[i:i + 1], **kw).videos for i in range(B)]
        p = [psnr(both[i:i + 1], ones[i]) for i in range(B)]
        same_both = outs.get("both") is None or torch.equal(outs["both"], both)
        same_ones = all(outs.get(i) is None or torch.equal(outs[i], ones[i]) for i in range(B))
        outs.setdefault("both", both)
        for i in range(B):
            outs.setdefault(i, ones[i])
        res.append(f"{p[0]:.0f}/{p[1]:.0f}{'' if same_both else ' BOTH-CHANGED'}{'' if same_ones else ' ONES-CHANGED'}")
    print(f"{mode:20s}", " | ".join(res), flush=True)


for m in sys.argv[1:] or ["default", "eager", "nooverlap", "default"]:
    run(m)
