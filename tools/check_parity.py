"""Parity check on REAL checkpoints (for maintainers; the build container has none): load the same weights into the
native networks and into the fp32 PyTorch restatement (oracle/, run on the GPU), evaluate one denoising step and a
short DDIM loop on seeded inputs, report rel-L2 / PSNR per output.  Test infrastructure: imports oracle/.

  python tools/check_parity.py --sd15 /path/to/stable-diffusion-v1-5 --inference-config configs/inference/inference-v3.yaml \
      --motion-module v3_sd15_mm.ckpt [--controlnet v3_sd15_sparsectrl_rgb.ckpt --controlnet-config latent_condition.yaml] \
      [--frames 16 --latent 32 --steps 4]
Without arguments it runs on seeded synthetic weights of the full SD-1.5 topology (same as tests/, larger)."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurons_amd import _lib, DDIMScheduler, NativeSparseCtrl, NativeUNet3D, NeuroclipsPipeline  # noqa: E402
from neurons_amd.sparsectrl import controlnet_config_from_unet  # noqa: E402
from neurons_amd.unet3d import UNet3DConfig, random_state_dict, state_dict_schema  # noqa: E402
from neurons_amd.weights import filter_motion_module  # noqa: E402
from oracle import animatediff_oracle as O  # noqa: E402


def metrics(name, got, want):
    got, want = got.detach().float().cpu(), want.detach().float().cpu()
    mse = ((got - want) ** 2).mean().item()
    rel = mse ** 0.5 / (want.pow(2).mean().item() ** 0.5 + 1e-12)
    rng = (want.max() - want.min()).item()
    psnr = 10 * np.log10(rng * rng / (mse + 1e-20))
    print(f"{name:55s} rel_l2 {rel:.3e}  PSNR {psnr:6.1f} dB  max|err| {(got - want).abs().max().item():.3e}")
    return psnr


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sd15")
    ap.add_argument("--inference-config")
    ap.add_argument("--motion-module")
    ap.add_argument("--controlnet")
    ap.add_argument("--controlnet-config")
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--latent", type=int, default=32)
    ap.add_argument("--steps", type=int, default=4)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    if a.sd15:
        import yaml
        extra = yaml.safe_load(open(a.inference_config))["unet_additional_kwargs"] if a.inference_config else {}
        unet = NativeUNet3D.from_pretrained_2d(a.sd15, subfolder="unet", unet_additional_kwargs=extra)
        ucfg = unet.config
        usd = dict(unet._pending)
        if a.motion_module:
            usd.update(filter_motion_module(torch.load(a.motion_module, map_location="cpu")))
        unet.load_state_dict(usd, strict=False)
    else:
        ucfg = UNet3DConfig()
        usd = random_state_dict(ucfg, _lib.NR_KIND_UNET3D, seed=1)
        unet = NativeUNet3D(ucfg)
        unet.load_state_dict(usd)
    missing = [k for k in state_dict_schema(ucfg) if k not in usd]
    if missing:
        raise SystemExit(f"{len(missing)} U-Net tensors missing, e.g. {missing[:3]} (pass --motion-module)")
    ck = dict(set_noisy_sample_input_to_zero=True, use_simplified_condition_embedding=True, conditioning_channels=4,
              motion_module_kwargs=dict(attention_block_types=["Temporal_Self"], temporal_position_encoding_max_len=32))
    if a.controlnet_config:
        import yaml
        ck = yaml.safe_load(open(a.controlnet_config)).get("controlnet_additional_kwargs", ck)
    ccfg = controlnet_config_from_unet(ucfg, ck)
    if a.controlnet:
        csd = torch.load(a.controlnet, map_location="cpu")
        csd = csd["controlnet"] if "controlnet" in csd else csd
        csd.pop("animatediff_config", "")
        csd = {k: v for k, v in csd.items() if "pos_encoder.pe" not in k}
    else:
        csd = random_state_dict(ccfg, _lib.NR_KIND_SPARSECTRL, seed=2)
    ctrl = NativeSparseCtrl(ccfg)
    ctrl.load_state_dict(csd)
    unet.to(dev)
    ctrl.to(dev)
    gu, gc = {k: v.float().to(dev) for k, v in usd.items()}, {k: v.float().to(dev) for k, v in csd.items()}
    ou, oc = O.OracleConfig.from_native(ucfg), O.OracleConfig.from_native(ccfg)

    F, L = a.frames, a.latent
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(2, 4, F, L, L, generator=g, device=dev)
    ctx = torch.randn(2, 77, ucfg.cross_attention_dim, generator=g, device=dev)
    cimg = torch.randn(1, 4, 1, L, L, generator=g, device=dev) * 0.18215
    cond = torch.zeros(1, 4, F, L, L, device=dev)
    cond[:, :, 0] = cimg[:, :, 0]
    mask = torch.zeros(1, 1, F, L, L, device=dev)
    mask[:, :, 0] = 1
    down, mid = ctrl(x, 500, encoder_hidden_states=ctx, controlnet_cond=cond, conditioning_mask=mask, return_dict=False)
    eps = unet(x, 500, encoder_hidden_states=ctx, down_block_additional_residuals=down, mid_block_additional_residual=mid).sample
    with torch.no_grad():
        rd, rm = O.sparse_controlnet_forward(gc, oc, x, 500, ctx, cond, mask, 1.0)
        ref = O.unet3d_forward(gu, ou, x, 500, ctx, rd, rm)
    metrics("SparseCtrl mid residual", mid.float(), rm)
    metrics("U-Net eps (one evaluation, t = 500, CFG batch 2)", eps, ref)
    sched = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="linear", steps_offset=1, clip_sample=False)
    pipe = NeuroclipsPipeline(None, None, None, unet, sched, ctrl).to(dev)
    lat = torch.randn(1, 4, F, L, L, generator=g, device=dev)
    noise = torch.randn(1, 4, F, L, L, generator=g, device=dev)
    out = pipe("", video_length=F, height=L * 8, width=L * 8, num_inference_steps=a.steps, guidance_scale=8.5, latents=lat, noise=noise.cpu(),
               text_embeddings=ctx, controlnet_images=cimg, controlnet_image_index=[0], low_strength=0.3, output_type="latent").videos
    with torch.no_grad():
        want, _ = O.neuroclips_denoise(gu, ou, gc, oc, lat, noise, ctx, cimg, (0,), a.steps, 8.5)
    p = metrics(f"{a.steps}-step DDIM loop, final latents", out, want)
    print("PASS" if p >= 40.0 else "FAIL (PSNR < 40 dB)")


if __name__ == "__main__":
    main()
