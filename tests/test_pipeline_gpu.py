"""GPU: the denoising loop (NeuroclipsPipeline.__call__) against the reference-generated C1 fixture:
8-frame 64x64 clip (8x8 latent), 10 DDIM steps, CFG 8.5, SparseCtrl on, explicit latents/noise/context.
Stated tolerance (north-star): PSNR >= 40 dB of the final latents w.r.t. the fp32 reference's dynamic range."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
sys.path.insert(0, os.path.dirname(HERE))

from test_engine_gpu import _tiny, metrics  # noqa: E402


def test_c1_loop_matches_reference_golden(cuda):
    from neurons_amd import DDIMScheduler, NeuroclipsPipeline
    g = np.load(os.path.join(GOLD, "c1_loop.npz"))
    unet, ctrl = _tiny()
    sched = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="linear", steps_offset=1, clip_sample=False)
    pipe = NeuroclipsPipeline(vae=None, text_encoder=None, tokenizer=None, unet=unet, scheduler=sched, controlnet=ctrl).to("cuda")
    out = pipe("", video_length=8, height=64, width=64, num_inference_steps=int(g["steps"]), guidance_scale=float(g["guidance"]),
               latents=torch.from_numpy(g["latents"]).cuda(), noise=torch.from_numpy(g["noise"]),
               text_embeddings=torch.from_numpy(g["ctx"]).cuda(), controlnet_images=torch.from_numpy(g["cimg"]).cuda(),
               controlnet_image_index=[0], low_strength=0.3, output_type="latent").videos
    assert list(sched.timesteps_host) == list(g["timesteps"])
    rel, psnr = metrics("C1 10-step loop final latents vs reference", out, g["final"])
    assert psnr >= 40.0, f"PSNR {psnr:.1f} dB < 40 dB"


@pytest.mark.parametrize("group", [1, 2, 3, 4])
def test_grouped_sparsectrl_schedule_matches_reference_and_per_step_schedule(cuda, group):
    """SparseCtrl (noisy sample zeroed) evaluated `group` DDIM steps at a time, one group ahead of the U-Net (forward_async /
    forward_after), against the reference fixture AND against the one-evaluation-per-step schedule.  10 steps: group 3 and 4 leave
    a partial last group.  The group size only changes GEMM tile plans (fp32 summation order), so the two schedules agree far
    inside the loop tolerance: >= 55 dB."""
    from neurons_amd import DDIMScheduler, NeuroclipsPipeline
    g = np.load(os.path.join(GOLD, "c1_loop.npz"))
    outs = []
    for grp in (1, group):
        unet, ctrl = _tiny()
        sched = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="linear", steps_offset=1, clip_sample=False)
        pipe = NeuroclipsPipeline(vae=None, text_encoder=None, tokenizer=None, unet=unet, scheduler=sched, controlnet=ctrl).to("cuda")
        pipe.controlnet_group = grp
        for _ in range(2):          # twice: the second clip re-uses the captured graphs and the context cache
            out = pipe("", video_length=8, height=64, width=64, num_inference_steps=int(g["steps"]), guidance_scale=float(g["guidance"]),
                       latents=torch.from_numpy(g["latents"]).cuda(), noise=torch.from_numpy(g["noise"]),
                       text_embeddings=torch.from_numpy(g["ctx"]).cuda(), controlnet_images=torch.from_numpy(g["cimg"]).cuda(),
                       controlnet_image_index=[0], low_strength=0.3, output_type="latent").videos
        outs.append(out.clone())
    _, psnr_ref = metrics(f"group {group}: final latents vs reference", outs[1], g["final"])
    _, psnr_sched = metrics(f"group {group} vs per-step schedule", outs[1], outs[0])
    assert psnr_ref >= 40.0 and psnr_sched >= 55.0


def test_pipeline_input_errors(cuda):
    from neurons_amd import DDIMScheduler, NeuroclipsPipeline
    unet, ctrl = _tiny()
    pipe = NeuroclipsPipeline(None, None, None, unet, DDIMScheduler(beta_start=0.00085, beta_end=0.012, clip_sample=False, steps_offset=1), ctrl).to("cuda")
    with pytest.raises(ValueError, match="divisible by 8"):
        pipe("", video_length=8, height=60, width=64, text_embeddings=torch.zeros(2, 77, 64))
    with pytest.raises(ValueError, match="prompt"):
        pipe(3, video_length=8, height=64, width=64)
    with pytest.raises(ValueError, match="Unexpected latents shape"):
        pipe("", video_length=8, height=64, width=64, latents=torch.zeros(1, 4, 8, 4, 4), text_embeddings=torch.zeros(2, 77, 64))


def test_c1_loop_to_pixels_with_native_vae(cuda):
    """Whole video path in HIP: 10-step loop + first-stage decode; pixels against the oracle decode of the REFERENCE's
    final latents (so the error includes everything the loop accumulated)."""
    from neurons_amd import DDIMScheduler, NeuroclipsPipeline
    from neurons_amd.vae import NativeVAEDecoder, vae_random_state_dict
    from oracle import vae_oracle as V
    from tiny_configs import tiny_vae_config
    g = np.load(os.path.join(GOLD, "c1_loop.npz"))
    unet, ctrl = _tiny()
    vcfg = tiny_vae_config()
    vsd = vae_random_state_dict(vcfg, seed=91)
    vae = NativeVAEDecoder(vcfg).to("cuda")
    vae.load_state_dict(vsd)
    sched = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="linear", steps_offset=1, clip_sample=False)
    pipe = NeuroclipsPipeline(vae=vae, text_encoder=None, tokenizer=None, unet=unet, scheduler=sched, controlnet=ctrl).to("cuda")
    video = pipe("", video_length=8, height=64, width=64, num_inference_steps=int(g["steps"]), guidance_scale=float(g["guidance"]),
                 latents=torch.from_numpy(g["latents"]).cuda(), noise=torch.from_numpy(g["noise"]),
                 text_embeddings=torch.from_numpy(g["ctx"]).cuda(), controlnet_images=torch.from_numpy(g["cimg"]).cuda(),
                 controlnet_image_index=[0], low_strength=0.3, output_type="tensor").videos
    assert tuple(video.shape) == (1, 3, 8, 64, 64) and video.dtype == torch.float32
    with torch.no_grad():
        ref = V.decode_latents(vsd, torch.from_numpy(g["final"]), len(vcfg.ch_mult), vcfg.num_res_blocks)
    rel, psnr = metrics("C1 loop + native VAE pixels vs oracle decode of reference latents", video, ref)
    assert psnr >= 35.0
