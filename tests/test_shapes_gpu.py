"""GPU: ragged / non-square / edge shapes and the pipeline's other branches (no CFG, no ControlNet) against the oracle."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from test_engine_gpu import metrics  # noqa: E402


def _nets(boc=(64, 128, 128, 128), ctxd=64, seed=31, pe_len=24):
    from neurons_amd import _lib, NativeSparseCtrl, NativeUNet3D
    from neurons_amd.sparsectrl import controlnet_config_from_unet
    from neurons_amd.unet3d import UNet3DConfig, random_state_dict
    ucfg = UNet3DConfig(block_out_channels=boc, cross_attention_dim=ctxd)
    ucfg.motion_module_kwargs = dict(ucfg.motion_module_kwargs, temporal_position_encoding_max_len=pe_len)
    ccfg = controlnet_config_from_unet(ucfg, dict(set_noisy_sample_input_to_zero=True, use_simplified_condition_embedding=True,
                                                  conditioning_channels=4,
                                                  motion_module_kwargs=dict(attention_block_types=["Temporal_Self"],
                                                                            temporal_position_encoding_max_len=32)))
    usd = random_state_dict(ucfg, _lib.NR_KIND_UNET3D, seed=seed)
    csd = random_state_dict(ccfg, _lib.NR_KIND_SPARSECTRL, seed=seed + 1)
    unet, ctrl = NativeUNet3D(ucfg).to("cuda"), NativeSparseCtrl(ccfg).to("cuda")
    unet.load_state_dict(usd)
    ctrl.load_state_dict(csd)
    return unet, ctrl, ucfg, ccfg, usd, csd


@pytest.mark.parametrize("F,h,w,b,L", [(3, 24, 40, 2, 77), (1, 8, 16, 1, 5), (5, 16, 8, 4, 33)])
def test_ragged_shapes_unet_and_ctrl_vs_oracle(cuda, F, h, w, b, L):
    """Non-square latents, odd frame counts (including a single frame), batch 1 / 4, short and odd context lengths."""
    from neurons_amd.synth import randn
    from oracle import animatediff_oracle as O
    unet, ctrl, ucfg, ccfg, usd, csd = _nets()
    sample = randn("r.sample", (b, 4, F, h, w), 1).cuda()
    ctx = randn("r.ctx", (b, L, 64), 2).cuda()
    cond = (randn("r.cond", (1, 4, F, h, w), 3) * 0.2).cuda()
    mask = torch.zeros(1, 1, F, h, w, device="cuda")
    mask[:, :, 0] = 1
    down, mid = ctrl(sample, 333, encoder_hidden_states=ctx, controlnet_cond=cond, conditioning_mask=mask, return_dict=False)
    eps = unet(sample, 333, encoder_hidden_states=ctx, down_block_additional_residuals=down, mid_block_additional_residual=mid).sample
    with torch.no_grad():
        gu, gc = {k: v.cuda() for k, v in usd.items()}, {k: v.cuda() for k, v in csd.items()}
        rd, rm = O.sparse_controlnet_forward(gc, O.OracleConfig.from_native(ccfg), sample, 333, ctx, cond, mask, 1.0)
        ref = O.unet3d_forward(gu, O.OracleConfig.from_native(ucfg), sample, 333, ctx, rd, rm)
    rel, psnr = metrics(f"F={F} {h}x{w} b={b} L={L}: U-Net with SparseCtrl residuals vs oracle", eps, ref)
    assert rel < 2.5e-2 and psnr > 35
    rel, _ = metrics("mid residual vs oracle", mid.float(), rm)
    assert rel < 2.5e-2


def test_pipeline_without_cfg_and_without_controlnet_vs_oracle(cuda):
    """guidance_scale <= 1 (batch of 1 per clip, no CFG combine) and controlnet_images=None (U-Net only)."""
    from neurons_amd import DDIMScheduler, NeuroclipsPipeline
    from neurons_amd.synth import randn
    from oracle import animatediff_oracle as O
    unet, ctrl, ucfg, ccfg, usd, csd = _nets()
    sched = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="linear", steps_offset=1, clip_sample=False)
    pipe = NeuroclipsPipeline(None, None, None, unet, sched, ctrl).to("cuda")
    lat = randn("p.lat", (1, 4, 4, 8, 8), 1).cuda()
    noise = randn("p.noise", (1, 4, 4, 8, 8), 2)
    ctx = randn("p.ctx", (2, 77, 64), 3).cuda()
    cimg = (randn("p.cimg", (1, 4, 1, 8, 8), 4) * 0.18215).cuda()
    gu, gc = {k: v.cuda() for k, v in usd.items()}, {k: v.cuda() for k, v in csd.items()}
    ou, oc = O.OracleConfig.from_native(ucfg), O.OracleConfig.from_native(ccfg)
    kw = dict(video_length=4, height=64, width=64, num_inference_steps=4, low_strength=0.3, output_type="latent")
    # (1) no CFG, with ControlNet
    out = pipe("", latents=lat, noise=noise, text_embeddings=ctx[1:], controlnet_images=cimg, controlnet_image_index=[0],
               guidance_scale=1.0, **kw).videos
    with torch.no_grad():
        ref, _ = O.neuroclips_denoise(gu, ou, gc, oc, lat, noise.cuda(), ctx[1:], cimg, (0,), 4, 1.0)
    rel, psnr = metrics("4-step loop without CFG vs oracle", out, ref)
    assert psnr >= 40.0
    # (2) CFG, no ControlNet images
    out = pipe("", latents=lat, noise=noise, text_embeddings=ctx, guidance_scale=7.5, **kw).videos
    with torch.no_grad():
        ref, _ = O.neuroclips_denoise(gu, ou, None, None, lat, noise.cuda(), ctx, None, (0,), 4, 7.5)
    rel, psnr = metrics("4-step loop without ControlNet vs oracle", out, ref)
    assert psnr >= 40.0


def test_frame_count_limits(cuda):
    """the temporal PE table bounds the clip length (24 rows for the v3 motion module, 32 for SparseCtrl)."""
    from neurons_amd.synth import randn
    unet, ctrl, *_ = _nets()
    ctx = randn("l.ctx", (1, 77, 64), 2).cuda()
    x24 = randn("l.x", (1, 4, 24, 8, 8), 1).cuda()
    assert torch.isfinite(unet(x24, 10, encoder_hidden_states=ctx).sample).all()
    with pytest.raises(RuntimeError, match="position"):
        unet(randn("l.x25", (1, 4, 25, 8, 8), 1).cuda(), 10, encoder_hidden_states=ctx)
    unet32, *_ = _nets(pe_len=32)
    assert torch.isfinite(unet32(randn("l.x32", (1, 4, 32, 8, 8), 1).cuda(), 10, encoder_hidden_states=ctx).sample).all()
