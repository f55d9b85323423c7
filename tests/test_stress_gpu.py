"""Parity in the numerical regime of real checkpoints, without checkpoints: every other GPU parity figure in this repo is on
N(0, 1/fan_in) weights, which never produce SD-1.5's heavy-tailed activations, sharply peaked softmax rows or large time-embedding rows.
``neurons_amd.synth.stress_state_dict`` plants those three traits (seeded) into the random state dicts; ``STRESS_LEVELS`` names three strengths.

What a full-size evaluation showed (profiles/r06_stress_probe.txt, tools/stress_probe.py): at L1 the engine is at 2.2e-2 of the fp32 oracle
(PyTorch's own bf16-autocast evaluation of the same oracle: 2.8e-2); from L2 on ANY bf16 path is 0.4-0.7 away from fp32 (bf16 q / k move logits of
magnitude 10-40 by 0.1-0.2; the random-weight network amplifies it) and the engine stays below the PyTorch-bf16 figure at every level.  Hence:
  * the loop gate (BASELINE config 2, 50 DDIM steps, SparseCtrl on, CFG 8.5) runs at L1 against the fp32 oracle with the north-star bar
    PSNR >= 40 dB (measured 41.4 dB); its rel-L2 (6.9e-2: the loop amplifies the per-evaluation 2.2e-2) is held to the same loop run by PyTorch's
    bf16 autocast on the oracle (engine <= 1.1 x that) and to 1e-1;
  * L2 and L3 are held to the yardstick: engine error <= 1.1 x the error of torch's bf16-autocast evaluation of the oracle (same weights, inputs, GPU);
  * one sgm unCLIP U-Net forward (config 3's network, 64x64 latent) at L1 against its fp32 oracle: bar as the un-stressed 96x96 forward (3.5e-2) or 1.1 x the
    PyTorch-bf16 evaluation of the same oracle, whichever is larger, and 5e-2 absolute (the engine measures 3.2e-2 .. 3.5e-2 by kernel selection)."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from test_engine_gpu import metrics  # noqa: E402


def _nets(cuda, level):
    from neurons_amd import _lib, NativeSparseCtrl, NativeUNet3D
    from neurons_amd.sparsectrl import controlnet_config_from_unet
    from neurons_amd.synth import STRESS_LEVELS, gpu_random_state_dict, stress_state_dict
    from neurons_amd.unet3d import UNet3DConfig, state_dict_schema
    ucfg = UNet3DConfig()
    ccfg = controlnet_config_from_unet(ucfg, dict(
        set_noisy_sample_input_to_zero=True, use_simplified_condition_embedding=True, conditioning_channels=4,
        motion_module_kwargs=dict(attention_block_types=["Temporal_Self"], temporal_position_encoding_max_len=32)))
    usd = stress_state_dict(gpu_random_state_dict(state_dict_schema(ucfg, _lib.NR_KIND_UNET3D), 1, cuda), 7, **STRESS_LEVELS[level])
    csd = stress_state_dict(gpu_random_state_dict(state_dict_schema(ccfg, _lib.NR_KIND_SPARSECTRL), 2, cuda), 8, **STRESS_LEVELS[level])
    unet, ctrl = NativeUNet3D(ucfg).to(cuda), NativeSparseCtrl(ccfg).to(cuda)
    unet.load_state_dict({k: v.cpu() for k, v in usd.items()})
    ctrl.load_state_dict({k: v.cpu() for k, v in csd.items()})
    return ucfg, ccfg, usd, csd, unet, ctrl


def test_c2_50_step_loop_on_stressed_weights_vs_oracle(cuda):
    from neurons_amd import DDIMScheduler, NeuroclipsPipeline
    from oracle import animatediff_oracle as O
    ucfg, ccfg, usd, csd, unet, ctrl = _nets(cuda, "L1")
    sched = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="linear", steps_offset=1, clip_sample=False)
    pipe = NeuroclipsPipeline(vae=None, text_encoder=None, tokenizer=None, unet=unet, scheduler=sched, controlnet=ctrl).to(cuda)
    g = torch.Generator(device=cuda).manual_seed(0)
    F, L, steps = 16, 32, 50
    lat, noise = torch.randn(1, 4, F, L, L, generator=g, device=cuda), torch.randn(1, 4, F, L, L, generator=g, device=cuda)
    ctx = torch.randn(2, 77, ucfg.cross_attention_dim, generator=g, device=cuda)
    cimg = torch.randn(1, 4, 1, L, L, generator=g, device=cuda) * 0.18215
    with torch.no_grad():
        want, _ = O.neuroclips_denoise(usd, O.OracleConfig.from_native(ucfg), csd, O.OracleConfig.from_native(ccfg), lat, noise, ctx, cimg,
                                       (0,), steps, 8.5)
        with torch.autocast("cuda", dtype=torch.bfloat16):      # the yardstick: PyTorch's own bf16 path through the SAME oracle loop
            yard, _ = O.neuroclips_denoise(usd, O.OracleConfig.from_native(ucfg), csd, O.OracleConfig.from_native(ccfg), lat, noise, ctx, cimg,
                                           (0,), steps, 8.5)
    out = pipe("", video_length=F, height=L * 8, width=L * 8, num_inference_steps=steps, guidance_scale=8.5, latents=lat, noise=noise,
               text_embeddings=ctx, controlnet_images=cimg, controlnet_image_index=[0], low_strength=0.3, output_type="latent").videos
    rel, psnr = metrics("stress L1: C2 final latents after 50 DDIM steps (full width, SparseCtrl, CFG 8.5) vs fp32 oracle", out, want)
    rel_y, psnr_y = metrics("stress L1: the oracle loop under torch bf16 autocast vs the fp32 oracle (yardstick)", yard.float(), want)
    assert psnr >= 40.0, f"PSNR {psnr:.1f} dB"
    # rel-L2: 3e-2 is the bar of the N(0, 1/fan_in) regime (measured 1.5e-2 there).  On L1 weights the 50-step loop amplifies the per-evaluation
    # error (2.2e-2, engine; 2.8e-2, PyTorch bf16) to 6.9e-2 (round 6) for the engine -- held to the PyTorch-bf16 loop on the same weights, and to
    # an absolute 1e-1; tools/stress_taps.py localises the per-op error at full width (profiles/r06_stress_taps.txt: no op stands out)
    assert rel <= max(3e-2, 1.1 * rel_y) and rel <= 1e-1, f"rel-L2 {rel:.3e} (PyTorch bf16 loop: {rel_y:.3e})"


@pytest.mark.parametrize("level", ["L2", "L3"])
def test_unet_evaluation_in_the_chaotic_regime_is_no_worse_than_torch_bf16(cuda, level):
    from oracle import animatediff_oracle as O
    ucfg, _, usd, _, unet, _ = _nets(cuda, level)
    oc = O.OracleConfig.from_native(ucfg)
    g = torch.Generator(device=cuda).manual_seed(0)
    sample = torch.randn(2, 4, 16, 32, 32, generator=g, device=cuda)
    ctx = torch.randn(2, 77, ucfg.cross_attention_dim, generator=g, device=cuda)
    with torch.no_grad():
        want = O.unet3d_forward(usd, oc, sample, 481, ctx)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            yard = O.unet3d_forward(usd, oc, sample, 481, ctx).float()
    got = unet(sample, 481, encoder_hidden_states=ctx).sample
    rel_e, _ = metrics(f"stress {level}: engine vs fp32 oracle", got, want)
    rel_y, _ = metrics(f"stress {level}: torch bf16-autocast oracle vs fp32 oracle (yardstick)", yard, want)
    assert torch.isfinite(got).all()
    assert rel_e <= 1.1 * rel_y + 5e-3, f"engine {rel_e:.3e} vs PyTorch bf16 {rel_y:.3e}"


def test_c3_unclip_unet_forward_on_stressed_weights_vs_oracle(cuda):
    from neurons_amd.sgm import NativeSGMUNet, SGMUNetConfig, sgm_state_dict_schema
    from neurons_amd.synth import STRESS_LEVELS, stress_state_dict
    from oracle import sgm_oracle as S
    cfg = SGMUNetConfig()
    g = torch.Generator(device=cuda).manual_seed(5)
    sd = {}
    for k, shape in sgm_state_dict_schema(cfg).items():
        z = torch.randn(shape, generator=g, device=cuda)
        sd[k] = 0.02 * z if k.endswith(".bias") else (1.0 + 0.1 * z if len(shape) == 1 else z / (int(np.prod(shape[1:])) ** 0.5))
    n_before = {k: v.clone() for k, v in sd.items() if v.dim() == 1 and k.endswith(".weight")}
    stress_state_dict(sd, 9, **STRESS_LEVELS["L1"])
    assert sum(1 for k, v in n_before.items() if not torch.equal(v, sd[k])) >= 100      # the sgm key names are recognised (norm gains changed)
    net = NativeSGMUNet(cfg).to(cuda)
    net.load_state_dict({k: v.cpu() for k, v in sd.items()})
    x = torch.randn(2, 4, 64, 64, generator=g, device=cuda)
    ctx = torch.randn(2, 256, 1664, generator=g, device=cuda)
    y = torch.randn(2, 1024, generator=g, device=cuda)
    t = torch.tensor([637.0, 637.0])
    got = net(x, t, context=ctx, y=y)
    with torch.no_grad():
        want = S.unet_forward(sd, cfg, x, t.to(cuda), ctx, y)
        with torch.autocast("cuda", dtype=torch.bfloat16):      # the yardstick: PyTorch's own bf16 path through the SAME oracle
            yard = S.unet_forward(sd, cfg, x, t.to(cuda), ctx, y).float()
    rel, psnr = metrics("stress L1: sgm unCLIP U-Net forward, 64x64 latent, vs fp32 oracle", got, want)
    rel_y, _ = metrics("stress L1: the sgm oracle under torch bf16 autocast vs the fp32 oracle (yardstick)", yard, want)
    # 3.5e-2 is the bar of the un-stressed 96x96 forward.  On L1 weights the depth-10 network sits AT that bar (3.2e-2 .. 3.5e-2 depending on which kernels serve the
    # C = 640 level: the register-panel form of lin160.hip, whose op-level error equals the tiled igemm's to four digits -- tools/panel_numerics.py -- moves the figure
    # from 3.22e-2 to 3.53e-2 by re-rolling bf16 roundings that the heavy-tailed network amplifies), so it is held to PyTorch's bf16 evaluation of the same oracle as
    # the other stress gates are, and to an absolute 5e-2
    assert rel <= max(3.5e-2, 1.1 * rel_y) and rel <= 5e-2 and psnr >= 35.0, f"rel-L2 {rel:.3e} (PyTorch bf16: {rel_y:.3e}), PSNR {psnr:.1f} dB"

