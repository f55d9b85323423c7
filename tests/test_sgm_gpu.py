"""GPU: the sgm unCLIP U-Net + Euler-EDM/CFG loop through the C ABI against reference-generated vectors (tiny width)
and against the pinned oracle at unclip6.yaml width.  Tolerances as in test_engine_gpu.py."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
GOLD = os.path.join(HERE, "golden", "sgm_tiny.npz")

from test_engine_gpu import metrics  # noqa: E402


def _tiny_net():
    from neurons_amd.sgm import NativeSGMUNet, sgm_random_state_dict
    from tiny_configs import tiny_sgm_config
    cfg = tiny_sgm_config()
    net = NativeSGMUNet(cfg).to("cuda")
    net.load_state_dict(sgm_random_state_dict(cfg, seed=71))
    return net, cfg


def test_tiny_sgm_unet_matches_reference_golden(cuda):
    g = np.load(GOLD)
    net, _ = _tiny_net()
    x, ctx, y = (torch.from_numpy(g[k]).cuda() for k in ("x", "ctx", "y"))
    eps = net(x, torch.from_numpy(g["t"]).float(), context=ctx, y=y)
    rel, psnr = metrics("tiny sgm UNetModel eps vs reference", eps, g["eps"])
    assert rel < 2.5e-2 and psnr > 35
    assert torch.equal(eps, net(x, torch.from_numpy(g["t"]).float(), context=ctx, y=y))


def test_tiny_sgm_euler_loop_matches_reference_golden(cuda):
    from neurons_amd.sgm import EulerEDMSampler
    g = np.load(GOLD)
    net, _ = _tiny_net()
    ctx, y = torch.from_numpy(g["ctx"]).cuda(), torch.from_numpy(g["y"]).cuda()
    c = {"crossattn": ctx[1:2], "vector": y[1:2]}
    uc = {"crossattn": ctx[0:1], "vector": y[1:2]}
    final = EulerEDMSampler(num_steps=4, scale=5.0)(net, torch.from_numpy(g["z"]).cuda(), cond=c, uc=uc)
    rel, psnr = metrics("tiny sgm 4-step Euler/CFG loop vs reference", final, g["loop_final"])
    assert psnr >= 40.0


def test_full_width_sgm_unet_vs_oracle(cuda):
    """unclip6.yaml width (2.5 G parameters, context 1664, y 1024) at a 32x32 latent, CFG batch 2."""
    from neurons_amd.sgm import NativeSGMUNet, SGMUNetConfig, sgm_state_dict_schema
    from neurons_amd.synth import randn
    from oracle import sgm_oracle as S
    cfg = SGMUNetConfig()
    g = torch.Generator(device="cuda").manual_seed(5)
    sd = {}
    for k, shape in sgm_state_dict_schema(cfg).items():
        z = torch.randn(shape, generator=g, device="cuda")
        if k.endswith(".bias"):
            z = 0.02 * z
        elif len(shape) == 1:
            z = 1.0 + 0.1 * z
        else:
            z = z / (int(np.prod(shape[1:])) ** 0.5)
        sd[k] = z
    net = NativeSGMUNet(cfg).to("cuda")
    net.load_state_dict({k: v.cpu() for k, v in sd.items()})
    x = randn("s.x", (2, 4, 32, 32), 1).cuda()
    ctx = randn("s.ctx", (2, 256, 1664), 2).cuda()
    y = randn("s.y", (2, 1024), 3).cuda()
    t = torch.tensor([500.0, 500.0])
    eps = net(x, t, context=ctx, y=y, in_scale=0.5)
    with torch.no_grad():
        ref = S.unet_forward(sd, cfg, x * 0.5, t.cuda(), ctx, y)
    rel, psnr = metrics("full-width sgm UNetModel vs oracle", eps, ref)
    assert rel < 2.5e-2 and psnr > 35


# ---- drop-in boundary of the keyframe path: NativeDiffusionEngine driven exactly as utils.unclip_recon drives DiffusionEngine ----
def _tiny_engine(cuda):
    from neurons_amd.sgm import NativeDiffusionEngine, sgm_random_state_dict
    from neurons_amd.vae import vae_random_state_dict
    from tiny_configs import tiny_sgm_config, tiny_vae_config
    g = np.load(os.path.join(HERE, "golden", "unclip_tiny.npz"))
    cfg, vcfg = tiny_sgm_config(), tiny_vae_config()
    eng = NativeDiffusionEngine(network_config=cfg, first_stage_config=vcfg, num_steps=int(g["num_steps"]), scale=5.0)
    eng.eval().requires_grad_(False)
    eng.to(cuda)
    # checkpoint-style key names (recon_keyframe_neurons_enhance.py:322-324: diffusion_engine.load_state_dict(ckpt['state_dict']))
    ck = {"model.diffusion_model." + k: v for k, v in sgm_random_state_dict(cfg, seed=71).items()}
    ck.update({"first_stage_model." + k: v for k, v in vae_random_state_dict(vcfg, seed=91).items()})
    ck["denoiser.sigmas"] = torch.zeros(1000)
    ck["conditioner.embedders.0.dummy"] = torch.zeros(1)
    eng.load_state_dict(ck)
    return eng, g


def _psnr01(got, want):
    return 10 * np.log10(1.0 / (((got - want) ** 2).mean().item() + 1e-20))


def test_boundary_engine_runs_unclip_recon_call_sequence(cuda):
    """a test-local function with the call sequence of utils.unclip_recon (oracle/unclip_harness.py) drives
    NativeDiffusionEngine unchanged; expected pixels from the reference's own function (tests/golden/unclip_tiny.npz)."""
    from oracle.unclip_harness import call_like_unclip_recon
    eng, g = _tiny_engine(cuda)
    t = {k: torch.from_numpy(g[k]) for k in ("tokens", "vector_suffix", "z", "uc_tokens", "noise", "offset")}
    img = call_like_unclip_recon(t["tokens"].to(cuda), eng, t["vector_suffix"].to(cuda), t, num_samples=1, offset_noise_level=0.04, device=cuda)
    assert tuple(img.shape) == (1, 3, 768, 768) and img.dtype == torch.float32
    st = int(g["stride"])
    want = torch.from_numpy(g["samples_sub"].astype(np.float32))
    psnr = _psnr01(img[:, :, ::st, ::st].float().cpu(), want)
    print(f"[boundary: unclip_recon call sequence on NativeDiffusionEngine vs reference] psnr={psnr:.1f} dB")
    assert psnr >= 35.0, psnr
    assert abs(img.double().mean().item() - float(g["samples_mean"])) < 5e-3


def test_boundary_foreign_closure_takes_generic_path_and_agrees_with_fused(cuda):
    """sampler(closure, …): a closure the sampler cannot recognise (it closes over a plain dict) runs the reference's loop on torch tensors
    with every network evaluation still in HIP (DiscreteDenoiser -> NativeOpenAIWrapper -> nr_sgm_unet_forward); its result must agree
    with the fused path (nr_edm_cfg_euler_step) to fp32 rounding of the per-step scalars."""
    eng, g = _tiny_engine(cuda)
    ctx = torch.from_numpy(np.load(GOLD)["ctx"]).cuda()
    y = torch.from_numpy(np.load(GOLD)["y"]).cuda()
    c = {"crossattn": ctx[1:2], "vector": y[1:2]}
    uc = {"crossattn": ctx[0:1], "vector": y[1:2]}
    z = torch.from_numpy(np.load(GOLD)["z"]).cuda()
    box = {"den": eng.denoiser, "model": eng.model}
    seen = []

    def foreign(x, sigma, cc):
        seen.append(float(sigma[0]))
        return box["den"](box["model"], x, sigma, cc)

    def native(x, sigma, cc):
        return eng.denoiser(eng.model, x, sigma, cc)

    assert eng.sampler._native_network(foreign) is None and eng.sampler._native_network(native) is eng.model.diffusion_model
    a = eng.sampler(foreign, z, cond=c, uc=uc, num_steps=4)
    b = eng.sampler(native, z, cond=c, uc=uc, num_steps=4)
    assert len(seen) == 4
    rel, psnr = metrics("boundary: generic (foreign closure) vs fused sampler path", a, b)
    assert rel < 1e-3
    rel, psnr = metrics("boundary: foreign closure vs reference 4-step loop", a, np.load(GOLD)["loop_final"])
    assert psnr >= 40.0
    # model(x, t, c_dict) with the reference's key names (wrappers.py:23-34)
    gg = np.load(GOLD)
    x, t = torch.from_numpy(gg["x"]).cuda(), torch.from_numpy(gg["t"]).cuda()
    eps = eng.model(x, t, {"crossattn": ctx, "vector": y})
    rel, psnr = metrics("boundary: model(x, t, c_dict) vs reference eps", eps, gg["eps"])
    assert rel < 2.5e-2 and psnr > 35


def test_reloading_emb_layers_rebuilds_the_stacked_projection(cuda):
    """the stacked time-embedding projection of the sgm U-Net is built from <block>.emb_layers.1.*; reloading
    those tensors after a forward must change the output (the converted copy used to be reused silently)."""
    net, cfg = _tiny_net()
    g = np.load(GOLD)
    x, ctx, y = (torch.from_numpy(g[k]).cuda() for k in ("x", "ctx", "y"))
    t = torch.from_numpy(g["t"]).float()
    net.auto_release_host_weights = False
    e0 = net(x, t, context=ctx, y=y)
    from neurons_amd.sgm import sgm_random_state_dict
    sd = sgm_random_state_dict(cfg, seed=71)
    part = {k: v * 1.5 + 0.1 for k, v in sd.items() if ".emb_layers.1." in k}
    assert part
    net.load_state_dict(part, strict=False)
    e1 = net(x, t, context=ctx, y=y)
    assert not torch.equal(e0, e1)
    sd.update(part)
    from oracle import sgm_oracle as S
    with torch.no_grad():
        ref = S.unet_forward({k: v.cuda() for k, v in sd.items()}, cfg, x, t.cuda(), ctx, y)
    rel, psnr = metrics("sgm U-Net after reloading emb_layers vs oracle with the new weights", e1, ref)
    assert rel < 2.5e-2 and psnr > 35
