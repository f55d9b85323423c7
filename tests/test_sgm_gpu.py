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
