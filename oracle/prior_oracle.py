"""Oracle (test infrastructure, never imported by the product) for the diffusion prior's sampling loop:
``BrainDiffusionPrior.p_sample`` / ``p_sample_loop_ddpm`` (/root/reference/model_variants/BrainModel_neurons.py:324-341,363-389) on top of
``dalle2_pytorch.DiffusionPrior.p_mean_variance`` / ``NoiseScheduler`` (dalle2-pytorch==1.15.6, requirements.txt:11).

PARITY UNPINNED: dalle2_pytorch is not vendored under /root/reference and the reference holds no test vectors for this path, so the
base-class arithmetic below is a restatement of the published algorithm (Ho et al. 2020 eq. 6-7 posterior; Nichol & Dhariwal 2021
cosine schedule), written independently of neurons_amd/prior.py (plain fp32 torch ops in the order the library applies them)."""
import math

import torch


class OracleNoiseScheduler:
    def __init__(self, timesteps=100, s=0.008):
        x = torch.linspace(0, timesteps, timesteps + 1, dtype=torch.float64)
        ac = torch.cos(((x / timesteps) + s) / (1 + s) * math.pi * 0.5) ** 2
        ac = ac / ac[0]
        betas = torch.clip(1 - (ac[1:] / ac[:-1]), 0, 0.999)
        alphas = 1.0 - betas
        alphas_cumprod = torch.cumprod(alphas, dim=0)
        alphas_cumprod_prev = torch.nn.functional.pad(alphas_cumprod[:-1], (1, 0), value=1.0)
        f32 = lambda v: v.to(torch.float32)
        self.num_timesteps = timesteps
        self.sqrt_alphas_cumprod = f32(torch.sqrt(alphas_cumprod))
        self.sqrt_one_minus_alphas_cumprod = f32(torch.sqrt(1.0 - alphas_cumprod))
        self.sqrt_recip_alphas_cumprod = f32(torch.sqrt(1.0 / alphas_cumprod))
        self.sqrt_recipm1_alphas_cumprod = f32(torch.sqrt(1.0 / alphas_cumprod - 1))
        posterior_variance = betas * (1.0 - alphas_cumprod_prev) / (1.0 - alphas_cumprod)
        self.posterior_log_variance_clipped = f32(torch.log(posterior_variance.clamp(min=1e-20)))
        self.posterior_mean_coef1 = f32(betas * torch.sqrt(alphas_cumprod_prev) / (1.0 - alphas_cumprod))
        self.posterior_mean_coef2 = f32((1.0 - alphas_cumprod_prev) * torch.sqrt(alphas) / (1.0 - alphas_cumprod))
        self.alphas_cumprod = alphas_cumprod


def p_sample(ns, net, x, t, text_cond, cond_scale=1.0, mode="x_start", clip_denoised=True, noise=None):
    """one ancestral step; net(x, times, **text_cond[, text_cond_drop_prob=, image_cond_drop_prob=]) -> prediction"""
    times = torch.full((x.shape[0],), t, device=x.device, dtype=torch.long)
    pred = net(x, times, **text_cond)
    if cond_scale != 1.0:
        null = net(x, times, text_cond_drop_prob=1.0, image_cond_drop_prob=1.0, **text_cond)
        pred = null + (pred - null) * cond_scale
    if mode == "v":
        x_start = ns.sqrt_alphas_cumprod[t] * x - ns.sqrt_one_minus_alphas_cumprod[t] * pred
    elif mode == "x_start":
        x_start = pred
    else:
        x_start = ns.sqrt_recip_alphas_cumprod[t] * x - ns.sqrt_recipm1_alphas_cumprod[t] * pred
    if clip_denoised and mode != "x_start":
        x_start = x_start.clamp(-1.0, 1.0)
    mean = ns.posterior_mean_coef1[t] * x_start + ns.posterior_mean_coef2[t] * x
    if t == 0:
        return mean, x_start
    return mean + (0.5 * ns.posterior_log_variance_clipped[t]).exp() * noise, x_start


def p_sample_loop_ddpm(ns, net, text_cond, noises, cond_scale=1.0, mode="x_start"):
    """noises = [x_T, eps_{T-1}, ..., eps_1] (explicit draws so that two implementations can be compared)"""
    x = noises[0]
    for k, i in enumerate(reversed(range(ns.num_timesteps))):
        x, _ = p_sample(ns, net, x, i, text_cond, cond_scale, mode, noise=None if i == 0 else noises[1 + k])
    return x
