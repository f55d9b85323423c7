"""ORACLE — test infrastructure, NOT product code (see oracle/animatediff_oracle.py for the rules).

fp32 PyTorch restatement of the sgm unCLIP keyframe denoising path: ``UNetModel.forward`` and the Euler-EDM /
DiscreteDenoiser / VanillaCFG loop that ``utils.unclip_recon`` drives.  All source is in /root/reference, so this
file is PINNED against golden vectors from the reference's own classes (oracle/gen_golden.py: gen_sgm).
Citations are relative to /root/reference/generative_models/sgm/modules unless noted.
"""
import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


def timestep_embedding(timesteps, dim, max_period=10000):
    """diffusionmodules/util.py:207-231."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32, device=timesteps.device) / half)
    args = timesteps[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def res_block(sd: SD, p: str, x, emb, groups=32):
    """ResBlock._forward — diffusionmodules/openaimodel.py:328-354 (no up/down, no scale-shift); GroupNorm32 util.py:274-276."""
    h = F.group_norm(x.float(), groups, sd[f"{p}.in_layers.0.weight"], sd[f"{p}.in_layers.0.bias"], 1e-5)
    h = F.conv2d(F.silu(h), sd[f"{p}.in_layers.2.weight"], sd[f"{p}.in_layers.2.bias"], padding=1)
    e = F.linear(F.silu(emb), sd[f"{p}.emb_layers.1.weight"], sd[f"{p}.emb_layers.1.bias"])
    h = h + e[..., None, None]
    h = F.group_norm(h.float(), groups, sd[f"{p}.out_layers.0.weight"], sd[f"{p}.out_layers.0.bias"], 1e-5)
    h = F.conv2d(F.silu(h), sd[f"{p}.out_layers.3.weight"], sd[f"{p}.out_layers.3.bias"], padding=1)
    if f"{p}.skip_connection.weight" in sd:
        x = F.conv2d(x, sd[f"{p}.skip_connection.weight"], sd[f"{p}.skip_connection.bias"])
    return x + h


def cross_attention(sd: SD, p: str, x, context, heads):
    """CrossAttention.forward — attention.py:281-344 (SDPA, scale d^-0.5, no qkv bias)."""
    q = F.linear(x, sd[f"{p}.to_q.weight"])
    context = x if context is None else context
    k = F.linear(context, sd[f"{p}.to_k.weight"])
    v = F.linear(context, sd[f"{p}.to_v.weight"])
    b, n, c = q.shape
    d = c // heads
    def split(t):
        return t.reshape(b, -1, heads, d).permute(0, 2, 1, 3)
    s = torch.matmul(split(q), split(k).transpose(-1, -2)) * d ** -0.5
    o = torch.matmul(s.softmax(dim=-1), split(v)).permute(0, 2, 1, 3).reshape(b, n, c)
    return F.linear(o, sd[f"{p}.to_out.0.weight"], sd[f"{p}.to_out.0.bias"])


def _ln(sd, p, x):
    return F.layer_norm(x, (x.shape[-1],), sd[f"{p}.weight"], sd[f"{p}.bias"], 1e-5)


def spatial_transformer(sd: SD, p: str, x, context, heads, depth, groups=32):
    """SpatialTransformer.forward — attention.py:702-723 (use_linear=True); BasicTransformerBlock._forward :551-572;
    FeedForward/GEGLU :87-113."""
    b, c, h, w = x.shape
    x_in = x
    y = F.group_norm(x, groups, sd[f"{p}.norm.weight"], sd[f"{p}.norm.bias"], 1e-6)
    y = y.permute(0, 2, 3, 1).reshape(b, h * w, c)
    y = F.linear(y, sd[f"{p}.proj_in.weight"], sd[f"{p}.proj_in.bias"])
    for dd in range(depth):
        bp = f"{p}.transformer_blocks.{dd}"
        y = cross_attention(sd, f"{bp}.attn1", _ln(sd, f"{bp}.norm1", y), None, heads) + y
        y = cross_attention(sd, f"{bp}.attn2", _ln(sd, f"{bp}.norm2", y), context, heads) + y
        hh = F.linear(_ln(sd, f"{bp}.norm3", y), sd[f"{bp}.ff.net.0.proj.weight"], sd[f"{bp}.ff.net.0.proj.bias"])
        val, gate = hh.chunk(2, dim=-1)
        y = F.linear(val * F.gelu(gate), sd[f"{bp}.ff.net.2.weight"], sd[f"{bp}.ff.net.2.bias"]) + y
    y = F.linear(y, sd[f"{p}.proj_out.weight"], sd[f"{p}.proj_out.bias"])
    y = y.reshape(b, h, w, c).permute(0, 3, 1, 2)
    return y + x_in


def unet_forward(sd: SD, cfg, x, timesteps, context, y, taps: Optional[dict] = None):
    """UNetModel.forward — diffusionmodules/openaimodel.py:816-853 with the block layout of :640-807.
    ``cfg`` is a neurons_amd.sgm.SGMUNetConfig (attribute access only)."""
    mc = cfg.model_channels
    chans = [m * mc for m in cfg.channel_mult]
    attn = [(2 ** i) in cfg.attention_resolutions for i in range(len(chans))]
    depth = list(cfg.transformer_depth)
    L = len(chans)
    t_emb = timestep_embedding(timesteps, mc)
    emb = F.linear(F.silu(F.linear(t_emb, sd["time_embed.0.weight"], sd["time_embed.0.bias"])), sd["time_embed.2.weight"], sd["time_embed.2.bias"])
    emb = emb + F.linear(F.silu(F.linear(y, sd["label_emb.0.0.weight"], sd["label_emb.0.0.bias"])), sd["label_emb.0.2.weight"], sd["label_emb.0.2.bias"])
    hs = []
    h = F.conv2d(x, sd["input_blocks.0.0.weight"], sd["input_blocks.0.0.bias"], padding=1)
    hs.append(h)
    idx = 1
    for lev in range(L):
        for _ in range(cfg.num_res_blocks):
            h = res_block(sd, f"input_blocks.{idx}.0", h, emb)
            if attn[lev]:
                h = spatial_transformer(sd, f"input_blocks.{idx}.1", h, context, chans[lev] // cfg.num_head_channels, depth[lev])
            if taps is not None:
                taps[f"input_blocks.{idx}"] = h
            hs.append(h)
            idx += 1
        if lev != L - 1:
            h = F.conv2d(h, sd[f"input_blocks.{idx}.0.op.weight"], sd[f"input_blocks.{idx}.0.op.bias"], stride=2, padding=1)   # :198-207
            hs.append(h)
            idx += 1
    h = res_block(sd, "middle_block.0", h, emb)
    h = spatial_transformer(sd, "middle_block.1", h, context, chans[-1] // cfg.num_head_channels, depth[-1])
    h = res_block(sd, "middle_block.2", h, emb)
    if taps is not None:
        taps["middle_block"] = h
    idx = 0
    for lev in reversed(range(L)):
        for i in range(cfg.num_res_blocks + 1):
            h = torch.cat([h, hs.pop()], dim=1)
            h = res_block(sd, f"output_blocks.{idx}.0", h, emb)
            sub = 1
            if attn[lev]:
                h = spatial_transformer(sd, f"output_blocks.{idx}.1", h, context, chans[lev] // cfg.num_head_channels, depth[lev])
                sub = 2
            if lev and i == cfg.num_res_blocks:
                h = F.interpolate(h, scale_factor=2, mode="nearest")                                                       # :139-157
                h = F.conv2d(h, sd[f"output_blocks.{idx}.{sub}.conv.weight"], sd[f"output_blocks.{idx}.{sub}.conv.bias"], padding=1)
            if taps is not None:
                taps[f"output_blocks.{idx}"] = h
            idx += 1
    h = F.group_norm(h.float(), 32, sd["out.0.weight"], sd["out.0.bias"], 1e-5)
    return F.conv2d(F.silu(h), sd["out.2.weight"], sd["out.2.bias"], padding=1)


# ---- sampler ---------------------------------------------------------------------------------------
def legacy_ddpm_sigmas(n, linear_start=0.00085, linear_end=0.0120, num_timesteps=1000, append_zero=True, flip=False):
    """LegacyDDPMDiscretization — diffusionmodules/discretizer.py:42-69, make_beta_schedule util.py:20-33."""
    betas = torch.linspace(linear_start ** 0.5, linear_end ** 0.5, num_timesteps, dtype=torch.float64) ** 2
    ac = np.cumprod(1.0 - betas.numpy(), axis=0)
    if n < num_timesteps:
        ts = np.linspace(num_timesteps - 1, 0, n, endpoint=False).astype(int)[::-1]
        ac = ac[ts]
    sig = torch.flip(torch.tensor((1 - ac) / ac, dtype=torch.float32) ** 0.5, (0,))
    if append_zero:
        sig = torch.cat([sig, sig.new_zeros([1])])
    return torch.flip(sig, (0,)) if flip else sig


def euler_edm_sample(sd: SD, cfg, x, cond, uc, num_steps, scale):
    """EulerEDMSampler.__call__ (sampling.py:114-135,98-112,216-220) with DiscreteDenoiser/EpsScaling
    (denoiser.py:23-75, denoiser_scaling.py:29-37), VanillaCFG (guiders.py:24-42), OpenAIWrapper (wrappers.py:23-34)."""
    sigmas = legacy_ddpm_sigmas(num_steps).to(x.device)
    table = legacy_ddpm_sigmas(1000, append_zero=False, flip=True).to(x.device)
    x = x * torch.sqrt(1.0 + sigmas[0] ** 2.0)
    ctx = torch.cat((uc["crossattn"], cond["crossattn"]), 0)
    vec = torch.cat((uc["vector"], cond["vector"]), 0)
    for i in range(len(sigmas) - 1):
        sigma, nxt = sigmas[i], sigmas[i + 1]
        idx = (sigma - table).abs().argmin()
        sq = table[idx]
        c_in, c_out = 1 / (sq ** 2 + 1.0) ** 0.5, -sq
        xin = torch.cat([x] * 2)
        t = idx.reshape(1).expand(xin.shape[0])
        den = unet_forward(sd, cfg, xin * c_in, t, ctx, vec) * c_out + xin
        du, dc = den.chunk(2)
        den = du + scale * (dc - du)
        d = (x - den) / sigma
        x = x + d * (nxt - sigma)
    return x


def unclip_recon(sd: SD, cfg, tokens, vector_suffix, z, uc_tokens, noise, offset, num_steps, decode_first_stage,
                 scale=5.0, offset_noise_level=0.04):
    """utils.unclip_recon (utils.py:302-350) on explicit draws: ``z`` (:308), ``uc_tokens`` = randn_like(x) (:318),
    ``noise`` (:323), ``offset`` = randn(n) (:328-330).  ``decode_first_stage`` is the first-stage decode (:343,
    models/diffusion.py:118-135), e.g. ``lambda z: vae_oracle.decode_first_stage(...)``.  PINNED: the fixture
    tests/golden/unclip_tiny.npz was produced by calling the reference's own function (oracle/gen_golden.py: gen_unclip)."""
    n = z.shape[0]
    c = {"crossattn": tokens.repeat(n, 1, 1), "vector": vector_suffix.repeat(n, 1)}                  # :315
    uc = {"crossattn": uc_tokens.repeat(n, 1, 1), "vector": vector_suffix.repeat(n, 1)}             # :318-319
    sigmas = legacy_ddpm_sigmas(num_steps).to(z.device)                                               # :324
    if offset_noise_level > 0.0:
        noise = noise + offset_noise_level * offset.reshape(-1, 1, 1, 1)                              # :327-331
    noised_z = (z + noise * sigmas[0]) / torch.sqrt(1.0 + sigmas[0] ** 2.0)                           # :332-335
    samples_z = euler_edm_sample(sd, cfg, noised_z, c, uc, num_steps, scale)                          # :340
    return torch.clamp(decode_first_stage(samples_z) * 0.8 + 0.2, min=0.0, max=1.0)                   # :343-348
