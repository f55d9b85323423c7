"""ORACLE — test infrastructure, NOT product code (see oracle/animatediff_oracle.py for the rules).

fp32 PyTorch restatement of the first-stage decoder: ``AutoencodingEngineLegacy.decode`` (generative_models/sgm/
models/autoencoder.py:490-494) -> ``Decoder.forward`` (sgm/modules/diffusionmodules/model.py:723-757), and the two
call sites ``DiffusionEngine.decode_first_stage`` (sgm/models/diffusion.py:118-135) and ``decode_latents``
(animatediff/pipelines/pipeline_animation.py:243-256).  All source is in /root/reference, so this file is PINNED
against golden vectors from the reference's own ``Decoder`` class (oracle/gen_golden.py: gen_vae).
Citations are relative to /root/reference/generative_models/sgm/modules/diffusionmodules/model.py unless noted.
"""
from typing import Dict

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


def _norm(sd, p, x):
    """Normalize(): GroupNorm(32, eps=1e-6, affine)  (:52-55)."""
    return F.group_norm(x, 32, sd[f"{p}.weight"], sd[f"{p}.bias"], 1e-6)


def resnet_block(sd: SD, p: str, x):
    """ResnetBlock.forward with temb=None, dropout 0  (:131-151)."""
    h = F.conv2d(F.silu(_norm(sd, f"{p}.norm1", x)), sd[f"{p}.conv1.weight"], sd[f"{p}.conv1.bias"], padding=1)
    h = F.conv2d(F.silu(_norm(sd, f"{p}.norm2", h)), sd[f"{p}.conv2.weight"], sd[f"{p}.conv2.bias"], padding=1)
    if f"{p}.nin_shortcut.weight" in sd:
        x = F.conv2d(x, sd[f"{p}.nin_shortcut.weight"], sd[f"{p}.nin_shortcut.bias"])
    return x + h


def attn_block(sd: SD, p: str, x):
    """AttnBlock.forward (:180-201): single-head attention over the h*w positions, scale C^-0.5."""
    b, c, hh, ww = x.shape
    hn = _norm(sd, f"{p}.norm", x)
    q, k, v = (F.conv2d(hn, sd[f"{p}.{n}.weight"], sd[f"{p}.{n}.bias"]).reshape(b, c, hh * ww).transpose(1, 2) for n in "qkv")
    s = torch.matmul(q, k.transpose(1, 2)) * c ** -0.5
    o = torch.matmul(s.softmax(dim=-1), v).transpose(1, 2).reshape(b, c, hh, ww)
    return x + F.conv2d(o, sd[f"{p}.proj_out.weight"], sd[f"{p}.proj_out.bias"])


def decode(sd: SD, z, num_levels: int, num_res_blocks: int, taps=None):
    """autoencoder.py:490-494 + Decoder.forward (:723-757)."""
    h = F.conv2d(z.float(), sd["post_quant_conv.weight"], sd["post_quant_conv.bias"])
    h = F.conv2d(h, sd["decoder.conv_in.weight"], sd["decoder.conv_in.bias"], padding=1)
    def tap(name, t):
        if taps is not None:
            taps[name] = t
    tap("decoder.conv_in", h)
    h = resnet_block(sd, "decoder.mid.block_1", h); tap("decoder.mid.block_1", h)
    h = attn_block(sd, "decoder.mid.attn_1", h); tap("decoder.mid.attn_1", h)
    h = resnet_block(sd, "decoder.mid.block_2", h); tap("decoder.mid.block_2", h)
    for lev in reversed(range(num_levels)):
        for j in range(num_res_blocks + 1):
            h = resnet_block(sd, f"decoder.up.{lev}.block.{j}", h); tap(f"decoder.up.{lev}.block.{j}", h)
        if lev != 0:
            # Upsample.forward (:67-71): nearest 2x then 3x3 conv
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = F.conv2d(h, sd[f"decoder.up.{lev}.upsample.conv.weight"], sd[f"decoder.up.{lev}.upsample.conv.bias"], padding=1)
            tap(f"decoder.up.{lev}.upsample", h)
    h = F.silu(_norm(sd, "decoder.norm_out", h))
    return F.conv2d(h, sd["decoder.conv_out.weight"], sd["decoder.conv_out.bias"], padding=1)


def encode_moments(sd: SD, x, num_levels: int, num_res_blocks: int, taps=None):
    """autoencoder.py:468-488 up to the posterior parameters: Encoder.forward (:584-609) -> quant_conv."""
    def tap(name, t):
        if taps is not None:
            taps[name] = t
    h = F.conv2d(x.float(), sd["encoder.conv_in.weight"], sd["encoder.conv_in.bias"], padding=1)
    tap("encoder.conv_in", h)
    for lev in range(num_levels):
        for j in range(num_res_blocks):
            h = resnet_block(sd, f"encoder.down.{lev}.block.{j}", h); tap(f"encoder.down.{lev}.block.{j}", h)
        if lev != num_levels - 1:
            # Downsample.forward (:84-91): zero-pad right/bottom by one, 3x3 stride-2 conv without padding
            h = F.conv2d(F.pad(h, (0, 1, 0, 1)), sd[f"encoder.down.{lev}.downsample.conv.weight"],
                         sd[f"encoder.down.{lev}.downsample.conv.bias"], stride=2)
            tap(f"encoder.down.{lev}.downsample", h)
    h = resnet_block(sd, "encoder.mid.block_1", h); tap("encoder.mid.block_1", h)
    h = attn_block(sd, "encoder.mid.attn_1", h); tap("encoder.mid.attn_1", h)
    h = resnet_block(sd, "encoder.mid.block_2", h); tap("encoder.mid.block_2", h)
    h = F.silu(_norm(sd, "encoder.norm_out", h))
    h = F.conv2d(h, sd["encoder.conv_out.weight"], sd["encoder.conv_out.bias"], padding=1)
    return F.conv2d(h, sd["quant_conv.weight"], sd["quant_conv.bias"])


def gaussian_sample(moments, noise):
    """DiagonalGaussianDistribution.__init__/sample (sgm/modules/distributions/distributions.py:25-42)."""
    mean, logvar = torch.chunk(moments, 2, dim=1)
    return mean + torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0)) * noise


def decode_first_stage(sd: SD, z, num_levels, num_res_blocks, scale_factor=0.18215):
    """sgm/models/diffusion.py:118-135."""
    return decode(sd, z / scale_factor, num_levels, num_res_blocks)


def decode_latents(sd: SD, latents, num_levels, num_res_blocks):
    """pipeline_animation.py:243-256 (frame by frame in the reference; frames are independent)."""
    b, c, f, h, w = latents.shape
    frames = (latents / 0.18215).permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)
    video = decode(sd, frames, num_levels, num_res_blocks)
    video = video.reshape(b, f, *video.shape[1:]).permute(0, 2, 1, 3, 4)
    return (video / 2 + 0.5).clamp(0, 1)
