"""ORACLE — test infrastructure, NOT product code.

A plain fp32 PyTorch restatement of the reference's video-denoising path (AnimateDiff temporal U-Net +
SparseCtrl + DDIM/CFG loop), written as pure functions over a state dict that uses the reference's parameter
names.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the
shipped path (neurons_amd/) never does.

Parity status: PINNED for everything whose source is under /root/reference — the functions below are checked
against outputs of the reference's own classes (imported in the build container by oracle/gen_golden.py,
vectors committed under tests/golden/).  UNPINNED for the pieces the reference takes from un-vendored
diffusers==0.11.1 (DDIMScheduler, Timesteps, TimestepEmbedding): no reference test or fixture exists for them
(SURVEY.md §4, §8c); they are restated from the published algorithm and cross-checked against in-repo siblings
(generative_models/sgm/modules/diffusionmodules/util.py:207-231 for the sinusoid,
animatediff/utils/util.py:211-221 for the DDIM update).

Every function cites the reference file:line it follows (paths relative to /root/reference).
Tensors use the reference's layouts: activations "b c f h w", tokens "(b f) (h w) c".
"""
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


@dataclass
class OracleConfig:
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    down_block_types: Tuple[str, ...] = ("CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "DownBlock3D")
    up_block_types: Tuple[str, ...] = ("UpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D")
    layers_per_block: int = 2
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    cross_attention_dim: int = 768
    attention_head_dim: int = 8            # number of heads
    use_motion_module: bool = True
    motion_num_heads: int = 8
    motion_attention_blocks: int = 2       # len(attention_block_types)
    motion_pe_max_len: int = 24
    motion_module_mid_block: bool = False
    conditioning_channels: int = 4         # SparseCtrl
    set_noisy_sample_input_to_zero: bool = True

    @staticmethod
    def from_native(cfg) -> "OracleConfig":
        """Build from a neurons_amd.unet3d.UNet3DConfig (plain attribute copy; no product code is executed)."""
        mm = cfg.motion_module_kwargs
        return OracleConfig(
            in_channels=cfg.in_channels, out_channels=cfg.out_channels, block_out_channels=tuple(cfg.block_out_channels),
            down_block_types=tuple(cfg.down_block_types), up_block_types=tuple(cfg.up_block_types),
            layers_per_block=cfg.layers_per_block, norm_num_groups=cfg.norm_num_groups, norm_eps=cfg.norm_eps,
            cross_attention_dim=cfg.cross_attention_dim, attention_head_dim=cfg.attention_head_dim,
            use_motion_module=cfg.use_motion_module, motion_num_heads=mm.get("num_attention_heads", 8),
            motion_attention_blocks=len(mm.get("attention_block_types", ())),
            motion_pe_max_len=mm.get("temporal_position_encoding_max_len", 24),
            motion_module_mid_block=cfg.motion_module_mid_block, conditioning_channels=cfg.conditioning_channels,
            set_noisy_sample_input_to_zero=cfg.set_noisy_sample_input_to_zero)


# ------------------------------------------------------------------------------------------------
# leaf ops
# ------------------------------------------------------------------------------------------------
def inflated_conv3d(x, w, b, stride=1, padding=1):
    """InflatedConv3d.forward — animatediff/models/resnet.py:10-18: per-frame nn.Conv2d."""
    bsz, c, f, h, wd = x.shape
    y = x.permute(0, 2, 1, 3, 4).reshape(bsz * f, c, h, wd)          # "b c f h w -> (b f) c h w"
    y = F.conv2d(y, w, b, stride=stride, padding=padding)
    return y.reshape(bsz, f, *y.shape[1:]).permute(0, 2, 1, 3, 4)    # "(b f) c h w -> b c f h w"


def inflated_groupnorm(x, w, b, groups, eps):
    """InflatedGroupNorm.forward — resnet.py:21-29: GroupNorm per frame."""
    bsz, c, f, h, wd = x.shape
    y = x.permute(0, 2, 1, 3, 4).reshape(bsz * f, c, h, wd)
    y = F.group_norm(y, groups, w, b, eps)
    return y.reshape(bsz, f, c, h, wd).permute(0, 2, 1, 3, 4)


def timesteps_proj(timesteps, dim):
    """diffusers Timesteps(dim, flip_sin_to_cos=True, freq_shift=0) (unet.py:101,386).  Same arithmetic as the
    in-repo generative_models/sgm/modules/diffusionmodules/util.py:207-231 (cos first)."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(0, half, dtype=torch.float32, device=timesteps.device) / half)
    args = timesteps[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def timestep_embedding_mlp(sd: SD, t_emb):
    """diffusers TimestepEmbedding: Linear -> SiLU -> Linear (unet.py:104,392; cf. sgm openaimodel.py:590-594)."""
    h = F.linear(t_emb, sd["time_embedding.linear_1.weight"], sd["time_embedding.linear_1.bias"])
    h = F.silu(h)
    return F.linear(h, sd["time_embedding.linear_2.weight"], sd["time_embedding.linear_2.bias"])


def resnet_block3d(sd: SD, p: str, x, temb, groups, eps):
    """ResnetBlock3D.forward — resnet.py:182-212 (time_embedding_norm="default", output_scale_factor=1)."""
    h = inflated_groupnorm(x, sd[f"{p}.norm1.weight"], sd[f"{p}.norm1.bias"], groups, eps)
    h = F.silu(h)
    h = inflated_conv3d(h, sd[f"{p}.conv1.weight"], sd[f"{p}.conv1.bias"])
    t = F.linear(F.silu(temb), sd[f"{p}.time_emb_proj.weight"], sd[f"{p}.time_emb_proj.bias"])[:, :, None, None, None]
    h = h + t
    h = inflated_groupnorm(h, sd[f"{p}.norm2.weight"], sd[f"{p}.norm2.bias"], groups, eps)
    h = F.silu(h)
    h = inflated_conv3d(h, sd[f"{p}.conv2.weight"], sd[f"{p}.conv2.bias"])
    if f"{p}.conv_shortcut.weight" in sd:
        x = inflated_conv3d(x, sd[f"{p}.conv_shortcut.weight"], sd[f"{p}.conv_shortcut.bias"], padding=0)
    return x + h


ATTN_SCORE_BUDGET_BYTES = 4 << 30      # fp32 score matrix held at once by cross_attention (slicing over batch*heads; values unchanged)


def _heads_to_batch(t, heads):
    """CrossAttention.reshape_heads_to_batch_dim — motion_module_new.py:181-186."""
    b, s, d = t.shape
    return t.reshape(b, s, heads, d // heads).permute(0, 2, 1, 3).reshape(b * heads, s, d // heads)


def _batch_to_heads(t, heads):
    """CrossAttention.reshape_batch_dim_to_heads — motion_module_new.py:188-193."""
    b, s, d = t.shape
    return t.reshape(b // heads, heads, s, d).permute(0, 2, 1, 3).reshape(b // heads, s, d * heads)


def cross_attention(sd: SD, p: str, x, ctx, heads):
    """CrossAttention.forward + _attention — motion_module_new.py:201-287 (no bias on q/k/v, scale = d^-0.5,
    baddbmm(alpha=scale) -> softmax -> bmm, to_out[0] with bias, dropout p=0)."""
    q = F.linear(x, sd[f"{p}.to_q.weight"])
    ctx = x if ctx is None else ctx
    k = F.linear(ctx, sd[f"{p}.to_k.weight"])
    v = F.linear(ctx, sd[f"{p}.to_v.weight"])
    d = q.shape[-1] // heads
    q, k, v = _heads_to_batch(q, heads), _heads_to_batch(k, heads), _heads_to_batch(v, heads)
    # The (batch*heads) problems are independent, so they may be evaluated a slice at a time without changing any value: at BASELINE
    # config 5 (64 x 64 latent, 32 frames) the full fp32 score tensor would be 34 GB ("materialises ... scores").
    nb = q.shape[0]
    per = max(1, int(ATTN_SCORE_BUDGET_BYTES // max(1, q.shape[1] * k.shape[1] * q.element_size())))
    outs = []
    for b0 in range(0, nb, per):
        qs, ks, vs = q[b0:b0 + per], k[b0:b0 + per], v[b0:b0 + per]
        scores = torch.baddbmm(torch.empty(qs.shape[0], qs.shape[1], ks.shape[1], dtype=q.dtype, device=q.device), qs,
                               ks.transpose(-1, -2), beta=0, alpha=d ** -0.5)
        probs = scores.softmax(dim=-1)
        outs.append(torch.bmm(probs, vs))
    o = _batch_to_heads(outs[0] if len(outs) == 1 else torch.cat(outs), heads)
    return F.linear(o, sd[f"{p}.to_out.0.weight"], sd[f"{p}.to_out.0.bias"])


def feed_forward(sd: SD, p: str, x):
    """FeedForward(GEGLU) — motion_module_new.py:441-471, GEGLU :497-518 (exact erf GELU)."""
    h = F.linear(x, sd[f"{p}.net.0.proj.weight"], sd[f"{p}.net.0.proj.bias"])
    val, gate = h.chunk(2, dim=-1)
    h = val * F.gelu(gate)
    return F.linear(h, sd[f"{p}.net.2.weight"], sd[f"{p}.net.2.bias"])


def layer_norm(sd: SD, p: str, x):
    return F.layer_norm(x, (x.shape[-1],), sd[f"{p}.weight"], sd[f"{p}.bias"], 1e-5)


def transformer3d(sd: SD, p: str, x, ctx, heads, groups):
    """Transformer3DModel.forward — attention.py:95-142 with one BasicTransformerBlock (:256-300; SC-attn and
    attn_temp branches are disabled by the NEURONS config, unet.py:89-90)."""
    bsz, c, f, h, w = x.shape
    y = x.permute(0, 2, 1, 3, 4).reshape(bsz * f, c, h, w)
    ctx_r = ctx.repeat_interleave(f, dim=0)                                   # 'b n c -> (b f) n c'  (:100)
    residual = y
    y = F.group_norm(y, groups, sd[f"{p}.norm.weight"], sd[f"{p}.norm.bias"], 1e-6)
    y = F.conv2d(y, sd[f"{p}.proj_in.weight"], sd[f"{p}.proj_in.bias"])
    y = y.permute(0, 2, 3, 1).reshape(bsz * f, h * w, c)
    b = f"{p}.transformer_blocks.0"
    y = cross_attention(sd, f"{b}.attn1", layer_norm(sd, f"{b}.norm1", y), None, heads) + y
    y = cross_attention(sd, f"{b}.attn2", layer_norm(sd, f"{b}.norm2", y), ctx_r, heads) + y
    y = feed_forward(sd, f"{b}.ff", layer_norm(sd, f"{b}.norm3", y)) + y
    y = y.reshape(bsz * f, h, w, c).permute(0, 3, 1, 2).contiguous()
    y = F.conv2d(y, sd[f"{p}.proj_out.weight"], sd[f"{p}.proj_out.bias"])
    y = y + residual
    return y.reshape(bsz, f, c, h, w).permute(0, 2, 1, 3, 4)


def positional_encoding_table(d_model, max_len, device):
    """PositionalEncoding.__init__ — motion_module.py:225-239."""
    position = torch.arange(max_len, device=device).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2, device=device) * (-math.log(10000.0) / d_model))
    pe = torch.zeros(1, max_len, d_model, device=device)
    pe[0, :, 0::2] = torch.sin(position * div_term)
    pe[0, :, 1::2] = torch.cos(position * div_term)
    return pe


def versatile_attention(sd: SD, p: str, x, video_length, heads, pe):
    """VersatileAttention.forward (Temporal mode) — motion_module.py:270-329: regroup "(b f) d c -> (b d) f c",
    add pe[:, :f], self-attention over frames, regroup back."""
    bf, d, c = x.shape
    b = bf // video_length
    y = x.reshape(b, video_length, d, c).permute(0, 2, 1, 3).reshape(b * d, video_length, c)
    y = y + pe[:, :video_length]
    y = cross_attention(sd, p, y, None, heads)
    return y.reshape(b, d, video_length, c).permute(0, 2, 1, 3).reshape(bf, d, c)


def temporal_transformer3d(sd: SD, p0: str, x, heads, groups, nblocks, pe_max_len):
    """VanillaTemporalModule -> TemporalTransformer3DModel.forward — motion_module.py:77-82,134-158, block :210-222."""
    p = f"{p0}.temporal_transformer"
    bsz, c, f, h, w = x.shape
    y = x.permute(0, 2, 1, 3, 4).reshape(bsz * f, c, h, w)
    residual = y
    y = F.group_norm(y, groups, sd[f"{p}.norm.weight"], sd[f"{p}.norm.bias"], 1e-6)
    y = y.permute(0, 2, 3, 1).reshape(bsz * f, h * w, c)
    y = F.linear(y, sd[f"{p}.proj_in.weight"], sd[f"{p}.proj_in.bias"])
    blk = f"{p}.transformer_blocks.0"
    pe = positional_encoding_table(c, pe_max_len, x.device)
    for k in range(nblocks):
        n = layer_norm(sd, f"{blk}.norms.{k}", y)
        y = versatile_attention(sd, f"{blk}.attention_blocks.{k}", n, f, heads, pe) + y
    y = feed_forward(sd, f"{blk}.ff", layer_norm(sd, f"{blk}.ff_norm", y)) + y
    y = F.linear(y, sd[f"{p}.proj_out.weight"], sd[f"{p}.proj_out.bias"])
    y = y.reshape(bsz * f, h, w, c).permute(0, 3, 1, 2).contiguous()
    y = y + residual
    return y.reshape(bsz, f, c, h, w).permute(0, 2, 1, 3, 4)


def upsample3d(sd: SD, p: str, x):
    """Upsample3D.forward — resnet.py:46-80: nearest x(1,2,2) then 3x3 conv."""
    x = F.interpolate(x, scale_factor=[1.0, 2.0, 2.0], mode="nearest")
    return inflated_conv3d(x, sd[f"{p}.conv.weight"], sd[f"{p}.conv.bias"])


def downsample3d(sd: SD, p: str, x):
    """Downsample3D.forward — resnet.py:98-106: 3x3 conv, stride 2, padding 1."""
    return inflated_conv3d(x, sd[f"{p}.conv.weight"], sd[f"{p}.conv.bias"], stride=2, padding=1)


# ------------------------------------------------------------------------------------------------
# blocks and networks
# ------------------------------------------------------------------------------------------------
def _down_blocks(sd: SD, cfg: OracleConfig, x, emb, ctx, taps=None):
    """CrossAttnDownBlock3D.forward / DownBlock3D.forward — unet_blocks.py:382-421,493-521."""
    skips = [x]
    L = len(cfg.block_out_channels)
    for i in range(L):
        bp = f"down_blocks.{i}"
        for j in range(cfg.layers_per_block):
            x = resnet_block3d(sd, f"{bp}.resnets.{j}", x, emb, cfg.norm_num_groups, cfg.norm_eps)
            if taps is not None:
                taps[f"{bp}.resnets.{j}"] = x
            if cfg.down_block_types[i] == "CrossAttnDownBlock3D":
                x = transformer3d(sd, f"{bp}.attentions.{j}", x, ctx, cfg.attention_head_dim, cfg.norm_num_groups)
                if taps is not None:
                    taps[f"{bp}.attentions.{j}"] = x
            if cfg.use_motion_module:
                x = temporal_transformer3d(sd, f"{bp}.motion_modules.{j}", x, cfg.motion_num_heads, cfg.norm_num_groups,
                                           cfg.motion_attention_blocks, cfg.motion_pe_max_len)
                if taps is not None:
                    taps[f"{bp}.motion_modules.{j}"] = x
            skips.append(x)
        if i != L - 1:
            x = downsample3d(sd, f"{bp}.downsamplers.0", x)
            if taps is not None:
                taps[f"{bp}.downsamplers.0"] = x
            skips.append(x)
    return x, skips


def _mid_block(sd: SD, cfg: OracleConfig, x, emb, ctx, taps=None):
    """UNetMidBlock3DCrossAttn.forward — unet_blocks.py:271-278."""
    x = resnet_block3d(sd, "mid_block.resnets.0", x, emb, cfg.norm_num_groups, cfg.norm_eps)
    if taps is not None:
        taps["mid_block.resnets.0"] = x
    x = transformer3d(sd, "mid_block.attentions.0", x, ctx, cfg.attention_head_dim, cfg.norm_num_groups)
    if taps is not None:
        taps["mid_block.attentions.0"] = x
    if cfg.use_motion_module and cfg.motion_module_mid_block:
        x = temporal_transformer3d(sd, "mid_block.motion_modules.0", x, cfg.motion_num_heads, cfg.norm_num_groups,
                                   cfg.motion_attention_blocks, cfg.motion_pe_max_len)
    x = resnet_block3d(sd, "mid_block.resnets.1", x, emb, cfg.norm_num_groups, cfg.norm_eps)
    if taps is not None:
        taps["mid_block.resnets.1"] = x
    return x


def _time_embedding(sd: SD, cfg: OracleConfig, timestep, batch, device):
    """unet.py:371-392 / sparse_controlnet.py:475-502: scalar or per-batch timestep -> (batch, 4*C0)."""
    t = timestep if torch.is_tensor(timestep) else torch.tensor([timestep], device=device)
    t = t.to(device).reshape(-1)
    t = t.expand(batch) if t.numel() == 1 else t
    return timestep_embedding_mlp(sd, timesteps_proj(t, cfg.block_out_channels[0]))


def unet3d_forward(sd: SD, cfg: OracleConfig, sample, timestep, encoder_hidden_states,
                   down_block_additional_residuals: Optional[Sequence[torch.Tensor]] = None,
                   mid_block_additional_residual: Optional[torch.Tensor] = None, taps: Optional[dict] = None):
    """UNet3DConditionModel.forward — animatediff/models/unet.py:357-475."""
    emb = _time_embedding(sd, cfg, timestep, sample.shape[0], sample.device)
    x = inflated_conv3d(sample, sd["conv_in.weight"], sd["conv_in.bias"])
    if taps is not None:
        taps["conv_in"] = x
    x, skips = _down_blocks(sd, cfg, x, emb, encoder_hidden_states, taps)
    if down_block_additional_residuals is not None:                              # unet.py:422-428
        skips = [s + (r.unsqueeze(2) if r.dim() == 4 else r) for s, r in zip(skips, down_block_additional_residuals)]
    x = _mid_block(sd, cfg, x, emb, encoder_hidden_states, taps)
    if mid_block_additional_residual is not None:                                # unet.py:436-439
        r = mid_block_additional_residual
        x = x + (r.unsqueeze(2) if r.dim() == 4 else r)
    L = len(cfg.block_out_channels)
    for i in range(L):                                                           # unet_blocks.py:621-667,735-760
        bp = f"up_blocks.{i}"
        for j in range(cfg.layers_per_block + 1):
            x = torch.cat([x, skips.pop()], dim=1)
            x = resnet_block3d(sd, f"{bp}.resnets.{j}", x, emb, cfg.norm_num_groups, cfg.norm_eps)
            if taps is not None:
                taps[f"{bp}.resnets.{j}"] = x
            if cfg.up_block_types[i] == "CrossAttnUpBlock3D":
                x = transformer3d(sd, f"{bp}.attentions.{j}", x, encoder_hidden_states, cfg.attention_head_dim, cfg.norm_num_groups)
                if taps is not None:
                    taps[f"{bp}.attentions.{j}"] = x
            if cfg.use_motion_module:
                x = temporal_transformer3d(sd, f"{bp}.motion_modules.{j}", x, cfg.motion_num_heads, cfg.norm_num_groups,
                                           cfg.motion_attention_blocks, cfg.motion_pe_max_len)
                if taps is not None:
                    taps[f"{bp}.motion_modules.{j}"] = x
        if i != L - 1:
            x = upsample3d(sd, f"{bp}.upsamplers.0", x)
            if taps is not None:
                taps[f"{bp}.upsamplers.0"] = x
    x = inflated_groupnorm(x, sd["conv_norm_out.weight"], sd["conv_norm_out.bias"], cfg.norm_num_groups, cfg.norm_eps)
    x = F.silu(x)
    return inflated_conv3d(x, sd["conv_out.weight"], sd["conv_out.bias"])


def sparse_controlnet_forward(sd: SD, cfg: OracleConfig, sample, timestep, encoder_hidden_states, controlnet_cond,
                              conditioning_mask, conditioning_scale=1.0, taps=None):
    """SparseControlNetModel.forward — animatediff/models/sparse_controlnet.py:467-581 (simplified condition
    embedding, concatenated mask, no guess mode).  ``taps`` (a dict) receives the activations by reference module name."""
    if cfg.set_noisy_sample_input_to_zero:
        sample = torch.zeros_like(sample)                                        # :468-469
    ctx = encoder_hidden_states.repeat(sample.shape[0] // encoder_hidden_states.shape[0], 1, 1)   # :491
    emb = _time_embedding(sd, cfg, timestep, sample.shape[0], sample.device)
    x = inflated_conv3d(sample, sd["conv_in.weight"], sd["conv_in.bias"])
    cond = torch.cat([controlnet_cond, conditioning_mask], dim=1)               # :517-518
    cond = inflated_conv3d(cond, sd["controlnet_cond_embedding.weight"], sd["controlnet_cond_embedding.bias"])
    reps = x.shape[0] // cond.shape[0]
    x = x + (cond if reps == 1 else cond.repeat(reps, 1, 1, 1, 1))             # :521 (batch-1 broadcast; B>1: SURVEY §8e)
    if taps is not None:
        taps["conv_in"] = x
    x, skips = _down_blocks(sd, cfg, x, emb, ctx, taps)
    x = _mid_block(sd, cfg, x, emb, ctx, taps)
    down = [inflated_conv3d(s, sd[f"controlnet_down_blocks.{i}.weight"], sd[f"controlnet_down_blocks.{i}.bias"], padding=0)
            * conditioning_scale for i, s in enumerate(skips)]                   # :551-566
    mid = inflated_conv3d(x, sd["controlnet_mid_block.weight"], sd["controlnet_mid_block.bias"], padding=0) * conditioning_scale
    return down, mid


# ------------------------------------------------------------------------------------------------
# DDIM + CFG loop (parity UNPINNED for the scheduler arithmetic: diffusers 0.11.1 is not in /root/reference)
# ------------------------------------------------------------------------------------------------
def ddim_alphas_cumprod(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="linear"):
    """configs/inference/inference-v3.yaml:16-21 -> diffusers "linear" betas."""
    if beta_schedule == "linear":
        betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
    elif beta_schedule == "scaled_linear":
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
    else:
        raise NotImplementedError(beta_schedule)
    return torch.cumprod(1.0 - betas, dim=0)


def ddim_timesteps(num_inference_steps, num_train_timesteps=1000, steps_offset=1):
    """DDIMScheduler.set_timesteps: arange(N)*(T//N) reversed, + steps_offset."""
    ratio = num_train_timesteps // num_inference_steps
    return [int(i * ratio) + steps_offset for i in range(num_inference_steps)][::-1]


def ddim_step(eps, t, x, alphas_cumprod, num_inference_steps, num_train_timesteps=1000):
    """DDIMScheduler.step, eta = 0, clip_sample = False (call site pipeline_neuroclips.py:483)."""
    prev_t = t - num_train_timesteps // num_inference_steps
    a_t = alphas_cumprod[t]
    a_prev = alphas_cumprod[prev_t] if prev_t >= 0 else torch.tensor(1.0)
    x0 = (x - (1 - a_t) ** 0.5 * eps) / a_t ** 0.5
    return a_prev ** 0.5 * x0 + (1 - a_prev) ** 0.5 * eps


def add_noise(x, noise, t, alphas_cumprod):
    """DDIMScheduler.add_noise (call site pipeline_neuroclips.py:423)."""
    a = alphas_cumprod[t]
    return a ** 0.5 * x + (1 - a) ** 0.5 * noise


def neuroclips_denoise(unet_sd: SD, unet_cfg: OracleConfig, ctrl_sd: Optional[SD], ctrl_cfg: Optional[OracleConfig],
                       latents, noise, text_embeddings, controlnet_images, controlnet_image_index=(0,),
                       num_inference_steps=50, guidance_scale=8.5, controlnet_conditioning_scale=1.0,
                       return_eps_steps: Sequence[int] = (), x_log: Optional[dict] = None):
    """NeuroclipsPipeline.__call__ denoising part — pipeline_neuroclips.py:377-489, on explicit latents / noise /
    text embeddings (uncond first).  ``latents`` are noised to timesteps[0] and all timesteps run.
    Diagnostics: ``return_eps_steps`` -> raw (pre-CFG) eps of those step indices; ``x_log`` (a dict) receives the latents
    each of those steps started from and, under ``"after"``, the latents after every step."""
    ac = ddim_alphas_cumprod()
    ts = ddim_timesteps(num_inference_steps)
    video_length = latents.shape[2]
    x = add_noise(latents, noise, ts[0], ac)                                     # :410-423
    do_cfg = guidance_scale > 1.0
    cond = mask = None
    if ctrl_sd is not None and controlnet_images is not None:                    # :447-458
        shape = list(controlnet_images.shape)
        shape[2] = video_length
        cond = torch.zeros(shape, device=latents.device)
        mshape = list(shape)
        mshape[1] = 1
        mask = torch.zeros(mshape, device=latents.device)
        idx = list(controlnet_image_index)
        cond[:, :, idx] = controlnet_images[:, :, :len(idx)]
        mask[:, :, idx] = 1
    eps_log = {}
    for i, t in enumerate(ts):
        xin = torch.cat([x] * 2) if do_cfg else x                               # :435
        down = mid = None
        if cond is not None:
            down, mid = sparse_controlnet_forward(ctrl_sd, ctrl_cfg, xin, t, text_embeddings, cond, mask,
                                                  controlnet_conditioning_scale)  # :460-467
        eps = unet3d_forward(unet_sd, unet_cfg, xin, t, text_embeddings, down, mid)   # :470-475
        if i in return_eps_steps:
            eps_log[i] = eps
            if x_log is not None:
                x_log[i] = x
        if do_cfg:                                                               # :478-480
            eu, et = eps.chunk(2)
            eps = eu + guidance_scale * (et - eu)
        x = ddim_step(eps, t, x, ac, num_inference_steps)                        # :483
        if x_log is not None:
            x_log.setdefault("after", []).append(x)
    return x, eps_log
