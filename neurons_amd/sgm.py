"""sgm unCLIP keyframe path — host mirror of the pieces ``utils.unclip_recon`` (utils.py:302-350) drives:

  NativeSGMUNet            ~ sgm.modules.diffusionmodules.openaimodel.UNetModel (:472; forward :816-853) behind
                             OpenAIWrapper (wrappers.py:23-34); every FLOP runs in libneurons_amd.so
  LegacyDDPMDiscretization ~ discretizer.py:42-69 (+ make_beta_schedule util.py:20-33, append_zero sgm/util.py:188)
  DiscreteDenoiser         ~ denoiser.py:42-75 with EpsScaling denoiser_scaling.py:29-37 (host scalars only)
  EulerEDMSampler          ~ sampling.py:41-62,98-135,216-220 with VanillaCFG guiders.py:24-42; the per-element
                             update (c_out/c_skip, CFG, to_d, Euler step) is the HIP kernel nr_edm_cfg_euler_step
  unclip_sample            ~ the sampling part of utils.unclip_recon (:308-340), on explicit z / noise tensors
"""
import ctypes as C
from dataclasses import dataclass
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from .unet3d import _on_device, _NativeNet, _ff_keys


@dataclass
class SGMUNetConfig:
    """generative_models/configs/unclip6.yaml:47-63 (network_config.params)."""
    in_channels: int = 4
    out_channels: int = 4
    model_channels: int = 320
    num_res_blocks: int = 2
    attention_resolutions: Tuple[int, ...] = (4, 2)
    channel_mult: Tuple[int, ...] = (1, 2, 4)
    num_head_channels: int = 64
    transformer_depth: Tuple[int, ...] = (1, 2, 10)
    context_dim: int = 1664
    adm_in_channels: int = 1024
    num_classes: str = "sequential"
    use_linear_in_transformer: bool = True
    norm_num_groups: int = 32        # GroupNorm32 (diffusionmodules/util.py:274-276), eps 1e-5


def _sgm_levels(cfg: SGMUNetConfig):
    chans = [m * cfg.model_channels for m in cfg.channel_mult]
    attn = [(2 ** i) in cfg.attention_resolutions for i in range(len(chans))]
    return chans, attn


def sgm_c_config(cfg: SGMUNetConfig) -> _lib.NrNetConfig:
    if cfg.num_classes != "sequential" or not cfg.use_linear_in_transformer:
        raise NotImplementedError("only the unclip6.yaml variant (num_classes='sequential', linear proj in/out) is built")
    if len(cfg.channel_mult) > _lib.NR_MAX_LEVELS:
        raise ValueError("too many levels")
    chans, attn = _sgm_levels(cfg)
    c = _lib.NrNetConfig()
    c.kind = _lib.NR_KIND_SGM_UNET
    c.in_channels, c.out_channels = cfg.in_channels, cfg.out_channels
    c.num_levels = len(chans)
    depth = list(cfg.transformer_depth) if not isinstance(cfg.transformer_depth, int) else [cfg.transformer_depth] * len(chans)
    for i, ch in enumerate(chans):
        c.block_out_channels[i] = ch
        c.down_block_has_attn[i] = 1 if attn[i] else 0
        c.up_block_has_attn[i] = 1 if attn[i] else 0
        c.transformer_depth[i] = depth[i]
    c.layers_per_block = cfg.num_res_blocks
    c.num_heads = 0
    c.num_head_channels = cfg.num_head_channels
    c.cross_attention_dim = cfg.context_dim
    c.norm_num_groups = cfg.norm_num_groups
    c.norm_eps = 1e-5
    c.adm_in_channels = cfg.adm_in_channels
    return c


def _sgm_res_keys(p, cin, cout, temb):
    k = {f"{p}.in_layers.0.weight": (cin,), f"{p}.in_layers.0.bias": (cin,), f"{p}.in_layers.2.weight": (cout, cin, 3, 3),
         f"{p}.in_layers.2.bias": (cout,), f"{p}.emb_layers.1.weight": (cout, temb), f"{p}.emb_layers.1.bias": (cout,),
         f"{p}.out_layers.0.weight": (cout,), f"{p}.out_layers.0.bias": (cout,), f"{p}.out_layers.3.weight": (cout, cout, 3, 3),
         f"{p}.out_layers.3.bias": (cout,)}
    if cin != cout:
        k[f"{p}.skip_connection.weight"] = (cout, cin, 1, 1)
        k[f"{p}.skip_connection.bias"] = (cout,)
    return k


def _sgm_st_keys(p, c, ctx, depth):
    k = {f"{p}.norm.weight": (c,), f"{p}.norm.bias": (c,), f"{p}.proj_in.weight": (c, c), f"{p}.proj_in.bias": (c,),
         f"{p}.proj_out.weight": (c, c), f"{p}.proj_out.bias": (c,)}
    for d in range(depth):
        b = f"{p}.transformer_blocks.{d}"
        for n in ("norm1", "norm2", "norm3"):
            k[f"{b}.{n}.weight"] = (c,)
            k[f"{b}.{n}.bias"] = (c,)
        for a, kd in (("attn1", c), ("attn2", ctx)):
            k[f"{b}.{a}.to_q.weight"] = (c, c)
            k[f"{b}.{a}.to_k.weight"] = (c, kd)
            k[f"{b}.{a}.to_v.weight"] = (c, kd)
            k[f"{b}.{a}.to_out.0.weight"] = (c, c)
            k[f"{b}.{a}.to_out.0.bias"] = (c,)
        k.update(_ff_keys(f"{b}.ff", c))
    return k


def sgm_state_dict_schema(cfg: SGMUNetConfig) -> Dict[str, tuple]:
    """Parameter names / shapes of the reference ``UNetModel.state_dict()`` (construction order openaimodel.py:640-813)."""
    chans, attn = _sgm_levels(cfg)
    L = len(chans)
    mc, temb, ctx = cfg.model_channels, 4 * cfg.model_channels, cfg.context_dim
    depth = list(cfg.transformer_depth)
    k = {"time_embed.0.weight": (temb, mc), "time_embed.0.bias": (temb,), "time_embed.2.weight": (temb, temb), "time_embed.2.bias": (temb,),
         "label_emb.0.0.weight": (temb, cfg.adm_in_channels), "label_emb.0.0.bias": (temb,),
         "label_emb.0.2.weight": (temb, temb), "label_emb.0.2.bias": (temb,),
         "input_blocks.0.0.weight": (mc, cfg.in_channels, 3, 3), "input_blocks.0.0.bias": (mc,)}
    idx, ch = 1, mc
    skip_chans = [mc]
    for lev in range(L):
        for _ in range(cfg.num_res_blocks):
            k.update(_sgm_res_keys(f"input_blocks.{idx}.0", ch, chans[lev], temb))
            ch = chans[lev]
            if attn[lev]:
                k.update(_sgm_st_keys(f"input_blocks.{idx}.1", ch, ctx, depth[lev]))
            skip_chans.append(ch)
            idx += 1
        if lev != L - 1:
            k[f"input_blocks.{idx}.0.op.weight"] = (ch, ch, 3, 3)
            k[f"input_blocks.{idx}.0.op.bias"] = (ch,)
            skip_chans.append(ch)
            idx += 1
    k.update(_sgm_res_keys("middle_block.0", ch, ch, temb))
    k.update(_sgm_st_keys("middle_block.1", ch, ctx, depth[-1]))
    k.update(_sgm_res_keys("middle_block.2", ch, ch, temb))
    idx = 0
    for lev in reversed(range(L)):
        for i in range(cfg.num_res_blocks + 1):
            ich = skip_chans.pop()
            k.update(_sgm_res_keys(f"output_blocks.{idx}.0", ch + ich, chans[lev], temb))
            ch = chans[lev]
            sub = 1
            if attn[lev]:
                k.update(_sgm_st_keys(f"output_blocks.{idx}.1", ch, ctx, depth[lev]))
                sub = 2
            if lev and i == cfg.num_res_blocks:
                k[f"output_blocks.{idx}.{sub}.conv.weight"] = (ch, ch, 3, 3)
                k[f"output_blocks.{idx}.{sub}.conv.bias"] = (ch,)
            idx += 1
    k["out.0.weight"] = (ch,)
    k["out.0.bias"] = (ch,)
    k["out.2.weight"] = (cfg.out_channels, mc, 3, 3)
    k["out.2.bias"] = (cfg.out_channels,)
    return k


def sgm_random_state_dict(cfg: SGMUNetConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Seeded synthetic weights (zero-initialised layers of the reference — out conv, proj_out, ResBlock out conv —
    are randomised so every path is numerically visible)."""
    from .synth import randn
    sd = {}
    for name, shape in sgm_state_dict_schema(cfg).items():
        z = randn(name, shape, seed)
        if name.endswith(".bias"):
            is_norm = ".norm" in name or "in_layers.0" in name or "out_layers.0" in name or name.startswith("out.0")
            t = (0.1 if is_norm else 0.02) * z
        elif len(shape) == 1:
            t = 1.0 + 0.1 * z
        else:
            t = z / (int(np.prod(shape[1:])) ** 0.5)
        sd[name] = t
    return sd


class NativeSGMUNet(_NativeNet):
    """``network(x * c_in, c_noise, cond)`` of denoiser.py:36-39: call as ``net(x, timesteps, context=..., y=...)``."""
    _kind = _lib.NR_KIND_SGM_UNET
    _config_cls = SGMUNetConfig

    def _build_cconf(self, config):
        return sgm_c_config(config)

    def _build_schema(self, config):
        return sgm_state_dict_schema(config)

    def _on_plan(self):
        b, f, h, w, L = self._plan_key
        dev = self.device
        self._io_x = torch.empty(b, self.config.in_channels, h, w, dtype=torch.float32, device=dev)
        self._io_ctx = torch.empty(b, L, self.config.context_dim, dtype=torch.float32, device=dev)
        self._io_y = torch.empty(b, self.config.adm_in_channels, dtype=torch.float32, device=dev)
        self._io_out = torch.empty(b, self.config.out_channels, h, w, dtype=torch.float32, device=dev)

    @_on_device
    def forward(self, x, timesteps=None, context=None, y=None, in_scale: float = 1.0, **kwargs):
        if kwargs:
            raise NotImplementedError(f"unsupported arguments {sorted(kwargs)}")
        if (y is None) or (context is None) or (timesteps is None):
            raise AssertionError("must specify y if and only if the model is class-conditional")   # openaimodel.py:832-834
        if not x.is_cuda:
            raise RuntimeError("NativeSGMUNet.forward: CUDA (ROCm) tensors required; there is no CPU fallback")
        b, c, h, w = x.shape
        if y.shape[0] != b or context.shape[0] != b:
            raise AssertionError("batch mismatch")                                                    # :840
        self._ensure_plan(b, 1, h, w, context.shape[1])
        self._io_x.copy_(x)
        self._set_context(context)
        self._io_y.copy_(y)
        ts = self._timesteps_host(timesteps, b)
        lib = _lib.load()
        _lib.check(lib.nr_sgm_unet_forward(self._h, torch.cuda.current_stream().cuda_stream, self._io_x.data_ptr(),
                                           float(in_scale), ts, self._io_ctx.data_ptr(), context.shape[1],
                                           self._io_y.data_ptr(), self._io_out.data_ptr()))
        return self._io_out.clone()

    __call__ = forward


# ---------------------------------------------------------------------------------------------------------
# sampler-side host logic (scalars / tables only)
# ---------------------------------------------------------------------------------------------------------
class LegacyDDPMDiscretization:
    def __init__(self, linear_start=0.00085, linear_end=0.0120, num_timesteps=1000):
        self.num_timesteps = num_timesteps
        # make_beta_schedule("linear"): linspace in sqrt(beta), float64 (diffusionmodules/util.py:20-33) — SURVEY F12
        betas = np.linspace(linear_start ** 0.5, linear_end ** 0.5, num_timesteps, dtype=np.float64) ** 2
        self.alphas_cumprod = np.cumprod(1.0 - betas, axis=0)

    def get_sigmas(self, n):
        if n < self.num_timesteps:
            timesteps = np.linspace(self.num_timesteps - 1, 0, n, endpoint=False).astype(int)[::-1]   # discretizer.py:11-14
            ac = self.alphas_cumprod[timesteps]
        elif n == self.num_timesteps:
            ac = self.alphas_cumprod
        else:
            raise ValueError
        sigmas = torch.tensor((1 - ac) / ac, dtype=torch.float32) ** 0.5
        return torch.flip(sigmas, (0,))

    def __call__(self, n, do_append_zero=True, device="cpu", flip=False):
        sigmas = self.get_sigmas(n)
        if do_append_zero:
            sigmas = torch.cat([sigmas, sigmas.new_zeros([1])])                                       # sgm/util.py:188-189
        sigmas = sigmas if not flip else torch.flip(sigmas, (0,))
        return sigmas.to(device)


class DiscreteDenoiser:
    """EpsScaling + sigma quantisation: the scalars the network call needs (denoiser.py:23-39,61-75)."""

    def __init__(self, num_idx=1000, discretization: Optional[LegacyDDPMDiscretization] = None):
        self.discretization = discretization or LegacyDDPMDiscretization()
        self.sigmas = self.discretization(num_idx, do_append_zero=False, flip=True)     # ascending table
        self.num_idx = num_idx

    def sigma_to_idx(self, sigma: float) -> int:
        return int((torch.as_tensor(sigma, dtype=torch.float32) - self.sigmas).abs().argmin())

    def scalars(self, sigma: float):
        """-> (sigma_quantised, c_in, c_noise index)"""
        idx = self.sigma_to_idx(sigma)
        sq = float(self.sigmas[idx])
        c_in = float(1.0 / (torch.tensor(sq, dtype=torch.float32) ** 2 + 1.0) ** 0.5)
        return sq, c_in, idx


class EulerEDMSampler:
    """EulerEDMSampler(s_churn=0) + VanillaCFG on the native network."""

    def __init__(self, num_steps=38, scale=5.0, discretization: Optional[LegacyDDPMDiscretization] = None):
        self.num_steps = num_steps
        self.scale = scale
        self.discretization = discretization or LegacyDDPMDiscretization()
        self.denoiser = DiscreteDenoiser(discretization=self.discretization)

    def __call__(self, network: NativeSGMUNet, x, cond: Dict[str, torch.Tensor], uc: Optional[Dict[str, torch.Tensor]] = None,
                 num_steps: Optional[int] = None):
        if not x.is_cuda:
            raise RuntimeError("EulerEDMSampler runs the HIP kernels: CUDA (ROCm) tensors required; there is no CPU fallback")
        uc = cond if uc is None else uc
        sigmas = self.discretization(self.num_steps if num_steps is None else num_steps)
        x = (x * torch.sqrt(1.0 + sigmas[0] ** 2.0).to(x.device)).to(torch.float32).contiguous()       # sampling.py:52
        ctx = torch.cat((uc["crossattn"], cond["crossattn"]), 0)                                     # guiders.py:38 (uc first)
        vec = torch.cat((uc["vector"], cond["vector"]), 0)
        lib = _lib.load()
        n = x.numel()
        for i in range(len(sigmas) - 1):
            s, s_next = float(sigmas[i]), float(sigmas[i + 1])
            sq, c_in, idx = self.denoiser.scalars(s)
            xin = torch.cat([x] * 2)                                                                  # guiders.py:42
            net = network(xin, float(idx), context=ctx, y=vec, in_scale=c_in)
            x_new = torch.empty_like(x)
            _lib.check(lib.nr_edm_cfg_euler_step(torch.cuda.current_stream().cuda_stream, net.data_ptr(), x.data_ptr(),
                                                 x_new.data_ptr(), n, float(self.scale), sq, s, s_next))
            x = x_new
        return x


def unclip_sample(network: NativeSGMUNet, tokens, vector_suffix, z, noise, uc_tokens, sampler: EulerEDMSampler,
                  offset_noise: Optional[torch.Tensor] = None, offset_noise_level: float = 0.04):
    """Sampling part of ``utils.unclip_recon`` (utils.py:308-340) on explicit tensors: ``tokens`` (1,256,1664) prior
    tokens x key-object mask, ``vector_suffix`` (1,1024), starting ``z`` and ``noise`` (n,4,h,w), ``uc_tokens`` the
    random unconditional tokens (:318), ``offset_noise`` (n,) the per-sample offset draw (:328-331)."""
    n = z.shape[0]
    c = {"crossattn": tokens.repeat(n, 1, 1).to(z.device), "vector": vector_suffix.repeat(n, 1).to(z.device)}
    uc = {"crossattn": uc_tokens.repeat(n, 1, 1).to(z.device), "vector": vector_suffix.repeat(n, 1).to(z.device)}
    sigmas = sampler.discretization(sampler.num_steps)
    sigma = sigmas[0].to(z.device)
    if offset_noise_level > 0.0 and offset_noise is not None:
        noise = noise + offset_noise_level * offset_noise.to(z.device).reshape(-1, 1, 1, 1)
    noised_z = z + noise * sigma
    noised_z = noised_z / torch.sqrt(1.0 + sigmas[0] ** 2.0).to(z.device)
    return sampler(network, noised_z, cond=c, uc=uc)
