"""First-stage (VAE) decoder — host mirror of what turns denoised latents into pixels on both reference paths:

  keyframes : ``DiffusionEngine.decode_first_stage`` (generative_models/sgm/models/diffusion.py:118-135) ->
              ``AutoencodingEngineLegacy.decode`` (sgm/models/autoencoder.py:490-494) = ``post_quant_conv`` ->
              ``Decoder.forward`` (sgm/modules/diffusionmodules/model.py:723-757), called by utils.unclip_recon (:343)
  video     : ``AnimationPipeline.decode_latents`` (animatediff/pipelines/pipeline_animation.py:243-256): per-frame
              ``vae.decode(latents / 0.18215).sample`` with diffusers ``AutoencoderKL`` — the same network under
              diffusers parameter names (``diffusers_vae_key_map`` is the inverse of the reference's
              ``convert_ldm_vae_checkpoint``, animatediff/utils/convert_from_ckpt.py:559-663, decoder half)

Every FLOP runs in libneurons_amd.so (kind NR_KIND_VAE_DECODER); there is no CPU fallback.
"""
import ctypes as C
from dataclasses import dataclass
from typing import Dict, Tuple

import numpy as np
import torch

from . import _lib
from .unet3d import _NativeNet


@dataclass
class VAEDecoderConfig:
    """generative_models/configs/unclip6.yaml:99-115 (first_stage_config.params.ddconfig) == SD-1.5 vae/config.json."""
    embed_dim: int = 4
    z_channels: int = 4
    out_ch: int = 3
    ch: int = 128
    ch_mult: Tuple[int, ...] = (1, 2, 4, 4)
    num_res_blocks: int = 2
    attn_resolutions: Tuple[int, ...] = ()
    attn_type: str = "vanilla"
    norm_num_groups: int = 32         # Normalize(): GroupNorm(32, eps=1e-6)  (model.py:52-55)
    in_channels: int = 4              # latent channels (for the module-like surface)


def vae_c_config(cfg: VAEDecoderConfig) -> _lib.NrNetConfig:
    if len(cfg.attn_resolutions) != 0 or cfg.attn_type not in ("vanilla", "vanilla-xformers"):
        raise NotImplementedError("only the SD VAE variant (mid-block attention only, vanilla attention) is built")
    if cfg.embed_dim != cfg.z_channels:
        raise NotImplementedError("embed_dim != z_channels")
    if len(cfg.ch_mult) > _lib.NR_MAX_LEVELS:
        raise ValueError("too many levels")
    c = _lib.NrNetConfig()
    c.kind = _lib.NR_KIND_VAE_DECODER
    c.in_channels, c.out_channels = cfg.z_channels, cfg.out_ch
    c.num_levels = len(cfg.ch_mult)
    for i, m in enumerate(cfg.ch_mult):
        c.block_out_channels[i] = cfg.ch * m
    c.layers_per_block = cfg.num_res_blocks
    c.num_heads = 1
    c.norm_num_groups = cfg.norm_num_groups
    c.norm_eps = 1e-6
    return c


def _res_keys(p, cin, cout):
    k = {f"{p}.norm1.weight": (cin,), f"{p}.norm1.bias": (cin,), f"{p}.conv1.weight": (cout, cin, 3, 3), f"{p}.conv1.bias": (cout,),
         f"{p}.norm2.weight": (cout,), f"{p}.norm2.bias": (cout,), f"{p}.conv2.weight": (cout, cout, 3, 3), f"{p}.conv2.bias": (cout,)}
    if cin != cout:
        k[f"{p}.nin_shortcut.weight"] = (cout, cin, 1, 1)
        k[f"{p}.nin_shortcut.bias"] = (cout,)
    return k


def vae_decoder_state_dict_schema(cfg: VAEDecoderConfig) -> Dict[str, tuple]:
    """Decoder-side parameter names / shapes of the reference first-stage ``state_dict()`` (post_quant_conv +
    ``decoder.*``; construction order model.py:658-709)."""
    chans = [cfg.ch * m for m in cfg.ch_mult]
    L = len(chans)
    cm = chans[-1]
    k = {"post_quant_conv.weight": (cfg.z_channels, cfg.embed_dim, 1, 1), "post_quant_conv.bias": (cfg.z_channels,),
         "decoder.conv_in.weight": (cm, cfg.z_channels, 3, 3), "decoder.conv_in.bias": (cm,)}
    k.update(_res_keys("decoder.mid.block_1", cm, cm))
    a = "decoder.mid.attn_1"
    k[f"{a}.norm.weight"] = (cm,)
    k[f"{a}.norm.bias"] = (cm,)
    for n in ("q", "k", "v", "proj_out"):
        k[f"{a}.{n}.weight"] = (cm, cm, 1, 1)
        k[f"{a}.{n}.bias"] = (cm,)
    k.update(_res_keys("decoder.mid.block_2", cm, cm))
    ch = cm
    for lev in reversed(range(L)):
        for j in range(cfg.num_res_blocks + 1):
            k.update(_res_keys(f"decoder.up.{lev}.block.{j}", ch, chans[lev]))
            ch = chans[lev]
        if lev != 0:
            k[f"decoder.up.{lev}.upsample.conv.weight"] = (ch, ch, 3, 3)
            k[f"decoder.up.{lev}.upsample.conv.bias"] = (ch,)
    k["decoder.norm_out.weight"] = (ch,)
    k["decoder.norm_out.bias"] = (ch,)
    k["decoder.conv_out.weight"] = (cfg.out_ch, ch, 3, 3)
    k["decoder.conv_out.bias"] = (cfg.out_ch,)
    return k


def vae_random_state_dict(cfg: VAEDecoderConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    from .synth import randn
    sd = {}
    for name, shape in vae_decoder_state_dict_schema(cfg).items():
        z = randn(name, shape, seed)
        if name.endswith(".bias"):
            t = (0.1 if ".norm" in name else 0.02) * z
        elif len(shape) == 1:
            t = 1.0 + 0.1 * z
        else:
            t = z / (int(np.prod(shape[1:])) ** 0.5)
        sd[name] = t
    return sd


def diffusers_vae_key_map(cfg: VAEDecoderConfig) -> Dict[str, str]:
    """diffusers ``AutoencoderKL`` decoder parameter name -> first-stage (LDM) name.  Inverse of the decoder half of
    ``convert_ldm_vae_checkpoint`` (convert_from_ckpt.py:559-663): up_blocks are numbered from the lowest
    resolution (``up_blocks.i`` = ``decoder.up.{L-1-i}``), the mid attention uses query/key/value/proj_attn Linear
    weights ([C][C], the 1x1 convs squeezed by conv_attn_to_linear :203-212), shortcuts are ``conv_shortcut``."""
    L = len(cfg.ch_mult)
    m = {"post_quant_conv": "post_quant_conv", "decoder.conv_in": "decoder.conv_in", "decoder.conv_norm_out": "decoder.norm_out",
         "decoder.conv_out": "decoder.conv_out", "decoder.mid_block.resnets.0": "decoder.mid.block_1",
         "decoder.mid_block.resnets.1": "decoder.mid.block_2", "decoder.mid_block.attentions.0.group_norm": "decoder.mid.attn_1.norm",
         "decoder.mid_block.attentions.0.query": "decoder.mid.attn_1.q", "decoder.mid_block.attentions.0.key": "decoder.mid.attn_1.k",
         "decoder.mid_block.attentions.0.value": "decoder.mid.attn_1.v", "decoder.mid_block.attentions.0.proj_attn": "decoder.mid.attn_1.proj_out"}
    for i in range(L):
        lev = L - 1 - i
        for j in range(cfg.num_res_blocks + 1):
            m[f"decoder.up_blocks.{i}.resnets.{j}"] = f"decoder.up.{lev}.block.{j}"
        if lev != 0:
            m[f"decoder.up_blocks.{i}.upsamplers.0.conv"] = f"decoder.up.{lev}.upsample.conv"
    out = {}
    sch = vae_decoder_state_dict_schema(cfg)
    inv = sorted(m.items(), key=lambda kv: -len(kv[1]))
    for key in sch:
        for new, old in inv:
            if key == old or key.startswith(old + "."):
                out[new + key[len(old):].replace("nin_shortcut", "conv_shortcut")] = key
                break
        else:
            raise KeyError(key)
    return out


def convert_diffusers_vae_state_dict(sd: Dict[str, torch.Tensor], cfg: VAEDecoderConfig) -> Dict[str, torch.Tensor]:
    """``AutoencoderKL.state_dict()`` (diffusers names) -> the decoder-side first-stage names this module loads."""
    km = diffusers_vae_key_map(cfg)
    sch = vae_decoder_state_dict_schema(cfg)
    out = {}
    for k, v in sd.items():
        if k in km:
            out[km[k]] = v.reshape(sch[km[k]])
    return out


class NativeVAEDecoder(_NativeNet):
    """``first_stage_model.decode`` / ``vae.decode(...).sample``.  ``load_state_dict`` takes first-stage names
    (``post_quant_conv.*``, ``decoder.*``); encoder / quant_conv / loss entries are ignored (strict=False)."""
    _kind = _lib.NR_KIND_VAE_DECODER
    _config_cls = VAEDecoderConfig
    MAX_IMAGES = 16      # images per engine launch (the plan's activation arena grows with it)

    def _build_cconf(self, config):
        return vae_c_config(config)

    def _build_schema(self, config):
        return vae_decoder_state_dict_schema(config)

    def _on_plan(self):
        b, f, h, w, L = self._plan_key
        dev = torch.device("cuda", torch.cuda.current_device())
        self.device = dev
        up = 2 ** (len(self.config.ch_mult) - 1)
        self._io_z = torch.empty(b, self.config.z_channels, h, w, dtype=torch.float32, device=dev)
        self._out_shape = (self.config.out_ch, h * up, w * up)

    def decode(self, z, z_scale: float = 1.0, unit_range: bool = False, chunk: int = None, post=None):
        """z [n][4][h][w] (any float dtype, on the GPU) -> fp32 [n][3][8h][8w].  ``z_scale`` multiplies the latent
        first (1 / scale_factor); ``post=(mul, add)`` fuses ``clamp(x * mul + add, 0, 1)`` into the last kernel
        (``unit_range`` = (0.5, 0.5)).  Images are independent (GroupNorm is per sample), so ``chunk`` only bounds
        the activation arena."""
        if unit_range:
            post = (0.5, 0.5)
        mul, add, clamp = (float(post[0]), float(post[1]), 1) if post is not None else (1.0, 0.0, 0)
        if not z.is_cuda:
            raise RuntimeError("NativeVAEDecoder.decode: CUDA (ROCm) tensors required; there is no CPU fallback")
        if z.dim() != 4 or z.shape[1] != self.config.z_channels:
            raise ValueError(f"expected [n][{self.config.z_channels}][h][w] latents, got {tuple(z.shape)}")
        n, _, h, w = z.shape
        chunk = min(n, chunk or self.MAX_IMAGES, self.MAX_IMAGES)
        lib = _lib.load()
        stream = torch.cuda.current_stream().cuda_stream
        outs = []
        for i in range(0, n, chunk):
            zc = z[i:i + chunk]
            self._ensure_plan(zc.shape[0], 1, h, w, 0)
            self._io_z.copy_(zc)
            out = torch.empty((zc.shape[0],) + self._out_shape, dtype=torch.float32, device=z.device)
            _lib.check(lib.nr_vae_decode(self._handle(), stream, self._io_z.data_ptr(), float(z_scale), mul, add, clamp,
                                         out.data_ptr()))
            outs.append(out)
        return outs[0] if len(outs) == 1 else torch.cat(outs)

    __call__ = decode

    # -- the two reference call sites ----------------------------------------------------------------------
    def decode_first_stage(self, z, scale_factor: float = 0.18215):
        """sgm/models/diffusion.py:118-135."""
        return self.decode(z, z_scale=1.0 / scale_factor)

    def decode_keyframe(self, samples_z, scale_factor: float = 0.18215):
        """The tail of utils.unclip_recon (:343-349): decode_first_stage then ``clamp(x * .8 + .2, 0, 1)``."""
        return self.decode(samples_z, z_scale=1.0 / scale_factor, post=(0.8, 0.2))

    def decode_latents(self, latents):
        """pipeline_animation.py:243-256: (b, c, f, h, w) latents -> (b, 3, f, 8h, 8w) video in [0, 1] (kept on the GPU;
        the reference's trailing ``.cpu().float().numpy()`` is the caller's)."""
        b, c, f, h, w = latents.shape
        frames = latents.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)
        video = self.decode(frames, z_scale=1.0 / 0.18215, unit_range=True)
        return video.reshape(b, f, *video.shape[1:]).permute(0, 2, 1, 3, 4)
