"""First-stage (VAE) decoder — host mirror of what turns denoised latents into pixels on both reference paths:

  keyframes : ``DiffusionEngine.decode_first_stage`` (generative_models/sgm/models/diffusion.py:118-135) ->
              ``AutoencodingEngineLegacy.decode`` (sgm/models/autoencoder.py:490-494) = ``post_quant_conv`` ->
              ``Decoder.forward`` (sgm/modules/diffusionmodules/model.py:723-757), called by utils.unclip_recon (:343)
  video     : ``AnimationPipeline.decode_latents`` (animatediff/pipelines/pipeline_animation.py:243-256): per-frame
              ``vae.decode(latents / 0.18215).sample`` with diffusers ``AutoencoderKL`` — the same network under
              diffusers parameter names (``diffusers_vae_key_map`` is the inverse of the reference's
              ``convert_ldm_vae_checkpoint``, animatediff/utils/convert_from_ckpt.py:559-663, decoder half)

  encode    : ``vae.encode(2 * x - 1).latent_dist.sample() * 0.18215`` (scripts/neuroclips_video.py:267,282) =
              ``Encoder.forward`` (model.py:584-609) -> ``quant_conv`` -> ``DiagonalGaussianDistribution``
              (sgm/modules/distributions/distributions.py:24-42)

Every FLOP runs in libneurons_amd.so (kinds NR_KIND_VAE_DECODER / NR_KIND_VAE_ENCODER); there is no CPU fallback.
``NativeAutoencoderKL`` bundles both behind the diffusers call surface the scripts use (``.encode(x).latent_dist``,
``.decode(z).sample``).
"""
import ctypes as C
from dataclasses import dataclass
from typing import Dict, Tuple

import numpy as np
import torch

from . import _lib
from .unet3d import _on_device, _NativeNet


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


def vae_encoder_state_dict_schema(cfg: VAEDecoderConfig) -> Dict[str, tuple]:
    """Encoder-side names / shapes (``encoder.*`` + quant_conv; construction order model.py:523-583)."""
    chans = [cfg.ch * m for m in cfg.ch_mult]
    L = len(chans)
    k = {"encoder.conv_in.weight": (cfg.ch, 3, 3, 3), "encoder.conv_in.bias": (cfg.ch,)}
    ch = cfg.ch
    for lev in range(L):
        for j in range(cfg.num_res_blocks):
            k.update(_res_keys(f"encoder.down.{lev}.block.{j}", ch, chans[lev]))
            ch = chans[lev]
        if lev != L - 1:
            k[f"encoder.down.{lev}.downsample.conv.weight"] = (ch, ch, 3, 3)
            k[f"encoder.down.{lev}.downsample.conv.bias"] = (ch,)
    k.update(_res_keys("encoder.mid.block_1", ch, ch))
    a = "encoder.mid.attn_1"
    k[f"{a}.norm.weight"] = (ch,)
    k[f"{a}.norm.bias"] = (ch,)
    for n in ("q", "k", "v", "proj_out"):
        k[f"{a}.{n}.weight"] = (ch, ch, 1, 1)
        k[f"{a}.{n}.bias"] = (ch,)
    k.update(_res_keys("encoder.mid.block_2", ch, ch))
    k["encoder.norm_out.weight"] = (ch,)
    k["encoder.norm_out.bias"] = (ch,)
    k["encoder.conv_out.weight"] = (2 * cfg.z_channels, ch, 3, 3)
    k["encoder.conv_out.bias"] = (2 * cfg.z_channels,)
    k["quant_conv.weight"] = (2 * cfg.embed_dim, 2 * cfg.z_channels, 1, 1)
    k["quant_conv.bias"] = (2 * cfg.embed_dim,)
    return k


def vae_random_state_dict(cfg: VAEDecoderConfig, seed: int = 0, encoder: bool = False) -> Dict[str, torch.Tensor]:
    from .synth import randn
    sd = {}
    for name, shape in (vae_encoder_state_dict_schema(cfg) if encoder else vae_decoder_state_dict_schema(cfg)).items():
        z = randn(name, shape, seed)
        if name.endswith(".bias"):
            t = (0.1 if ".norm" in name else 0.02) * z
        elif len(shape) == 1:
            t = 1.0 + 0.1 * z
        else:
            t = z / (int(np.prod(shape[1:])) ** 0.5)
        sd[name] = t
    return sd


def diffusers_vae_key_map(cfg: VAEDecoderConfig, encoder: bool = False) -> Dict[str, str]:
    """diffusers ``AutoencoderKL`` parameter name -> first-stage (LDM) name, decoder half (default) or encoder half.
    Inverse of ``convert_ldm_vae_checkpoint`` (convert_from_ckpt.py:559-663): up_blocks are numbered from the lowest
    resolution (``up_blocks.i`` = ``decoder.up.{L-1-i}``), down_blocks in order, the mid attention uses
    query/key/value/proj_attn Linear weights ([C][C], the 1x1 convs squeezed by conv_attn_to_linear :203-212),
    shortcuts are ``conv_shortcut``."""
    L = len(cfg.ch_mult)
    side = "encoder" if encoder else "decoder"
    m = {f"{side}.conv_in": f"{side}.conv_in", f"{side}.conv_norm_out": f"{side}.norm_out", f"{side}.conv_out": f"{side}.conv_out",
         f"{side}.mid_block.resnets.0": f"{side}.mid.block_1", f"{side}.mid_block.resnets.1": f"{side}.mid.block_2",
         f"{side}.mid_block.attentions.0.group_norm": f"{side}.mid.attn_1.norm", f"{side}.mid_block.attentions.0.query": f"{side}.mid.attn_1.q",
         f"{side}.mid_block.attentions.0.key": f"{side}.mid.attn_1.k", f"{side}.mid_block.attentions.0.value": f"{side}.mid.attn_1.v",
         f"{side}.mid_block.attentions.0.proj_attn": f"{side}.mid.attn_1.proj_out"}
    if encoder:
        m["quant_conv"] = "quant_conv"
        for i in range(L):
            for j in range(cfg.num_res_blocks):
                m[f"encoder.down_blocks.{i}.resnets.{j}"] = f"encoder.down.{i}.block.{j}"
            if i != L - 1:
                m[f"encoder.down_blocks.{i}.downsamplers.0.conv"] = f"encoder.down.{i}.downsample.conv"
    else:
        m["post_quant_conv"] = "post_quant_conv"
        for i in range(L):
            lev = L - 1 - i
            for j in range(cfg.num_res_blocks + 1):
                m[f"decoder.up_blocks.{i}.resnets.{j}"] = f"decoder.up.{lev}.block.{j}"
            if lev != 0:
                m[f"decoder.up_blocks.{i}.upsamplers.0.conv"] = f"decoder.up.{lev}.upsample.conv"
    out = {}
    sch = vae_encoder_state_dict_schema(cfg) if encoder else vae_decoder_state_dict_schema(cfg)
    inv = sorted(m.items(), key=lambda kv: -len(kv[1]))
    for key in sch:
        for new, old in inv:
            if key == old or key.startswith(old + "."):
                out[new + key[len(old):].replace("nin_shortcut", "conv_shortcut")] = key
                break
        else:
            raise KeyError(key)
    return out


def convert_diffusers_vae_state_dict(sd: Dict[str, torch.Tensor], cfg: VAEDecoderConfig, encoder: bool = False) -> Dict[str, torch.Tensor]:
    """``AutoencoderKL.state_dict()`` (diffusers names) -> the first-stage names of the chosen half."""
    km = diffusers_vae_key_map(cfg, encoder)
    sch = vae_encoder_state_dict_schema(cfg) if encoder else vae_decoder_state_dict_schema(cfg)
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
        dev = self.device
        up = 2 ** (len(self.config.ch_mult) - 1)
        self._io_z = torch.empty(b, self.config.z_channels, h, w, dtype=torch.float32, device=dev)
        self._out_shape = (self.config.out_ch, h * up, w * up)

    @_on_device
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


def vae_encoder_c_config(cfg: VAEDecoderConfig) -> _lib.NrNetConfig:
    c = vae_c_config(cfg)
    c.kind = _lib.NR_KIND_VAE_ENCODER
    c.in_channels, c.out_channels = 3, 2 * cfg.z_channels
    return c


class DiagonalGaussianDistribution:
    """sgm/modules/distributions/distributions.py:24-72 on the GPU-resident moments of the native encoder."""

    def __init__(self, parameters):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)

    def _draw(self, noise, scale):
        n, c2, h, w = self.parameters.shape
        out = torch.empty(n, c2 // 2, h, w, dtype=torch.float32, device=self.parameters.device)
        _lib.check(_lib.load().nr_gaussian_sample(torch.cuda.current_stream().cuda_stream, self.parameters.data_ptr(),
                                                  noise.data_ptr() if noise is not None else None, out.data_ptr(), n, c2 // 2, h * w,
                                                  float(scale)))
        return out

    def sample(self, generator=None, scale: float = 1.0, host_draw: bool = False):
        """``mean + std * randn(mean.shape)``.  The draw is torch's, on the parameters' device with the optional generator
        as diffusers' twin does (the one scripts/neuroclips_video.py:267 calls); ``host_draw`` reproduces the sgm
        variant's host draw + ``.to(device)`` (distributions.py:39-41).  The arithmetic is nr_gaussian_sample."""
        shape = self.mean.shape
        if host_draw:
            noise = torch.randn(shape, generator=generator, dtype=torch.float32)
        else:
            noise = torch.randn(shape, generator=generator, device=self.parameters.device, dtype=torch.float32)
        return self._draw(noise.to(self.parameters.device).contiguous(), scale)

    def mode(self, scale: float = 1.0):
        return self._draw(None, scale)


class NativeVAEEncoder(_NativeNet):
    """``first_stage_model.encode`` up to the posterior.  ``load_state_dict`` takes first-stage names (``encoder.*``,
    ``quant_conv.*``); decoder entries are ignored with strict=False."""
    _kind = _lib.NR_KIND_VAE_ENCODER
    _config_cls = VAEDecoderConfig
    MAX_IMAGES = 16

    def _build_cconf(self, config):
        return vae_encoder_c_config(config)

    def _build_schema(self, config):
        return vae_encoder_state_dict_schema(config)

    def _on_plan(self):
        b, f, h, w, L = self._plan_key
        dev = self.device
        self._io_x = torch.empty(b, 3, h, w, dtype=torch.float32, device=dev)

    @_on_device
    def moments(self, x, in_mul: float = 1.0, in_add: float = 0.0):
        """x [n][3][h][w] on the GPU -> fp32 (mean | logvar) [n][2z][h/8][w/8]; the network sees x * in_mul + in_add."""
        if not x.is_cuda:
            raise RuntimeError("NativeVAEEncoder: CUDA (ROCm) tensors required; there is no CPU fallback")
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError(f"expected [n][3][h][w] images, got {tuple(x.shape)}")
        n, _, h, w = x.shape
        down = 2 ** (len(self.config.ch_mult) - 1)
        lib = _lib.load()
        stream = torch.cuda.current_stream().cuda_stream
        outs = []
        for i in range(0, n, self.MAX_IMAGES):
            xc = x[i:i + self.MAX_IMAGES]
            self._ensure_plan(xc.shape[0], 1, h, w, 0)
            self._io_x.copy_(xc)
            out = torch.empty(xc.shape[0], 2 * self.config.z_channels, h // down, w // down, dtype=torch.float32, device=x.device)
            _lib.check(lib.nr_vae_encode(self._handle(), stream, self._io_x.data_ptr(), float(in_mul), float(in_add), out.data_ptr()))
            outs.append(out)
        return outs[0] if len(outs) == 1 else torch.cat(outs)

    def encode(self, x, in_mul: float = 1.0, in_add: float = 0.0):
        return DiagonalGaussianDistribution(self.moments(x, in_mul, in_add))

    __call__ = encode


class _EncodeOutput:
    def __init__(self, dist):
        self.latent_dist = dist


class _DecodeOutput:
    def __init__(self, sample):
        self.sample = sample


class NativeAutoencoderKL:
    """The ``vae`` object of the reference scripts (diffusers ``AutoencoderKL`` surface) on the two native engines:
    ``vae.encode(2 * x - 1).latent_dist.sample() * 0.18215`` (scripts/neuroclips_video.py:267) and
    ``vae.decode(latents).sample`` (pipeline_animation.py:250) work unchanged.  ``load_state_dict`` accepts
    first-stage (LDM) names or diffusers names."""

    def __init__(self, config: VAEDecoderConfig = None):
        self.config = config or VAEDecoderConfig()
        self.encoder = NativeVAEEncoder(self.config)
        self.decoder = NativeVAEDecoder(self.config)
        self.dtype = torch.float32
        self.device = torch.device("cpu")

    def to(self, device=None, dtype=None):
        self.encoder.to(device)
        self.decoder.to(device)
        if device is not None:
            self.device = self.decoder.device
        return self

    def eval(self):
        return self

    def requires_grad_(self, flag=False):
        return self

    def load_state_dict(self, sd, strict=True):
        if any(k.startswith(("decoder.up_blocks", "encoder.down_blocks")) for k in sd):
            sd = {**convert_diffusers_vae_state_dict(sd, self.config, False), **convert_diffusers_vae_state_dict(sd, self.config, True)}
        dec = {k: v for k, v in sd.items() if k in self.decoder._schema}
        enc = {k: v for k, v in sd.items() if k in self.encoder._schema}
        m1, _ = self.decoder.load_state_dict(dec, strict=False)
        m2, _ = self.encoder.load_state_dict(enc, strict=False)
        missing = m1 + m2
        unexpected = [k for k in sd if k not in dec and k not in enc and not k.startswith("loss.")]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict for NativeAutoencoderKL: missing {missing[:6]} unexpected {unexpected[:6]}")
        return missing, unexpected

    def load_ldm_state_dict(self, sd, strict=True):
        """First-stage tensors of an LDM / DreamBooth checkpoint (``first_stage_model.`` prefix already stripped): the native
        engines use these names as they are, so no key conversion is needed (reference: convert_ldm_vae_checkpoint, util.py:138)."""
        return self.load_state_dict(sd, strict=strict)

    def encode(self, x):
        return _EncodeOutput(self.encoder.encode(x))

    def decode(self, z):
        return _DecodeOutput(self.decoder.decode(z))
