"""Generate the golden vectors under tests/golden/ by running the REFERENCE's own classes.

Runs only in the build container (needs /root/reference; the GPU box never sees it).  Nothing from the
reference is copied: its modules are imported from where they lie, fed deterministic synthetic weights and
inputs (neurons_amd.synth: a recipe, not data), and only inputs/outputs are stored.

Scaffolding: the reference imports un-vendored ``diffusers==0.11.1`` and ``torchvision``.  We
register throw-away in-memory stand-ins for the NON-arithmetic pieces (ConfigMixin, ModelMixin, BaseOutput,
logging, is_xformers_available).  The arithmetic pieces are wired to the reference's own text:
  diffusers.models.attention.{CrossAttention, FeedForward}  <- animatediff/models/motion_module_new.py:119,429
  diffusers.models.embeddings.Timesteps                      <- generative_models/sgm/modules/diffusionmodules/util.py:207
  diffusers.models.embeddings.TimestepEmbedding              =  Linear -> SiLU -> Linear (as sgm openaimodel.py:590-594)
The DDIM scheduler has no in-repo source; the loop fixture uses oracle.animatediff_oracle's restatement for the
scheduler and the reference classes for the networks (parity for the scheduler stays "unpinned").

Usage:  python oracle/gen_golden.py            (writes tests/golden/*.npz, a few hundred KB each)
"""
import functools
import importlib.util
import json
import inspect
import logging
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)


# --------------------------------------------------------------------------------------------------
# scaffolding
# --------------------------------------------------------------------------------------------------
class _FrozenDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


def _register_to_config(init):
    @functools.wraps(init)
    def inner(self, *args, **kwargs):
        sig = inspect.signature(init)
        bound = sig.bind(self, *args, **kwargs)
        bound.apply_defaults()
        cfg = {k: v for k, v in bound.arguments.items() if k != "self"}
        object.__setattr__(self, "_internal_dict", _FrozenDict(cfg))
        init(self, *args, **kwargs)
        # diffusers 0.11.1 ConfigMixin.register_to_config also exposes every config entry as an attribute (the reference pipeline reads
        # ``self.unet.in_channels``, pipeline_neuroclips.py:382); attributes the module defines itself win
        for k, v in cfg.items():
            if not hasattr(self, k):
                try:
                    object.__setattr__(self, k, v)
                except Exception:
                    pass
    return inner


class _ConfigMixin:
    @property
    def config(self):
        return self._internal_dict


class _ModelMixin(nn.Module):
    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device


class _BaseOutput:
    pass


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_scaffolding():
    if "diffusers" in sys.modules and getattr(sys.modules["diffusers"], "_nr_stub", False):
        return
    log = types.SimpleNamespace(get_logger=lambda name=None: logging.getLogger(name or "ref"))
    d = _mod("diffusers", _nr_stub=True, __version__="0.11.1-stub")
    d.__path__ = []
    _mod("diffusers.configuration_utils", ConfigMixin=_ConfigMixin, register_to_config=_register_to_config, FrozenDict=_FrozenDict)
    _mod("diffusers.modeling_utils", ModelMixin=_ModelMixin)
    u = _mod("diffusers.utils", BaseOutput=_BaseOutput, logging=log)
    u.__path__ = []
    _mod("diffusers.utils.import_utils", is_xformers_available=lambda: False)
    m = _mod("diffusers.models")
    m.__path__ = []
    _mod("diffusers.models.unet_2d_condition", UNet2DConditionModel=type("UNet2DConditionModel", (), {}))
    tv = _mod("torchvision")
    tv.__path__ = []

    # arithmetic: the reference's own vendored copy of diffusers' attention/FF
    spec = importlib.util.spec_from_file_location("ref_motion_module_new", f"{REF}/animatediff/models/motion_module_new.py")
    mmn = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mmn)

    class AdaLayerNorm(nn.Module):  # never instantiated (num_embeds_ada_norm=None on the NEURONS path)
        def __init__(self, *a, **k):
            raise NotImplementedError

    _mod("diffusers.models.attention", CrossAttention=mmn.CrossAttention, FeedForward=mmn.FeedForward, AdaLayerNorm=AdaLayerNorm)

    # arithmetic: sinusoid from the reference's sgm util
    spec = importlib.util.spec_from_file_location("ref_sgm_util", f"{REF}/generative_models/sgm/modules/diffusionmodules/util.py")
    sgm_util = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sgm_util)

    class Timesteps(nn.Module):
        def __init__(self, num_channels, flip_sin_to_cos, downscale_freq_shift):
            super().__init__()
            assert flip_sin_to_cos and downscale_freq_shift == 0, "SD-1.5 config (unet.py:48-49)"
            self.num_channels = num_channels

        def forward(self, timesteps):
            return sgm_util.timestep_embedding(timesteps, self.num_channels)

    class TimestepEmbedding(nn.Module):
        def __init__(self, in_channels, time_embed_dim, act_fn="silu"):
            super().__init__()
            self.linear_1 = nn.Linear(in_channels, time_embed_dim)
            self.act = nn.SiLU()
            self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)

        def forward(self, sample):
            return self.linear_2(self.act(self.linear_1(sample)))

    _mod("diffusers.models.embeddings", Timesteps=Timesteps, TimestepEmbedding=TimestepEmbedding)
    if REF not in sys.path:
        sys.path.insert(0, REF)


def reference_classes():
    install_scaffolding()
    from animatediff.models.unet import UNet3DConditionModel
    from animatediff.models.sparse_controlnet import SparseControlNetModel
    from animatediff.models import attention as ref_attention, motion_module as ref_mm, resnet as ref_resnet
    return UNet3DConditionModel, SparseControlNetModel, ref_attention, ref_mm, ref_resnet


# shared tiny configurations live in tests/tiny_configs.py (plain configuration, no reference import)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from tiny_configs import tiny_clip_config, tiny_ctrl_config, tiny_sgm_config, tiny_unet_config, tiny_vae_config  # noqa: E402,F401
def build_reference_unet(cfg, sd):
    UNet3D, _, _, _, _ = reference_classes()
    mm = dict(cfg.motion_module_kwargs)
    mm["attention_block_types"] = list(mm["attention_block_types"])
    net = UNet3D(sample_size=cfg.sample_size, in_channels=cfg.in_channels, out_channels=cfg.out_channels,
                 down_block_types=tuple(cfg.down_block_types), up_block_types=tuple(cfg.up_block_types),
                 block_out_channels=tuple(cfg.block_out_channels), layers_per_block=cfg.layers_per_block,
                 norm_num_groups=cfg.norm_num_groups, norm_eps=cfg.norm_eps, cross_attention_dim=cfg.cross_attention_dim,
                 attention_head_dim=cfg.attention_head_dim, use_inflated_groupnorm=True, use_motion_module=True,
                 motion_module_resolutions=tuple(cfg.motion_module_resolutions), motion_module_mid_block=False,
                 motion_module_type="Vanilla", motion_module_kwargs=mm,
                 unet_use_cross_frame_attention=False, unet_use_temporal_attention=False)
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(k.endswith("pos_encoder.pe") for k in missing), missing
    ref_keys = {k for k in net.state_dict().keys()}
    assert ref_keys == set(sd.keys()), "schema differs from the reference's state_dict keys"
    return net.eval()


def build_reference_ctrl(cfg, sd):
    _, Ctrl, _, _, _ = reference_classes()
    mm = dict(cfg.motion_module_kwargs)
    mm["attention_block_types"] = list(mm["attention_block_types"])
    net = Ctrl(in_channels=cfg.in_channels, conditioning_channels=cfg.conditioning_channels,
               down_block_types=tuple(cfg.down_block_types), block_out_channels=tuple(cfg.block_out_channels),
               layers_per_block=cfg.layers_per_block, norm_num_groups=cfg.norm_num_groups, norm_eps=cfg.norm_eps,
               cross_attention_dim=cfg.cross_attention_dim, attention_head_dim=cfg.attention_head_dim,
               use_motion_module=True, motion_module_resolutions=(1, 2, 4, 8), motion_module_mid_block=False,
               motion_module_type="Vanilla", motion_module_kwargs=mm, concate_conditioning_mask=True,
               use_simplified_condition_embedding=True, set_noisy_sample_input_to_zero=True)
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(k.endswith("pos_encoder.pe") for k in missing), missing
    assert set(net.state_dict().keys()) == set(sd.keys()), "schema differs from the reference's state_dict keys"
    return net.eval()


def _sub(t, n=4096):
    """deterministic subsample of a big tensor: (flat indices, values)"""
    flat = t.detach().reshape(-1)
    if flat.numel() <= n:
        idx = np.arange(flat.numel())
    else:
        idx = (np.arange(n, dtype=np.int64) * 2654435761 % flat.numel())
    return idx.astype(np.int64), flat[torch.from_numpy(idx)].numpy().astype(np.float32)


@torch.no_grad()
def gen_networks(out_dir):
    from neurons_amd import _lib
    from neurons_amd.synth import randn
    from neurons_amd.unet3d import random_state_dict
    ucfg, ccfg = tiny_unet_config(), tiny_ctrl_config()
    usd = random_state_dict(ucfg, _lib.NR_KIND_UNET3D, seed=11)
    csd = random_state_dict(ccfg, _lib.NR_KIND_SPARSECTRL, seed=12)
    unet = build_reference_unet(ucfg, usd)
    ctrl = build_reference_ctrl(ccfg, csd)
    B, F, H, W = 1, 8, 8, 8
    sample = randn("sample", (2 * B, 4, F, H, W), 21)
    ctx = randn("ctx", (2 * B, 77, ucfg.cross_attention_dim), 22)
    cond = torch.zeros(B, 4, F, H, W)
    cond[:, :, 0] = randn("cond", (B, 4, H, W), 23) * 0.18215
    mask = torch.zeros(B, 1, F, H, W)
    mask[:, :, 0] = 1
    t = 681

    # hook a few intermediate modules of the reference U-Net
    taps = {}
    names = ["down_blocks.0.resnets.0", "down_blocks.0.attentions.0", "down_blocks.0.motion_modules.0",
             "down_blocks.1.downsamplers.0", "mid_block.resnets.1", "up_blocks.1.upsamplers.0", "up_blocks.3.motion_modules.2"]
    mods = dict(unet.named_modules())
    hooks = []
    for n in names:
        def mk(n):
            def hook(m, i, o):
                taps[n] = o.sample if hasattr(o, "sample") else o
            return hook
        hooks.append(mods[n].register_forward_hook(mk(n)))
    eps_plain = unet(sample, t, encoder_hidden_states=ctx).sample
    for h in hooks:
        h.remove()
    down, mid = ctrl(sample, t, encoder_hidden_states=ctx, controlnet_cond=cond, conditioning_mask=mask,
                     conditioning_scale=1.0, guess_mode=False, return_dict=False)
    eps_ctrl = unet(sample, t, encoder_hidden_states=ctx, down_block_additional_residuals=down,
                    mid_block_additional_residual=mid).sample
    out = dict(sample=sample.numpy(), ctx=ctx.numpy(), cond=cond.numpy(), mask=mask.numpy(), t=np.int64(t),
               eps_plain=eps_plain.numpy(), eps_ctrl=eps_ctrl.numpy(), mid_res=mid.numpy())
    for i, d in enumerate(down):
        out[f"down_res_{i}"] = d.numpy()
    for n, v in taps.items():
        idx, val = _sub(v)
        out[f"tap_idx:{n}"] = idx
        out[f"tap_val:{n}"] = val
        out[f"tap_shape:{n}"] = np.array(v.shape)
    np.savez_compressed(os.path.join(out_dir, "tiny_networks.npz"), **out)
    print("tiny_networks:", {k: getattr(v, "shape", None) for k, v in out.items() if not k.startswith("tap_")})
    return unet, ctrl, usd, csd, ucfg, ccfg


@torch.no_grad()
def gen_loop(out_dir, unet, ctrl):
    """BASELINE config 1: single 8-frame 64x64 clip (8x8 latent), 10 DDIM steps, guidance 8.5, random conditioning."""
    from neurons_amd.synth import randn
    from oracle import animatediff_oracle as O
    B, F, H, W, N, s = 1, 8, 8, 8, 10, 8.5
    latents = randn("c1.latents", (B, 4, F, H, W), 31)
    noise = randn("c1.noise", (B, 4, F, H, W), 32)
    ctx = randn("c1.ctx", (2 * B, 77, 64), 33)
    cimg = randn("c1.cimg", (B, 4, 1, H, W), 34) * 0.18215
    ac = O.ddim_alphas_cumprod()
    ts = O.ddim_timesteps(N)
    x = O.add_noise(latents, noise, ts[0], ac)
    cond = torch.zeros(B, 4, F, H, W)
    cond[:, :, [0]] = cimg[:, :, :1]
    mask = torch.zeros(B, 1, F, H, W)
    mask[:, :, [0]] = 1
    eps_log = {}
    for i, t in enumerate(ts):
        xin = torch.cat([x] * 2)
        down, mid = ctrl(xin, t, encoder_hidden_states=ctx, controlnet_cond=cond, conditioning_mask=mask,
                         conditioning_scale=1.0, guess_mode=False, return_dict=False)
        eps = unet(xin, t, encoder_hidden_states=ctx, down_block_additional_residuals=down,
                   mid_block_additional_residual=mid).sample
        if i in (0, 1, N - 1):
            eps_log[i] = eps.numpy()
        eu, et = eps.chunk(2)
        e = eu + s * (et - eu)
        x = O.ddim_step(e, t, x, ac, N)
    out = dict(latents=latents.numpy(), noise=noise.numpy(), ctx=ctx.numpy(), cimg=cimg.numpy(), steps=np.int64(N),
               guidance=np.float64(s), timesteps=np.array(ts), final=x.numpy())
    for i, e in eps_log.items():
        out[f"eps_step_{i}"] = e
    np.savez_compressed(os.path.join(out_dir, "c1_loop.npz"), **out)
    print("c1_loop: final latents", x.shape, "abs mean", x.abs().mean().item())


@torch.no_grad()
def gen_leaf_ops(out_dir):
    """Full-width leaf modules of the reference (head dims 40/80/160, Cin 2560 @ 4x4, temporal attention with PE)."""
    from neurons_amd.synth import randn
    _, _, ref_attention, ref_mm, ref_resnet = reference_classes()
    CrossAttention = sys.modules["diffusers.models.attention"].CrossAttention
    FeedForward = sys.modules["diffusers.models.attention"].FeedForward
    out = {}

    def fill(mod, tag, seed):
        sd = {}
        for k, v in mod.state_dict().items():
            if k.endswith("pos_encoder.pe"):
                continue
            z = randn(f"{tag}.{k}", tuple(v.shape), seed)
            if v.dim() == 1:
                z = (1.0 + 0.1 * z) if k.endswith("weight") else 0.05 * z
            else:
                z = z / (int(np.prod(v.shape[1:])) ** 0.5)
            sd[k] = z
        mod.load_state_dict(sd, strict=False)
        return mod.eval()

    for C in (320, 640, 1280):
        a = fill(CrossAttention(query_dim=C, heads=8, dim_head=C // 8), f"attn{C}", 41)
        x = randn(f"attn{C}.x", (2, 48, C), 42)
        out[f"selfattn{C}.y"] = a(x).numpy()
        ac = fill(CrossAttention(query_dim=C, cross_attention_dim=768, heads=8, dim_head=C // 8), f"xattn{C}", 43)
        ctx = randn(f"xattn{C}.ctx", (2, 77, 768), 44)
        out[f"crossattn{C}.y"] = ac(x, encoder_hidden_states=ctx).numpy()
    ff = fill(FeedForward(320, activation_fn="geglu"), "ff320", 45)
    out["ff320.y"] = ff(randn("ff320.x", (2, 48, 320), 46)).numpy()
    va = fill(ref_mm.VersatileAttention(attention_mode="Temporal", cross_attention_dim=None, query_dim=320, heads=8, dim_head=40,
                                        temporal_position_encoding=True, temporal_position_encoding_max_len=24), "va320", 47)
    out["va320.y"] = va(randn("va320.x", (2 * 16, 6, 320), 48), video_length=16).numpy()
    rb = fill(ref_resnet.ResnetBlock3D(in_channels=2560, out_channels=1280, temb_channels=1280, eps=1e-5, groups=32,
                                       use_inflated_groupnorm=True), "rb2560", 49)
    y = rb(randn("rb2560.x", (1, 2560, 2, 4, 4), 50), randn("rb2560.temb", (1, 1280), 51))
    out["rb2560.y"] = y.numpy()
    tm = fill(ref_mm.VanillaTemporalModule(in_channels=320, num_attention_heads=8, num_transformer_block=1,
                                           attention_block_types=("Temporal_Self", "Temporal_Self"),
                                           temporal_position_encoding=True, temporal_position_encoding_max_len=24,
                                           zero_initialize=False), "tm320", 52)
    out["tm320.y"] = tm(randn("tm320.x", (1, 320, 16, 3, 3), 53), None, None).numpy()
    t3 = fill(ref_attention.Transformer3DModel(8, 40, in_channels=320, num_layers=1, cross_attention_dim=768, norm_num_groups=32,
                                               unet_use_cross_frame_attention=False, unet_use_temporal_attention=False), "t3d320", 54)
    out["t3d320.y"] = t3(randn("t3d320.x", (1, 320, 2, 4, 4), 55), encoder_hidden_states=randn("t3d320.ctx", (1, 77, 768), 56)).sample.numpy()
    # The same two composite modules at the row counts where the engine switches to its fused kernels (tattn.hip / ffpanel.hip need
    # >= 4096 rows at C = 320): 16 frames x 16x16 = 4096 rows, 2 frames x 48x48 = 4608 rows.  Outputs stored as a deterministic subsample.
    tmb = fill(ref_mm.VanillaTemporalModule(in_channels=320, num_attention_heads=8, num_transformer_block=1,
                                            attention_block_types=("Temporal_Self", "Temporal_Self"),
                                            temporal_position_encoding=True, temporal_position_encoding_max_len=24,
                                            zero_initialize=False), "tm320big", 61)
    yb = tmb(randn("tm320big.x", (1, 320, 16, 16, 16), 62), None, None)
    out["tm320big.idx"], out["tm320big.val"] = _sub(yb, 16384)
    out["tm320big.shape"] = np.array(yb.shape)
    # the 32-frame clips of BASELINE config 5 (temporal_position_encoding_max_len = 32): 32 frames x 12x12 = 4608 rows reach the two-row-tile form
    # of the fused temporal-attention kernel (tattn.hip, F = 32)
    tm32 = fill(ref_mm.VanillaTemporalModule(in_channels=320, num_attention_heads=8, num_transformer_block=1,
                                             attention_block_types=("Temporal_Self", "Temporal_Self"),
                                             temporal_position_encoding=True, temporal_position_encoding_max_len=32,
                                             zero_initialize=False), "tm320f32", 66)
    yb = tm32(randn("tm320f32.x", (1, 320, 32, 12, 12), 67), None, None)
    out["tm320f32.idx"], out["tm320f32.val"] = _sub(yb, 16384)
    out["tm320f32.shape"] = np.array(yb.shape)
    t3b = fill(ref_attention.Transformer3DModel(8, 40, in_channels=320, num_layers=1, cross_attention_dim=768, norm_num_groups=32,
                                                unet_use_cross_frame_attention=False, unet_use_temporal_attention=False), "t3d320big", 63)
    yb = t3b(randn("t3d320big.x", (1, 320, 2, 48, 48), 64), encoder_hidden_states=randn("t3d320big.ctx", (1, 77, 768), 65)).sample
    out["t3d320big.idx"], out["t3d320big.val"] = _sub(yb, 16384)
    out["t3d320big.shape"] = np.array(yb.shape)
    up = fill(ref_resnet.Upsample3D(64, use_conv=True, out_channels=64), "up64", 57)
    out["up64.y"] = up(randn("up64.x", (1, 64, 2, 3, 5), 58)).numpy()
    dn = fill(ref_resnet.Downsample3D(64, use_conv=True, out_channels=64, padding=1, name="op"), "dn64", 59)
    out["dn64.y"] = dn(randn("dn64.x", (1, 64, 2, 6, 10), 60)).numpy()
    np.savez_compressed(os.path.join(out_dir, "leaf_ops.npz"), **out)
    print("leaf_ops:", {k: v.shape for k, v in out.items()})


@torch.no_grad()
def gen_leaf_wide(out_dir):
    """Round 5: the two composite modules of the 16x16 and 8x8 levels -- C = 640 (8 heads of d = 80) and C = 1280 (d = 160) -- at
    exactly the shapes BASELINE config 2 runs them: CFG batch 2 x 16 frames x 16x16 (8192 rows) and x 8x8 (2048 rows).  Reference classes
    (attention.py:95-142,256-300; motion_module.py:134-158,210-222), weights / inputs from the Philox recipe of gen_leaf_ops, outputs stored as
    a deterministic subsample (tests/golden/leaf_wide.npz)."""
    from neurons_amd.synth import randn
    _, _, ref_attention, ref_mm, _ = reference_classes()
    out = {}

    def fill(mod, tag, seed):
        sd = {}
        for k, v in mod.state_dict().items():
            if k.endswith("pos_encoder.pe"):
                continue
            z = randn(f"{tag}.{k}", tuple(v.shape), seed)
            if v.dim() == 1:
                z = (1.0 + 0.1 * z) if k.endswith("weight") else 0.05 * z
            else:
                z = z / (int(np.prod(v.shape[1:])) ** 0.5)
            sd[k] = z
        mod.load_state_dict(sd, strict=False)
        return mod.eval()

    for C, hw, seed in ((640, 16, 71), (1280, 8, 75)):
        tm = fill(ref_mm.VanillaTemporalModule(in_channels=C, num_attention_heads=8, num_transformer_block=1,
                                               attention_block_types=("Temporal_Self", "Temporal_Self"),
                                               temporal_position_encoding=True, temporal_position_encoding_max_len=24,
                                               zero_initialize=False), f"tm{C}", seed)
        y = tm(randn(f"tm{C}.x", (2, C, 16, hw, hw), seed + 1), None, None)
        out[f"tm{C}.idx"], out[f"tm{C}.val"] = _sub(y, 16384)
        out[f"tm{C}.shape"] = np.array(y.shape)
        t3 = fill(ref_attention.Transformer3DModel(8, C // 8, in_channels=C, num_layers=1, cross_attention_dim=768, norm_num_groups=32,
                                                   unet_use_cross_frame_attention=False, unet_use_temporal_attention=False), f"t3d{C}", seed + 2)
        y = t3(randn(f"t3d{C}.x", (2, C, 16, hw, hw), seed + 3), encoder_hidden_states=randn(f"t3d{C}.ctx", (2, 77, 768), seed + 4)).sample
        out[f"t3d{C}.idx"], out[f"t3d{C}.val"] = _sub(y, 16384)
        out[f"t3d{C}.shape"] = np.array(y.shape)
        print(f"leaf_wide: C = {C} done", flush=True)
    np.savez_compressed(os.path.join(out_dir, "leaf_wide.npz"), **out)
    print("leaf_wide:", {k: v.shape for k, v in out.items()})


# --------------------------------------------------------------------------------------------------
# sgm unCLIP path: import the reference's own UNetModel / sampler (bypass sgm/__init__.py, which
# pulls Lightning/open_clip, by pre-registering namespace packages; omegaconf is only used in annotations)
# --------------------------------------------------------------------------------------------------
def install_sgm_scaffolding():
    if "sgm" in sys.modules and getattr(sys.modules["sgm"], "_nr_stub", False):
        return
    base = f"{REF}/generative_models/sgm"
    for name, path in (("sgm", base), ("sgm.modules", f"{base}/modules"), ("sgm.modules.diffusionmodules", f"{base}/modules/diffusionmodules")):
        m = types.ModuleType(name)
        m.__path__ = [path]
        m._nr_stub = True
        sys.modules[name] = m
    if "omegaconf" not in sys.modules:
        _mod("omegaconf", ListConfig=list, OmegaConf=type("OmegaConf", (), {}), DictConfig=dict)


def build_reference_sgm(cfg, sd):
    install_sgm_scaffolding()
    from sgm.modules.diffusionmodules.openaimodel import UNetModel
    net = UNetModel(in_channels=cfg.in_channels, model_channels=cfg.model_channels, out_channels=cfg.out_channels,
                    num_res_blocks=cfg.num_res_blocks, attention_resolutions=list(cfg.attention_resolutions),
                    channel_mult=list(cfg.channel_mult), num_classes="sequential", use_checkpoint=False,
                    num_head_channels=cfg.num_head_channels, use_linear_in_transformer=True,
                    transformer_depth=list(cfg.transformer_depth), context_dim=cfg.context_dim,
                    adm_in_channels=cfg.adm_in_channels, spatial_transformer_attn_type="softmax")
    assert set(net.state_dict().keys()) == set(sd.keys()), set(net.state_dict().keys()) ^ set(sd.keys())
    net.load_state_dict(sd, strict=True)
    return net.eval()


@torch.no_grad()
def gen_sgm(out_dir):
    from neurons_amd.sgm import sgm_random_state_dict
    from neurons_amd.synth import randn
    install_sgm_scaffolding()
    from sgm.modules.diffusionmodules.denoiser import DiscreteDenoiser
    from sgm.modules.diffusionmodules.discretizer import LegacyDDPMDiscretization
    from sgm.modules.diffusionmodules.sampling import EulerEDMSampler
    from sgm.modules.diffusionmodules.wrappers import OpenAIWrapper
    cfg = tiny_sgm_config()
    sd = sgm_random_state_dict(cfg, seed=71)
    net = build_reference_sgm(cfg, sd)
    x = randn("sgm.x", (2, 4, 16, 16), 72)
    ctx = randn("sgm.ctx", (2, 24, cfg.context_dim), 73)
    y = randn("sgm.y", (2, cfg.adm_in_channels), 74)
    t = torch.tensor([437, 437])
    out = dict(x=x.numpy(), ctx=ctx.numpy(), y=y.numpy(), t=t.numpy())
    out["eps"] = net(x, timesteps=t, context=ctx, y=y).numpy()
    disc = LegacyDDPMDiscretization()
    out["sigmas38"] = disc(38).numpy()
    out["sigmas50"] = disc(50).numpy()
    # 4-step Euler-EDM + VanillaCFG(5.0) loop exactly as utils.unclip_recon wires it (utils.py:337-340)
    sampler = EulerEDMSampler(num_steps=4, discretization_config={"target": "sgm.modules.diffusionmodules.discretizer.LegacyDDPMDiscretization"},
                              guider_config={"target": "sgm.modules.diffusionmodules.guiders.VanillaCFG", "params": {"scale": 5.0}},
                              device="cpu")
    denoiser = DiscreteDenoiser(scaling_config={"target": "sgm.modules.diffusionmodules.denoiser_scaling.EpsScaling"}, num_idx=1000,
                                discretization_config={"target": "sgm.modules.diffusionmodules.discretizer.LegacyDDPMDiscretization"})
    model = OpenAIWrapper(net)
    z = randn("sgm.z", (1, 4, 16, 16), 75)
    c = {"crossattn": ctx[1:2], "vector": y[1:2]}
    uc = {"crossattn": ctx[0:1], "vector": y[1:2]}
    final = sampler(lambda xx, sigma, cc: denoiser(model, xx, sigma, cc), z.clone(), cond=c, uc=uc)
    out["z"] = z.numpy()
    out["loop_final"] = final.numpy()
    np.savez_compressed(os.path.join(out_dir, "sgm_tiny.npz"), **out)
    print("sgm_tiny:", {k: v.shape for k, v in out.items()}, "final abs mean", final.abs().mean().item())


# --------------------------------------------------------------------------------------------------
# first-stage decoder: the reference's own sgm Decoder (+ post_quant_conv as AutoencodingEngineLegacy.decode applies
# it, autoencoder.py:459,490-494; that class itself needs Lightning) and the diffusers<->LDM VAE key map from the
# reference's convert_ldm_vae_checkpoint
# --------------------------------------------------------------------------------------------------
@torch.no_grad()
def gen_vae(out_dir):
    from neurons_amd.vae import vae_random_state_dict, diffusers_vae_key_map, VAEDecoderConfig
    from neurons_amd.synth import randn
    install_sgm_scaffolding()
    from sgm.modules.diffusionmodules.model import Decoder
    cfg = tiny_vae_config()
    sd = vae_random_state_dict(cfg, seed=91)
    dec = Decoder(ch=cfg.ch, out_ch=cfg.out_ch, ch_mult=cfg.ch_mult, num_res_blocks=cfg.num_res_blocks, attn_resolutions=[],
                  in_channels=3, resolution=64, z_channels=cfg.z_channels, attn_type="vanilla", double_z=True)
    dsd = {k[len("decoder."):]: v for k, v in sd.items() if k.startswith("decoder.")}
    assert set(dec.state_dict().keys()) == set(dsd.keys()), set(dec.state_dict().keys()) ^ set(dsd.keys())
    dec.load_state_dict(dsd, strict=True)
    dec.eval()
    pq = torch.nn.Conv2d(cfg.embed_dim, cfg.z_channels, 1)
    pq.load_state_dict({"weight": sd["post_quant_conv.weight"], "bias": sd["post_quant_conv.bias"]})
    z = randn("vae.z", (2, 4, 8, 8), 92)
    out = dict(z=z.numpy())
    out["image"] = dec(pq(z / 0.18215)).numpy()          # decode_first_stage (diffusion.py:118-135)
    # decode_latents (pipeline_animation.py:243-256) on a (1, 4, 2, 8, 16) clip: per-frame decode, /2+.5, clamp
    lat = randn("vae.lat", (1, 4, 2, 8, 16), 93) * 0.5
    fr = (lat / 0.18215).permute(0, 2, 1, 3, 4).reshape(2, 4, 8, 16)
    vid = torch.cat([dec(pq(fr[i:i + 1])) for i in range(2)])
    out["lat"] = lat.numpy()
    out["video"] = (vid.reshape(1, 2, 3, 64, 128).permute(0, 2, 1, 3, 4) / 2 + 0.5).clamp(0, 1).numpy()
    # encoder: vae.encode(2 * x - 1).latent_dist.sample() * 0.18215 (scripts/neuroclips_video.py:267) with the reference's
    # Encoder + quant_conv + DiagonalGaussianDistribution; the noise of .sample() is recorded by seeding torch
    from sgm.modules.diffusionmodules.model import Encoder
    spec = importlib.util.spec_from_file_location("ref_sgm_dist", f"{REF}/generative_models/sgm/modules/distributions/distributions.py")
    dist = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(dist)
    esd = vae_random_state_dict(cfg, seed=94, encoder=True)
    enc = Encoder(ch=cfg.ch, out_ch=cfg.out_ch, ch_mult=cfg.ch_mult, num_res_blocks=cfg.num_res_blocks, attn_resolutions=[],
                  in_channels=3, resolution=64, z_channels=cfg.z_channels, attn_type="vanilla", double_z=True)
    e2 = {k[len("encoder."):]: v for k, v in esd.items() if k.startswith("encoder.")}
    assert set(enc.state_dict().keys()) == set(e2.keys()), set(enc.state_dict().keys()) ^ set(e2.keys())
    enc.load_state_dict(e2, strict=True)
    enc.eval()
    qc = torch.nn.Conv2d(2 * cfg.z_channels, 2 * cfg.embed_dim, 1)
    qc.load_state_dict({"weight": esd["quant_conv.weight"], "bias": esd["quant_conv.bias"]})
    img = torch.rand(torch.Size((2, 3, 64, 128)), generator=torch.Generator().manual_seed(95))
    moments = qc(enc(2 * img - 1))
    post = dist.DiagonalGaussianDistribution(moments)
    torch.manual_seed(96)
    lat_s = post.sample() * 0.18215
    torch.manual_seed(96)
    out.update(enc_img=img.numpy(), enc_moments=moments.numpy(), enc_noise=torch.randn(post.mean.shape).numpy(),
               enc_sample=lat_s.numpy(), enc_mode=(post.mode() * 0.18215).numpy())
    np.savez_compressed(os.path.join(out_dir, "vae_tiny.npz"), **out)
    print("vae_tiny:", {k: v.shape for k, v in out.items()}, "image abs mean", float(np.abs(out["image"]).mean()),
          "video mean", float(out["video"].mean()), "clamped frac", float(((out["video"] == 0) | (out["video"] == 1)).mean()))
    # key map: run the reference converter on an LDM-named VAE checkpoint whose tensors encode their own names
    conv = load_ref_converter()
    full = VAEDecoderConfig()
    from neurons_amd.vae import vae_decoder_state_dict_schema
    names = list(vae_decoder_state_dict_schema(full).keys())
    ck = {"first_stage_model." + k: torch.full((1,), float(i)) for i, k in enumerate(names)}
    # the converter also reads encoder / quant_conv entries: give it the ones it indexes unconditionally
    for k in ("encoder.conv_in", "encoder.conv_out", "encoder.norm_out", "quant_conv"):
        ck[f"first_stage_model.{k}.weight"] = torch.zeros(1)
        ck[f"first_stage_model.{k}.bias"] = torch.zeros(1)
    for k in ("q", "k", "v", "proj_out"):      # the converter squeezes these 1x1-conv weights ([:, :, 0, 0] / [:, :, 0])
        n = f"first_stage_model.decoder.mid.attn_1.{k}.weight"
        ck[n] = ck[n].reshape(1, 1, 1, 1)
    class _Cfg(dict):
        __getattr__ = dict.__getitem__
    vcfg = _Cfg(down_block_types=["DownEncoderBlock2D"] * 4, up_block_types=["UpDecoderBlock2D"] * 4, layers_per_block=2)
    conv_sd = conv.convert_ldm_vae_checkpoint(ck, vcfg)
    ref_map = {k: names[int(v.reshape(-1)[0])] for k, v in conv_sd.items() if k.startswith(("decoder.", "post_quant_conv."))}
    ours = diffusers_vae_key_map(full)
    # encoder half: same trick on an encoder-named checkpoint
    from neurons_amd.vae import vae_encoder_state_dict_schema
    enames = list(vae_encoder_state_dict_schema(full).keys())
    ck = {"first_stage_model." + k: torch.full((1,), float(i)) for i, k in enumerate(enames)}
    for k in ("decoder.conv_in", "decoder.conv_out", "decoder.norm_out", "post_quant_conv"):
        ck[f"first_stage_model.{k}.weight"] = torch.zeros(1)
        ck[f"first_stage_model.{k}.bias"] = torch.zeros(1)
    for k in ("q", "k", "v", "proj_out"):
        n = f"first_stage_model.encoder.mid.attn_1.{k}.weight"
        ck[n] = ck[n].reshape(1, 1, 1, 1)
    conv_sd = conv.convert_ldm_vae_checkpoint(ck, vcfg)
    ref_emap = {k: enames[int(v.reshape(-1)[0])] for k, v in conv_sd.items() if k.startswith(("encoder.", "quant_conv."))}
    ours_e = diffusers_vae_key_map(full, encoder=True)
    with open(os.path.join(out_dir, "vae_keys.json"), "w") as f:
        json.dump({"diffusers_to_ldm": ref_map, "diffusers_to_ldm_encoder": ref_emap}, f, indent=0, sort_keys=True)
    print("vae_keys.json:", len(ref_map), "decoder keys; map equal to ours:", ref_map == ours, ";", len(ref_emap),
          "encoder keys; equal:", ref_emap == ours_e)
    if ref_map != ours or ref_emap != ours_e:
        print(sorted(set(ref_map.items()) ^ set(ours.items()))[:10], sorted(set(ref_emap.items()) ^ set(ours_e.items()))[:10])


# --------------------------------------------------------------------------------------------------
# CLIP text encoder: the installed transformers CLIPTextModel (the class _encode_prompt calls; the reference pins
# transformers==4.47.1, this container has a newer release with the same CLIP text arithmetic)
# --------------------------------------------------------------------------------------------------
@torch.no_grad()
def gen_clip(out_dir):
    for name in [m for m in sys.modules if m == "diffusers" or m.startswith("diffusers.")]:
        pass    # the in-memory diffusers stubs do not interfere with transformers' CLIP
    tv_stub = sys.modules.pop("torchvision", None)
    try:
        import transformers
        from transformers import CLIPTextConfig as HFConfig, CLIPTextModel
    finally:
        if tv_stub is not None:
            sys.modules["torchvision"] = tv_stub
    from neurons_amd.clip import clip_random_state_dict
    cfg = tiny_clip_config()
    sd = clip_random_state_dict(cfg, seed=97)
    hf = CLIPTextModel(HFConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                                num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                                max_position_embeddings=cfg.max_position_embeddings, hidden_act="quick_gelu",
                                projection_dim=64, attn_implementation="eager"))
    hsd = hf.state_dict()
    own = {k for k in hsd if not k.endswith("position_ids")}
    # transformers 4.x (the reference's pin, and every SD-1.5 checkpoint) prefixes the keys with "text_model."; 5.x dropped it
    strip = not any(k.startswith("text_model.") for k in own)
    load = {(k[len("text_model."):] if strip else k): v for k, v in sd.items()}
    assert own == set(load.keys()), own ^ set(load.keys())
    missing, unexpected = hf.load_state_dict(load, strict=False)
    assert not unexpected and all(k.endswith("position_ids") for k in missing), (missing, unexpected)
    hf.eval()
    g = torch.Generator().manual_seed(98)
    ids = torch.randint(0, cfg.vocab_size, (2, 77), generator=g)
    ids[1, 10:] = cfg.vocab_size - 1           # a short prompt padded with the end token, like the "" negative prompt
    out = hf(ids, attention_mask=None)[0]
    np.savez_compressed(os.path.join(out_dir, "clip_tiny.npz"), ids=ids.numpy().astype(np.int32), last_hidden_state=out.numpy(),
                        transformers_version=np.array(transformers.__version__))
    print("clip_tiny:", tuple(out.shape), "abs mean", out.abs().mean().item(), "transformers", transformers.__version__)


# --------------------------------------------------------------------------------------------------
# weight ingestion: run the reference's own converter / LoRA-merge functions on synthetic checkpoints
# --------------------------------------------------------------------------------------------------
def _load_ref_module(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def load_ref_converter():
    """Import the reference's animatediff/utils/convert_from_ckpt.py under the in-memory diffusers scaffolding."""
    install_scaffolding()
    # stand-ins for names the converter module only imports (never calls on this path)
    dm = sys.modules["diffusers.models"]
    for n in ("AutoencoderKL", "PriorTransformer", "UNet2DConditionModel", "ControlNetModel"):
        setattr(dm, n, type(n, (), {}))
    names = ("DDIMScheduler DDPMScheduler DPMSolverMultistepScheduler EulerAncestralDiscreteScheduler EulerDiscreteScheduler "
             "HeunDiscreteScheduler LMSDiscreteScheduler PNDMScheduler UnCLIPScheduler").split()
    _mod("diffusers.schedulers", **{n: type(n, (), {}) for n in names})
    sys.modules["diffusers.utils.import_utils"].BACKENDS_MAPPING = {}
    sys.modules["diffusers"].StableDiffusionPipeline = type("StableDiffusionPipeline", (), {})
    tv_stub = sys.modules.pop("torchvision", None)     # transformers probes torchvision with find_spec(): hide the stub meanwhile
    try:
        conv = _load_ref_module("ref_convert_from_ckpt", f"{REF}/animatediff/utils/convert_from_ckpt.py")
    finally:
        if tv_stub is not None:
            sys.modules["torchvision"] = tv_stub
    return conv


@torch.no_grad()
def gen_weights(out_dir):
    from neurons_amd import _lib
    from neurons_amd.synth import randn
    from neurons_amd.unet3d import random_state_dict, state_dict_schema
    from neurons_amd.weights import LDM_UNET_PREFIX, ldm_unet_key_map
    conv = load_ref_converter()
    lora = _load_ref_module("ref_convert_lora", f"{REF}/animatediff/utils/convert_lora_safetensor_to_diffusers.py")

    # (1) LDM -> diffusers key map of the SD-1.5 topology: feed a checkpoint of 1-element tensors carrying an id
    cfg = tiny_unet_config()
    km = ldm_unet_key_map(cfg)                       # our claim; the reference decides
    ckpt = {LDM_UNET_PREFIX + k: torch.tensor([float(i)]) for i, k in enumerate(sorted(km))}
    ids = {float(i): k for i, k in enumerate(sorted(km))}
    converted = conv.convert_ldm_unet_checkpoint(dict(ckpt), {"layers_per_block": cfg.layers_per_block, "class_embed_type": None})
    ref_map = {ids[float(v)]: k for k, v in converted.items()}

    # (2) LoRA merges on the reference U-Net (torch modules)
    sd = random_state_dict(cfg, _lib.NR_KIND_UNET3D, seed=11)
    unet = build_reference_unet(cfg, sd)
    pipe = types.SimpleNamespace(unet=unet, text_encoder=None)
    targets = ["down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_q", "up_blocks.1.attentions.2.transformer_blocks.0.attn2.to_k",
               "mid_block.attentions.0.transformer_blocks.0.ff.net.2", "down_blocks.1.attentions.1.proj_in",
               "up_blocks.3.attentions.0.transformer_blocks.0.attn2.to_out.0"]
    shapes = state_dict_schema(cfg)
    kohya = {}
    for t in targets:
        w = shapes[t + ".weight"]
        r = 4
        name = "lora_unet_" + t.replace(".", "_")
        if len(w) == 4:
            kohya[name + ".lora_down.weight"] = randn(name + ".d", (r, w[1], 1, 1), 81)
            kohya[name + ".lora_up.weight"] = randn(name + ".u", (w[0], r, 1, 1), 82)
        else:
            kohya[name + ".lora_down.weight"] = randn(name + ".d", (r, w[1]), 81)
            kohya[name + ".lora_up.weight"] = randn(name + ".u", (w[0], r), 82)
        kohya[name + ".alpha"] = torch.tensor(4.0)
    lora.convert_lora(pipe, kohya, alpha=0.8)
    dl = {}
    for t in ("down_blocks.0.attentions.0.transformer_blocks.0.attn1.processor.to_q_lora", "up_blocks.2.attentions.1.transformer_blocks.0.attn2.processor.to_out_lora",
              "down_blocks.2.motion_modules.0.temporal_transformer.transformer_blocks.0.attention_blocks.0.processor.to_v_lora"):
        base = t.replace("processor.", "").replace("_lora", "").replace("to_out", "to_out.0") + ".weight"
        w = shapes[base]
        dl[t + ".down.weight"] = randn(t + ".d", (4, w[1]), 83)
        dl[t + ".up.weight"] = randn(t + ".u", (w[0], 4), 84)
    lora.load_diffusers_lora(pipe, dl, alpha=0.7)
    merged = unet.state_dict()
    changed = {k: [float(merged[k].double().sum()), float(merged[k].double().abs().sum())] for k in merged
               if not k.endswith("pos_encoder.pe") and not torch.equal(merged[k], sd[k])}
    with open(os.path.join(out_dir, "weights.json"), "w") as f:
        json.dump({"ldm_to_diffusers": ref_map, "lora_changed_checksums": changed, "kohya_targets": targets}, f, indent=0, sort_keys=True)
    print("weights.json:", len(ref_map), "mapped keys;", len(changed), "tensors changed by LoRA; map equal to ours:", ref_map == km)


# --------------------------------------------------------------------------------------------------
# a18: the reference's OWN utils.unclip_recon (utils.py:302-350), imported from where it lies and called on a stand-in
# DiffusionEngine that is wired from the reference's sampler / denoiser / wrapper / UNetModel / Decoder classes
# (DiffusionEngine itself needs pytorch_lightning; what unclip_recon touches of it is .ema_scope(), .sampler, .denoiser,
# .model and .decode_first_stage, the last restated from models/diffusion.py:118-135 + autoencoder.py:490-494).
# The four random draws inside unclip_recon (z, uc tokens, noise, offset) are recorded by replaying the same seed.
# --------------------------------------------------------------------------------------------------
def load_ref_utils():
    """Import /root/reference/utils.py: its module-level imports of torchvision.transforms / webdataset /
    generative_models.sgm (Lightning) get non-arithmetic stand-ins; unclip_recon itself only uses torch + append_dims."""
    install_sgm_scaffolding()
    class _Inert:                       # utils.py builds a torchvision Compose/Resize at import time (:265-267, pixcorr only)
        def __call__(self, *a, **k):
            return self

        def __getattr__(self, k):
            return self

    tv = sys.modules.get("torchvision") or _mod("torchvision")
    tv.__path__ = []
    tr = _mod("torchvision.transforms")
    tr.__getattr__ = lambda k: _Inert()
    tv.transforms = tr
    if "webdataset" not in sys.modules:
        _mod("webdataset")
    base = f"{REF}/generative_models"
    for name, path in (("generative_models", base), ("generative_models.sgm", f"{base}/sgm")):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [path]
            sys.modules[name] = m
    return _load_ref_module("ref_utils", f"{REF}/utils.py")


@torch.no_grad()
def gen_unclip(out_dir, num_steps=5, seed=123):
    import contextlib
    import warnings
    from neurons_amd.sgm import sgm_random_state_dict
    from neurons_amd.vae import vae_random_state_dict
    from neurons_amd.synth import randn
    ref_utils = load_ref_utils()
    from sgm.modules.diffusionmodules.denoiser import DiscreteDenoiser
    from sgm.modules.diffusionmodules.model import Decoder
    from sgm.modules.diffusionmodules.sampling import EulerEDMSampler
    from sgm.modules.diffusionmodules.wrappers import OpenAIWrapper
    cfg, vcfg = tiny_sgm_config(), tiny_vae_config()
    net = build_reference_sgm(cfg, sgm_random_state_dict(cfg, seed=71))
    vsd = vae_random_state_dict(vcfg, seed=91)
    dec = Decoder(ch=vcfg.ch, out_ch=vcfg.out_ch, ch_mult=vcfg.ch_mult, num_res_blocks=vcfg.num_res_blocks, attn_resolutions=[],
                  in_channels=3, resolution=64, z_channels=vcfg.z_channels, attn_type="vanilla", double_z=True)
    dec.load_state_dict({k[len("decoder."):]: v for k, v in vsd.items() if k.startswith("decoder.")}, strict=True)
    dec.eval()
    pq = torch.nn.Conv2d(vcfg.embed_dim, vcfg.z_channels, 1)
    pq.load_state_dict({"weight": vsd["post_quant_conv.weight"], "bias": vsd["post_quant_conv.bias"]})
    dcfg = {"target": "sgm.modules.diffusionmodules.discretizer.LegacyDDPMDiscretization"}
    engine = types.SimpleNamespace(
        ema_scope=contextlib.nullcontext,
        sampler=EulerEDMSampler(num_steps=num_steps, discretization_config=dcfg, device="cpu",
                                guider_config={"target": "sgm.modules.diffusionmodules.guiders.VanillaCFG", "params": {"scale": 5.0}}),
        denoiser=DiscreteDenoiser(scaling_config={"target": "sgm.modules.diffusionmodules.denoiser_scaling.EpsScaling"}, num_idx=1000,
                                  discretization_config=dcfg),
        model=OpenAIWrapper(net),
        decode_first_stage=lambda z: dec(pq(z / 0.18215)))     # diffusion.py:118-135 (scale_factor 0.18215, unclip6.yaml)
    x = randn("unclip.tokens", (1, 24, cfg.context_dim), 76)              # prior_out[[i]] * mask  (recon_keyframe_neurons_enhance.py:458-462)
    vector_suffix = randn("unclip.vec", (1, cfg.adm_in_channels), 77)
    torch.manual_seed(seed)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")                                   # torch.cuda.amp.autocast on a CPU-only build: disabled, fp32
        samples = ref_utils.unclip_recon(x, engine, vector_suffix, num_samples=1, offset_noise_level=0.04, device="cpu")
    # replay the draws in the order unclip_recon makes them (utils.py:308, :318, :323, :328-330)
    torch.manual_seed(seed)
    z = torch.randn(1, 4, 96, 96)
    uc_tokens = torch.randn_like(x)
    noise = torch.randn_like(z)
    offset = torch.randn(z.shape[0])
    assert tuple(samples.shape) == (1, 3, 768, 768)
    st = 4                                                                 # keep every 4th pixel: the fixture stays < 1 MB
    np.savez_compressed(os.path.join(out_dir, "unclip_tiny.npz"), tokens=x.numpy(), vector_suffix=vector_suffix.numpy(), z=z.numpy(),
                        uc_tokens=uc_tokens.numpy(), noise=noise.numpy(), offset=offset.numpy(), num_steps=np.array(num_steps),
                        seed=np.array(seed), stride=np.array(st), samples_sub=samples[:, :, ::st, ::st].numpy().astype(np.float16),
                        samples_mean=np.array(samples.double().mean().item()), samples_sq=np.array((samples.double() ** 2).mean().item()),
                        clamped_frac=np.array(((samples == 0) | (samples == 1)).double().mean().item()))
    print("unclip_tiny: samples", tuple(samples.shape), "mean", samples.mean().item(), "std", samples.std().item(),
          "clamped frac", ((samples == 0) | (samples == 1)).double().mean().item())


# --------------------------------------------------------------------------------------------------
# a1 pinned by the reference's OWN NeuroclipsPipeline.__call__ (pipeline_neuroclips.py:321-501)
# --------------------------------------------------------------------------------------------------
class _OracleDDIM:
    """diffusers-0.11.1 DDIMScheduler call surface (the source is not in /root/reference: parity of the scheduler arithmetic stays
    UNPINNED) over the oracle's restatement, so that the reference's ``__call__`` can run."""
    order = 1
    init_noise_sigma = 1.0

    def __init__(self):
        from oracle import animatediff_oracle as O
        self.O = O
        self.config = _FrozenDict(steps_offset=1, clip_sample=False)
        self.ac = O.ddim_alphas_cumprod()

    def set_timesteps(self, n, device=None):
        self.n = n
        self.timesteps = torch.tensor(self.O.ddim_timesteps(n), dtype=torch.long)

    def scale_model_input(self, x, t):
        return x

    def add_noise(self, x, noise, timesteps):
        assert timesteps.numel() == x.shape[0] and bool((timesteps == timesteps[0]).all())
        return self.O.add_noise(x, noise, int(timesteps[0]), self.ac)

    def step(self, eps, t, x, eta=0.0):
        assert eta == 0.0
        return types.SimpleNamespace(prev_sample=self.O.ddim_step(eps, int(t), x, self.ac, self.n))


def load_reference_pipeline():
    """Import animatediff/pipelines/pipeline_neuroclips.py from where it lies.  Extra NON-arithmetic stand-ins for what it imports from
    diffusers: the DiffusionPipeline base (register_modules / device / progress_bar), scheduler class NAMES (type annotations only),
    AutoencoderKL (annotation only), deprecate, is_accelerate_available."""
    install_scaffolding()
    tv = sys.modules.pop("torchvision", None)       # transformers probes torchvision with find_spec: the stand-in has no __spec__
    from transformers import CLIPTextModel, CLIPTokenizer  # noqa: F401  (annotation-only imports of the reference file)
    if tv is not None:
        sys.modules["torchvision"] = tv

    class _Bar:
        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

        def update(self, n=1):
            pass

    class DiffusionPipeline:
        def register_modules(self, **kw):
            for k, v in kw.items():
                setattr(self, k, v)

        @property
        def device(self):
            return torch.device("cpu")

        def progress_bar(self, iterable=None, total=None):
            return _Bar()

        def set_progress_bar_config(self, **kw):
            pass

    u = sys.modules["diffusers.utils"]
    u.is_accelerate_available = lambda: False
    u.deprecate = lambda *a, **k: None
    sys.modules["diffusers.models"].AutoencoderKL = type("AutoencoderKL", (), {})
    _mod("diffusers.pipeline_utils", DiffusionPipeline=DiffusionPipeline)
    names = ["DDIMScheduler", "DPMSolverMultistepScheduler", "EulerAncestralDiscreteScheduler", "EulerDiscreteScheduler",
             "LMSDiscreteScheduler", "PNDMScheduler"]
    _mod("diffusers.schedulers", **{n: type(n, (), {}) for n in names})
    reference_classes()
    from animatediff.pipelines.pipeline_neuroclips import NeuroclipsPipeline
    return NeuroclipsPipeline


@torch.no_grad()
def gen_pipeline_call(out_dir, unet=None, ctrl=None):
    """tests/golden/a1_call.npz: inputs and outputs of the reference's own ``NeuroclipsPipeline.__call__`` on the tiny networks
    (BASELINE config 1 shapes: 8 frames, 8x8 latent, 10 DDIM steps, guidance 8.5), with the tokenizer / text encoder / VAE stand-ins of
    tests/fake_modules.py and the oracle's DDIM restatement as ``scheduler``.  The ``noise`` the call draws inside (:418) is captured by
    wrapping ``torch.randn_like``; the unused ``keylatents`` draw before it (:395-405) is checked by replaying the RNG.
    Cases: (A) low_strength 0.3, one condition frame [0]; (B) low_strength 0.0, two condition frames [0, 5]; plus the facts
    (C) low_strength 0.0 == low_strength 0.3 on case A's inputs (the F8 quirk) and (D) negative_prompt given as a str."""
    from neurons_amd import _lib
    from neurons_amd.synth import randn
    from neurons_amd.unet3d import random_state_dict
    from fake_modules import FakeTextEncoder, FakeTokenizer, FakeVAE
    Pipe = load_reference_pipeline()
    ucfg, ccfg = tiny_unet_config(), tiny_ctrl_config()
    if unet is None:
        unet = build_reference_unet(ucfg, random_state_dict(ucfg, _lib.NR_KIND_UNET3D, seed=11))
        ctrl = build_reference_ctrl(ccfg, random_state_dict(ccfg, _lib.NR_KIND_SPARSECTRL, seed=12))
    B, F, H, W, N, s = 1, 8, 8, 8, 10, 8.5
    prompt = "a person walking a dog"
    out = dict(steps=np.int64(N), guidance=np.float64(s), prompt=np.array(prompt), frames=np.int64(F))

    def run(tag, seed, low_strength, index, nframes_cond, negative_prompt=None, store=True):
        pipe = Pipe(vae=FakeVAE(), text_encoder=FakeTextEncoder(ucfg.cross_attention_dim), tokenizer=FakeTokenizer(), unet=unet,
                    scheduler=_OracleDDIM(), controlnet=ctrl)
        latents = randn(f"a1.{tag}.latents", (B, 4, F, H, W), 51)
        cimg = randn(f"a1.{tag}.cimg", (B, 4, nframes_cond, H, W), 52) * 0.18215
        drawn, traj = [], []
        real = torch.randn_like

        def spy(t, *a, **k):
            r = real(t, *a, **k)
            drawn.append(r.clone())
            return r

        torch.manual_seed(seed)
        torch.randn_like = spy
        try:
            res = pipe(prompt, video_length=F, height=H * 8, width=W * 8, num_inference_steps=N, guidance_scale=s,
                       negative_prompt=negative_prompt, latents=latents.clone(), controlnet_images=cimg.clone(),
                       controlnet_image_index=list(index), low_strength=low_strength,
                       callback=lambda i, t, lat: traj.append((i, int(t), lat.clone())), callback_steps=1)
        finally:
            torch.randn_like = real
        assert len(drawn) == 1 and len(traj) == N
        noise = drawn[0]
        # RNG order (:395-405 then :418): one unused randn of the latent shape, then the noise
        torch.manual_seed(seed)
        _keylatents = torch.randn(B, 4, F, H, W)
        assert torch.equal(torch.randn(B, 4, F, H, W), noise), "noise is the SECOND draw of the latent shape"
        videos = res.videos
        assert tuple(videos.shape) == (B, 3, F, H * 8, W * 8) and videos.dtype == torch.float32
        if store:
            out.update({f"{tag}.latents": latents.numpy(), f"{tag}.cimg": cimg.numpy(), f"{tag}.noise": noise.numpy(),
                        f"{tag}.index": np.array(index), f"{tag}.low_strength": np.float64(low_strength), f"{tag}.seed": np.int64(seed),
                        f"{tag}.final_latents": traj[-1][2].numpy(), f"{tag}.latents_after_step0": traj[0][2].numpy(),
                        f"{tag}.timesteps": np.array([t for _, t, _ in traj]),
                        f"{tag}.videos_sub": videos[:, :, :, ::8, ::8].numpy()})       # the fake VAE is 8x nearest: every 8th pixel is everything
        return traj[-1][2], videos

    fa, _ = run("A", 1234, 0.3, (0,), 1)
    run("B", 4321, 0.0, (0, 5), 2)
    fc, _ = run("A", 1234, 0.0, (0,), 1, store=False)
    out["quirk_low_strength_0_equals_0p3"] = np.array(bool(torch.equal(fa, fc)))
    fd, _ = run("A", 1234, 0.3, (0,), 1, negative_prompt="", store=False)
    out["negative_prompt_empty_str_equals_none"] = np.array(bool(torch.equal(fa, fd)))
    np.savez_compressed(os.path.join(out_dir, "a1_call.npz"), **out)
    print("a1_call:", {k: getattr(v, "shape", v) for k, v in out.items()})


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    unet, ctrl, *_ = gen_networks(out_dir)
    gen_loop(out_dir, unet, ctrl)
    gen_leaf_ops(out_dir)
    gen_leaf_wide(out_dir)
    gen_sgm(out_dir)
    gen_vae(out_dir)
    gen_clip(out_dir)
    gen_weights(out_dir)
    gen_unclip(out_dir)
    gen_pipeline_call(out_dir, unet, ctrl)
    for f in sorted(os.listdir(out_dir)):
        print(f, os.path.getsize(os.path.join(out_dir, f)) // 1024, "KiB")
