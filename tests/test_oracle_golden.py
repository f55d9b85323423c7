"""CPU: pin the oracle (oracle/animatediff_oracle.py) against golden vectors produced by the REFERENCE's own
classes (oracle/gen_golden.py, run in the build container).  fp32 vs fp32: tolerance 2e-4 relative to the
output scale (different but equivalent op orderings, e.g. fused vs split projections)."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
sys.path.insert(0, os.path.dirname(HERE))

from neurons_amd import _lib  # noqa: E402
from neurons_amd.synth import randn  # noqa: E402
from neurons_amd.unet3d import random_state_dict  # noqa: E402
from oracle import animatediff_oracle as O  # noqa: E402
from tiny_configs import tiny_ctrl_config, tiny_unet_config  # noqa: E402


def _close(name, got, want, tol=2e-4):
    got = got.detach().float()
    want = torch.as_tensor(want).float()
    assert got.shape == want.shape, (name, got.shape, want.shape)
    err = (got - want).abs().max().item()
    scale = want.abs().max().item()
    assert err <= tol * scale + 1e-6, f"{name}: max err {err:.3e} vs scale {scale:.3e}"


@pytest.fixture(scope="module")
def tiny():
    g = np.load(os.path.join(GOLD, "tiny_networks.npz"))
    ucfg, ccfg = tiny_unet_config(), tiny_ctrl_config()
    usd = random_state_dict(ucfg, _lib.NR_KIND_UNET3D, seed=11)
    csd = random_state_dict(ccfg, _lib.NR_KIND_SPARSECTRL, seed=12)
    return g, O.OracleConfig.from_native(ucfg), O.OracleConfig.from_native(ccfg), usd, csd


def test_synth_inputs_regenerate(tiny):
    g = tiny[0]
    assert np.array_equal(randn("sample", (2, 4, 8, 8, 8), 21).numpy(), g["sample"])
    assert np.array_equal(randn("ctx", (2, 77, 64), 22).numpy(), g["ctx"])


@torch.no_grad()
def test_unet_forward_matches_reference(tiny):
    g, ucfg, _, usd, _ = tiny
    taps = {}
    eps = O.unet3d_forward(usd, ucfg, torch.from_numpy(g["sample"]), int(g["t"]), torch.from_numpy(g["ctx"]), taps=taps)
    _close("eps_plain", eps, g["eps_plain"])
    for key in g.files:
        if key.startswith("tap_idx:"):
            n = key.split(":", 1)[1]
            assert tuple(taps[n].shape) == tuple(g[f"tap_shape:{n}"])
            got = taps[n].reshape(-1)[torch.from_numpy(g[key])]
            _close(f"tap {n}", got, g[f"tap_val:{n}"])


@torch.no_grad()
def test_sparsectrl_and_residual_path_match_reference(tiny):
    g, ucfg, ccfg, usd, csd = tiny
    sample, ctx = torch.from_numpy(g["sample"]), torch.from_numpy(g["ctx"])
    down, mid = O.sparse_controlnet_forward(csd, ccfg, sample, int(g["t"]), ctx, torch.from_numpy(g["cond"]),
                                            torch.from_numpy(g["mask"]), 1.0)
    assert len(down) == 12
    for i, d in enumerate(down):
        _close(f"down_res_{i}", d, g[f"down_res_{i}"])
    _close("mid_res", mid, g["mid_res"])
    eps = O.unet3d_forward(usd, ucfg, sample, int(g["t"]), ctx, down, mid)
    _close("eps_ctrl", eps, g["eps_ctrl"])


@torch.no_grad()
def test_c1_loop_matches_reference(tiny):
    """BASELINE config 1 (8 f, 8x8 latent, 10 DDIM steps, CFG 8.5): reference networks + restated scheduler."""
    _, ucfg, ccfg, usd, csd = tiny
    g = np.load(os.path.join(GOLD, "c1_loop.npz"))
    final, eps_log = O.neuroclips_denoise(usd, ucfg, csd, ccfg, torch.from_numpy(g["latents"]), torch.from_numpy(g["noise"]),
                                          torch.from_numpy(g["ctx"]), torch.from_numpy(g["cimg"]), (0,), int(g["steps"]),
                                          float(g["guidance"]), 1.0, return_eps_steps=(0, 1, int(g["steps"]) - 1))
    assert O.ddim_timesteps(int(g["steps"])) == list(g["timesteps"])
    _close("eps step 0", eps_log[0], g["eps_step_0"])
    _close("eps step 1", eps_log[1], g["eps_step_1"], tol=1e-3)
    _close("final latents", final, g["final"], tol=5e-3)   # 10 chained fp32 evaluations


def _fill(prefix_sd_shapes, tag, seed):
    sd = {}
    for k, shape in prefix_sd_shapes.items():
        z = randn(f"{tag}.{k}", shape, seed)
        if len(shape) == 1:
            z = (1.0 + 0.1 * z) if k.endswith("weight") else 0.05 * z
        else:
            z = z / (int(np.prod(shape[1:])) ** 0.5)
        sd[k] = z
    return sd


@torch.no_grad()
def test_leaf_ops_match_reference():
    g = np.load(os.path.join(GOLD, "leaf_ops.npz"))
    for C in (320, 640, 1280):   # head dims 40 / 80 / 160
        shp = {"to_q.weight": (C, C), "to_k.weight": (C, C), "to_v.weight": (C, C), "to_out.0.weight": (C, C), "to_out.0.bias": (C,)}
        sd = {f"a.{k}": v for k, v in _fill(shp, f"attn{C}", 41).items()}
        x = randn(f"attn{C}.x", (2, 48, C), 42)
        _close(f"selfattn{C}", O.cross_attention(sd, "a", x, None, 8), g[f"selfattn{C}.y"])
        shp = {"to_q.weight": (C, C), "to_k.weight": (C, 768), "to_v.weight": (C, 768), "to_out.0.weight": (C, C), "to_out.0.bias": (C,)}
        sd = {f"a.{k}": v for k, v in _fill(shp, f"xattn{C}", 43).items()}
        ctx = randn(f"xattn{C}.ctx", (2, 77, 768), 44)
        _close(f"crossattn{C}", O.cross_attention(sd, "a", x, ctx, 8), g[f"crossattn{C}.y"])
    shp = {"net.0.proj.weight": (2560, 320), "net.0.proj.bias": (2560,), "net.2.weight": (320, 1280), "net.2.bias": (320,)}
    sd = {f"f.{k}": v for k, v in _fill(shp, "ff320", 45).items()}
    _close("ff320", O.feed_forward(sd, "f", randn("ff320.x", (2, 48, 320), 46)), g["ff320.y"])
    # VersatileAttention with positional encoding, F = 16
    shp = {"to_q.weight": (320, 320), "to_k.weight": (320, 320), "to_v.weight": (320, 320), "to_out.0.weight": (320, 320), "to_out.0.bias": (320,)}
    sd = {f"v.{k}": v for k, v in _fill(shp, "va320", 47).items()}
    pe = O.positional_encoding_table(320, 24, "cpu")
    _close("va320", O.versatile_attention(sd, "v", randn("va320.x", (32, 6, 320), 48), 16, 8, pe), g["va320.y"])
    # ResnetBlock3D 2560 -> 1280 at 4x4 (the widest conv on the path)
    shp = {"norm1.weight": (2560,), "norm1.bias": (2560,), "conv1.weight": (1280, 2560, 3, 3), "conv1.bias": (1280,),
           "time_emb_proj.weight": (1280, 1280), "time_emb_proj.bias": (1280,), "norm2.weight": (1280,), "norm2.bias": (1280,),
           "conv2.weight": (1280, 1280, 3, 3), "conv2.bias": (1280,), "conv_shortcut.weight": (1280, 2560, 1, 1),
           "conv_shortcut.bias": (1280,)}
    sd = {f"r.{k}": v for k, v in _fill(shp, "rb2560", 49).items()}
    y = O.resnet_block3d(sd, "r", randn("rb2560.x", (1, 2560, 2, 4, 4), 50), randn("rb2560.temb", (1, 1280), 51), 32, 1e-5)
    _close("rb2560", y, g["rb2560.y"])
    # up / down sample
    sd = {f"u.{k}": v for k, v in _fill({"conv.weight": (64, 64, 3, 3), "conv.bias": (64,)}, "up64", 57).items()}
    _close("up64", O.upsample3d(sd, "u", randn("up64.x", (1, 64, 2, 3, 5), 58)), g["up64.y"])
    sd = {f"d.{k}": v for k, v in _fill({"conv.weight": (64, 64, 3, 3), "conv.bias": (64,)}, "dn64", 59).items()}
    _close("dn64", O.downsample3d(sd, "d", randn("dn64.x", (1, 64, 2, 6, 10), 60)), g["dn64.y"])


@torch.no_grad()
def test_composite_modules_match_reference():
    """Transformer3DModel and VanillaTemporalModule at full width 320 (needs the schema helpers for key names)."""
    from neurons_amd.unet3d import _motion_keys, _transformer_keys
    g = np.load(os.path.join(GOLD, "leaf_ops.npz"))
    shp = {k[len("m."):]: v for k, v in _motion_keys("m", 320, 2).items()}
    sd = {f"m.{k}": v for k, v in _fill(shp, "tm320", 52).items()}
    y = O.temporal_transformer3d(sd, "m", randn("tm320.x", (1, 320, 16, 3, 3), 53), 8, 32, 2, 24)
    _close("tm320", y, g["tm320.y"])
    shp = {k[len("t."):]: v for k, v in _transformer_keys("t", 320, 768).items()}
    sd = {f"t.{k}": v for k, v in _fill(shp, "t3d320", 54).items()}
    y = O.transformer3d(sd, "t", randn("t3d320.x", (1, 320, 2, 4, 4), 55), randn("t3d320.ctx", (1, 77, 768), 56), 8, 32)
    _close("t3d320", y, g["t3d320.y"])
    # the same modules at the row counts where the engine runs its fused kernels (4096 / 4608 rows); the GPU twin is tests/test_leaf_gpu.py
    shp = {k[len("m."):]: v for k, v in _motion_keys("m", 320, 2).items()}
    sd = {f"m.{k}": v for k, v in _fill(shp, "tm320big", 61).items()}
    y = O.temporal_transformer3d(sd, "m", randn("tm320big.x", (1, 320, 16, 16, 16), 62), 8, 32, 2, 24)
    assert tuple(y.shape) == tuple(g["tm320big.shape"])
    _close("tm320big", y.reshape(-1)[torch.from_numpy(g["tm320big.idx"])], g["tm320big.val"])
    # 32-frame clips (BASELINE config 5): the two-row-tile form of the fused temporal-attention kernel
    sd = {f"m.{k}": v for k, v in _fill({k[len("m."):]: v for k, v in _motion_keys("m", 320, 2).items()}, "tm320f32", 66).items()}
    y = O.temporal_transformer3d(sd, "m", randn("tm320f32.x", (1, 320, 32, 12, 12), 67), 8, 32, 2, 32)
    assert tuple(y.shape) == tuple(g["tm320f32.shape"])
    _close("tm320f32", y.reshape(-1)[torch.from_numpy(g["tm320f32.idx"])], g["tm320f32.val"])
    shp = {k[len("t."):]: v for k, v in _transformer_keys("t", 320, 768).items()}
    sd = {f"t.{k}": v for k, v in _fill(shp, "t3d320big", 63).items()}
    y = O.transformer3d(sd, "t", randn("t3d320big.x", (1, 320, 2, 48, 48), 64), randn("t3d320big.ctx", (1, 77, 768), 65), 8, 32)
    assert tuple(y.shape) == tuple(g["t3d320big.shape"])
    _close("t3d320big", y.reshape(-1)[torch.from_numpy(g["t3d320big.idx"])], g["t3d320big.val"])


@torch.no_grad()
def test_oracle_wide_leaf_modules_match_reference_golden():
    """C = 640 (d = 80) and C = 1280 (d = 160) composite modules at the shapes BASELINE config 2 runs them (8192 / 2048 rows):
    tests/golden/leaf_wide.npz from the reference classes (oracle/gen_golden.py: gen_leaf_wide); the GPU twin is tests/test_leaf_gpu.py."""
    from neurons_amd.unet3d import _motion_keys, _transformer_keys
    g = np.load(os.path.join(GOLD, "leaf_wide.npz"))
    for C, hw, seed in ((640, 16, 71), (1280, 8, 75)):
        sd = {f"m.{k}": v for k, v in _fill({k[len("m."):]: v for k, v in _motion_keys("m", C, 2).items()}, f"tm{C}", seed).items()}
        y = O.temporal_transformer3d(sd, "m", randn(f"tm{C}.x", (2, C, 16, hw, hw), seed + 1), 8, 32, 2, 24)
        assert tuple(y.shape) == tuple(g[f"tm{C}.shape"])
        _close(f"tm{C}", y.reshape(-1)[torch.from_numpy(g[f"tm{C}.idx"])], g[f"tm{C}.val"])
        sd = {f"t.{k}": v for k, v in _fill({k[len("t."):]: v for k, v in _transformer_keys("t", C, 768).items()}, f"t3d{C}", seed + 2).items()}
        y = O.transformer3d(sd, "t", randn(f"t3d{C}.x", (2, C, 16, hw, hw), seed + 3), randn(f"t3d{C}.ctx", (2, 77, 768), seed + 4), 8, 32)
        assert tuple(y.shape) == tuple(g[f"t3d{C}.shape"])
        _close(f"t3d{C}", y.reshape(-1)[torch.from_numpy(g[f"t3d{C}.idx"])], g[f"t3d{C}.val"])


@torch.no_grad()
def test_sliced_attention_of_the_oracle_is_value_identical(tiny, monkeypatch):
    """The oracle evaluates softmax(q k^T) v a slice of (batch * heads) at a time when the fp32 score tensor would exceed its budget
    (BASELINE config 5: 34 GB).  Forcing one problem per slice must reproduce the reference golden exactly as the unsliced path does."""
    g, ucfg, _, usd, _ = tiny
    sample, ctx = torch.from_numpy(g["sample"]), torch.from_numpy(g["ctx"])
    a = O.unet3d_forward(usd, ucfg, sample, int(g["t"]), ctx)
    monkeypatch.setattr(O, "ATTN_SCORE_BUDGET_BYTES", 1)
    b = O.unet3d_forward(usd, ucfg, sample, int(g["t"]), ctx)
    _close("eps_plain (sliced attention)", b, g["eps_plain"])
    assert (a - b).abs().max().item() <= 1e-6 * a.abs().max().item()


@torch.no_grad()
def test_sparsectrl_frames_without_condition_are_identical_before_the_first_motion_module(tiny):
    """The premise of the engine's identical-frame evaluation (nr_sparsectrl_set_condition_frames), on the reference-pinned oracle: with the
    noisy sample zeroed, every frame whose condition and mask are zero carries THE SAME activations through conv_in + cond embedding,
    down_blocks[0].resnets[0] and attentions[0]; the first motion module is what makes them differ (sparse_controlnet.py:468-469,513-545;
    unet_blocks.py:382-421)."""
    g, _, ccfg, _, csd = tiny
    sample, ctx = torch.from_numpy(g["sample"]), torch.from_numpy(g["ctx"])
    cond, mask = torch.from_numpy(g["cond"]), torch.from_numpy(g["mask"])          # condition on frame 0 only
    taps = {}
    O.sparse_controlnet_forward(csd, ccfg, sample, int(g["t"]), ctx, cond, mask, 1.0, taps=taps)
    assert "down_blocks.0.attentions.0" in taps and "down_blocks.0.motion_modules.0" in taps, sorted(taps)[:8]
    for name in ("conv_in", "down_blocks.0.resnets.0", "down_blocks.0.attentions.0"):
        t = taps[name]                                       # b c f h w
        rest = t[:, :, 1:]
        spread = (rest - rest[:, :, :1]).abs().max().item()
        assert spread <= 1e-6 * t.abs().max().item(), (name, spread)
        assert (t[:, :, 0] - t[:, :, 1]).abs().max().item() > 1e-3 * t.abs().max().item(), name     # the conditioned frame does differ
    mm = taps["down_blocks.0.motion_modules.0"]
    assert (mm[:, :, 1:] - mm[:, :, 1:2]).abs().max().item() > 1e-4 * mm.abs().max().item()       # temporal attention mixes the frames
