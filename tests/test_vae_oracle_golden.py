"""CPU: pin oracle/vae_oracle.py and the VAE host logic against vectors produced by the reference's own sgm Decoder
and its convert_ldm_vae_checkpoint (oracle/gen_golden.py: gen_vae)."""
import json
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
GOLD = os.path.join(HERE, "golden", "vae_tiny.npz")

from neurons_amd.vae import (VAEDecoderConfig, convert_diffusers_vae_state_dict, diffusers_vae_key_map,  # noqa: E402
                             vae_decoder_state_dict_schema, vae_random_state_dict, NativeVAEDecoder)
from oracle import vae_oracle as V  # noqa: E402
from tiny_configs import tiny_vae_config  # noqa: E402


def _close(name, got, want, tol=2e-4):
    got, want = got.detach().float(), torch.as_tensor(want).float()
    assert got.shape == want.shape
    err, scale = (got - want).abs().max().item(), want.abs().max().item()
    assert err <= tol * scale + 1e-6, f"{name}: {err:.3e} vs scale {scale:.3e}"


@torch.no_grad()
def test_decode_first_stage_matches_reference():
    g = np.load(GOLD)
    cfg = tiny_vae_config()
    sd = vae_random_state_dict(cfg, seed=91)
    img = V.decode_first_stage(sd, torch.from_numpy(g["z"]), len(cfg.ch_mult), cfg.num_res_blocks)
    _close("image", img, g["image"])


@torch.no_grad()
def test_decode_latents_matches_reference():
    g = np.load(GOLD)
    cfg = tiny_vae_config()
    sd = vae_random_state_dict(cfg, seed=91)
    vid = V.decode_latents(sd, torch.from_numpy(g["lat"]), len(cfg.ch_mult), cfg.num_res_blocks)
    _close("video", vid, g["video"])
    assert float(((g["video"] == 0) | (g["video"] == 1)).mean()) > 0.05      # the clamp is exercised


@torch.no_grad()
def test_encoder_moments_and_sampling_match_reference():
    g = np.load(GOLD)
    cfg = tiny_vae_config()
    esd = vae_random_state_dict(cfg, seed=94, encoder=True)
    m = V.encode_moments(esd, 2 * torch.from_numpy(g["enc_img"]) - 1, len(cfg.ch_mult), cfg.num_res_blocks)
    _close("moments", m, g["enc_moments"])
    _close("sample", V.gaussian_sample(torch.from_numpy(g["enc_moments"]), torch.from_numpy(g["enc_noise"])) * 0.18215, g["enc_sample"], tol=1e-6)
    _close("mode", torch.from_numpy(g["enc_moments"])[:, :4] * 0.18215, g["enc_mode"], tol=1e-6)


def test_schema_is_the_sd_vae_decoder():
    sch = vae_decoder_state_dict_schema(VAEDecoderConfig())
    n = sum(int(np.prod(s)) for s in sch.values())
    assert len(sch) == 140 and n == 49_490_199          # SD-1.5 / unclip6.yaml first-stage decoder + post_quant_conv
    assert sch["decoder.mid.attn_1.q.weight"] == (512, 512, 1, 1) and sch["decoder.up.1.block.0.nin_shortcut.weight"] == (256, 512, 1, 1)
    from neurons_amd.vae import vae_encoder_state_dict_schema
    esch = vae_encoder_state_dict_schema(VAEDecoderConfig())
    assert len(esch) == 108 and sum(int(np.prod(s)) for s in esch.values()) == 34_163_664
    assert esch["encoder.conv_out.weight"] == (8, 512, 3, 3) and esch["quant_conv.weight"] == (8, 8, 1, 1)


def test_diffusers_key_map_matches_reference_converter():
    with open(os.path.join(HERE, "golden", "vae_keys.json")) as f:
        ref = json.load(f)["diffusers_to_ldm"]
    cfg = VAEDecoderConfig()
    with open(os.path.join(HERE, "golden", "vae_keys.json")) as f:
        assert diffusers_vae_key_map(cfg, encoder=True) == json.load(f)["diffusers_to_ldm_encoder"]
    assert diffusers_vae_key_map(cfg) == ref
    # a diffusers-named state dict (Linear attention weights, conv_shortcut) converts to loadable first-stage names
    sch = vae_decoder_state_dict_schema(cfg)
    dsd = {}
    for dk, lk in ref.items():
        shape = sch[lk]
        if "attentions.0" in dk and dk.endswith(".weight") and len(shape) == 4:
            shape = shape[:2]
        dsd[dk] = torch.zeros(shape)
    dsd["encoder.conv_in.weight"] = torch.zeros(1)
    out = convert_diffusers_vae_state_dict(dsd, cfg)
    assert set(out) == set(sch) and all(tuple(out[k].shape) == tuple(sch[k]) for k in sch)


def test_product_path_refuses_cpu_and_bad_variants():
    dec = NativeVAEDecoder(tiny_vae_config())
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        dec.decode(torch.zeros(1, 4, 8, 8))
    with pytest.raises(RuntimeError, match="MI355X only"):
        dec.to("cpu")
    with pytest.raises(NotImplementedError):
        NativeVAEDecoder(VAEDecoderConfig(attn_resolutions=(32,)))
    missing, unexpected = dec.load_state_dict({"encoder.conv_in.weight": torch.zeros(1)}, strict=False)
    assert "decoder.conv_in.weight" in missing and unexpected == ["encoder.conv_in.weight"]
