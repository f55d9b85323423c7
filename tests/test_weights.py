"""CPU: weight ingestion (neurons_amd/weights.py) against the reference's own converter / LoRA-merge functions
(results recorded in tests/golden/weights.json by oracle/gen_golden.py: gen_weights)."""
import json
import os
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from neurons_amd import _lib, NativeUNet3D  # noqa: E402
from neurons_amd.synth import randn  # noqa: E402
from neurons_amd.unet3d import UNet3DConfig, random_state_dict, state_dict_schema  # noqa: E402
from neurons_amd import weights as W  # noqa: E402
from oracle.gen_golden import tiny_unet_config  # noqa: E402

GOLD = json.load(open(os.path.join(HERE, "golden", "weights.json")))


def test_ldm_key_map_equals_reference_converter():
    assert W.ldm_unet_key_map(tiny_unet_config()) == GOLD["ldm_to_diffusers"]
    full = W.ldm_unet_key_map(UNet3DConfig())
    sch = state_dict_schema(UNet3DConfig())
    assert len(full) == sum(1 for k in sch if "motion_modules." not in k) == 686      # SD-1.5 U-Net tensors
    assert full["input_blocks.3.0.op.weight"] == "down_blocks.0.downsamplers.0.conv.weight"
    assert full["output_blocks.2.1.conv.bias"] == "up_blocks.0.upsamplers.0.conv.bias"
    assert full["output_blocks.5.2.conv.weight"] == "up_blocks.1.upsamplers.0.conv.weight"
    assert full["middle_block.1.transformer_blocks.0.attn2.to_k.weight"] == "mid_block.attentions.0.transformer_blocks.0.attn2.to_k.weight"


def test_convert_ldm_checkpoint_roundtrip():
    cfg = tiny_unet_config()
    sd = random_state_dict(cfg, seed=3)
    km = W.ldm_unet_key_map(cfg)
    ckpt = {"model.diffusion_model." + old: sd[new] for old, new in km.items()}
    ckpt["first_stage_model.decoder.conv_in.weight"] = torch.zeros(1)            # non-U-Net entries are ignored
    k = "model.diffusion_model.input_blocks.1.1.proj_in.weight"                  # some checkpoints store linear proj_in
    ckpt[k] = ckpt[k].reshape(ckpt[k].shape[0], -1)
    out = W.convert_ldm_unet_checkpoint(ckpt, cfg)
    assert set(out) == set(km.values())
    assert all(torch.equal(out[k], sd[k]) for k in out)


def _lora_inputs(cfg):
    shapes = state_dict_schema(cfg)
    kohya = {}
    for t in GOLD["kohya_targets"]:
        w = shapes[t + ".weight"]
        name = "lora_unet_" + t.replace(".", "_")
        if len(w) == 4:
            kohya[name + ".lora_down.weight"] = randn(name + ".d", (4, w[1], 1, 1), 81)
            kohya[name + ".lora_up.weight"] = randn(name + ".u", (w[0], 4, 1, 1), 82)
        else:
            kohya[name + ".lora_down.weight"] = randn(name + ".d", (4, w[1]), 81)
            kohya[name + ".lora_up.weight"] = randn(name + ".u", (w[0], 4), 82)
        kohya[name + ".alpha"] = torch.tensor(4.0)
    kohya["lora_te_text_model_encoder_layers_0_self_attn_k_proj.lora_down.weight"] = torch.zeros(4, 8)
    dl = {}
    for t in ("down_blocks.0.attentions.0.transformer_blocks.0.attn1.processor.to_q_lora",
              "up_blocks.2.attentions.1.transformer_blocks.0.attn2.processor.to_out_lora",
              "down_blocks.2.motion_modules.0.temporal_transformer.transformer_blocks.0.attention_blocks.0.processor.to_v_lora"):
        base = t.replace("processor.", "").replace("_lora", "").replace("to_out", "to_out.0") + ".weight"
        w = shapes[base]
        dl[t + ".down.weight"] = randn(t + ".d", (4, w[1]), 83)
        dl[t + ".up.weight"] = randn(t + ".u", (w[0], 4), 84)
    return kohya, dl


def test_lora_merges_equal_reference():
    cfg = tiny_unet_config()
    sd = random_state_dict(cfg, _lib.NR_KIND_UNET3D, seed=11)
    net = NativeUNet3D(cfg)
    net.load_state_dict(sd)
    kohya, dl = _lora_inputs(cfg)
    deltas, skipped = W.kohya_lora_deltas(kohya, cfg, alpha=0.8)
    assert skipped == ["lora_te_text_model_encoder_layers_0_self_attn_k_proj.lora_down.weight"]
    W.apply_deltas(net, deltas)
    W.apply_deltas(net, W.diffusers_lora_deltas(dl, 0.7))
    changed = {k for k in sd if not torch.equal(net._pending[k].float(), sd[k])}
    assert changed == set(GOLD["lora_changed_checksums"])
    for k, (s, a) in GOLD["lora_changed_checksums"].items():
        got = net._pending[k].double()
        assert abs(float(got.sum()) - s) <= 1e-4 * max(1.0, abs(a)) and abs(float(got.abs().sum()) - a) <= 1e-5 * a, k
    with pytest.raises(KeyError):
        W.kohya_lora_deltas({"lora_unet_nonexistent_layer.lora_down.weight": torch.zeros(1, 1)}, cfg)


def test_motion_module_filter():
    sd = {"state_dict": {"down_blocks.0.motion_modules.0.temporal_transformer.proj_in.weight": torch.zeros(1),
                         "down_blocks.0.motion_modules.0.temporal_transformer.transformer_blocks.0.attention_blocks.0.pos_encoder.pe": torch.zeros(1),
                         "conv_in.weight": torch.zeros(1)}}
    assert list(W.filter_motion_module(sd)) == ["down_blocks.0.motion_modules.0.temporal_transformer.proj_in.weight"]
