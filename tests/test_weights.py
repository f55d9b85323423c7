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
from tiny_configs import tiny_unet_config  # noqa: E402

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


def test_from_pretrained_2d_local_directory(tmp_path):
    """UNet3DConditionModel.from_pretrained_2d (unet.py:477-572) on a local SD-style directory: 2-D config.json + safetensors
    weights; the motion-module keys are the only missing ones; unknown / unsupported options are refused."""
    import json
    from safetensors.torch import save_file
    from neurons_amd import NativeUNet3D
    from neurons_amd.unet3d import random_state_dict, state_dict_schema
    from tiny_configs import tiny_unet_config
    cfg = tiny_unet_config()
    sd = random_state_dict(cfg, seed=5)
    sd2d = {k: v.contiguous() for k, v in sd.items() if "motion_modules." not in k}
    root = os.path.join(tmp_path, "sd", "unet")
    os.makedirs(root)
    save_file(sd2d, os.path.join(root, "diffusion_pytorch_model.safetensors"))
    conf = {"_class_name": "UNet2DConditionModel", "_diffusers_version": "0.6.0", "act_fn": "silu", "attention_head_dim": cfg.attention_head_dim,
            "block_out_channels": list(cfg.block_out_channels), "center_input_sample": False, "cross_attention_dim": cfg.cross_attention_dim,
            "down_block_types": ["CrossAttnDownBlock2D"] * 3 + ["DownBlock2D"], "downsample_padding": 1, "flip_sin_to_cos": True, "freq_shift": 0,
            "in_channels": 4, "layers_per_block": 2, "mid_block_scale_factor": 1, "norm_eps": 1e-05, "norm_num_groups": cfg.norm_num_groups,
            "out_channels": 4, "sample_size": 64, "up_block_types": ["UpBlock2D"] + ["CrossAttnUpBlock2D"] * 3}
    with open(os.path.join(root, "config.json"), "w") as f:
        json.dump(conf, f)
    extra = dict(use_inflated_groupnorm=True, use_motion_module=True, motion_module_resolutions=[1, 2, 4, 8], motion_module_mid_block=False,
                 motion_module_type="Vanilla", motion_module_kwargs=dict(cfg.motion_module_kwargs))
    unet = NativeUNet3D.from_pretrained_2d(os.path.join(tmp_path, "sd"), subfolder="unet", unet_additional_kwargs=extra)
    assert unet.config.down_block_types == ("CrossAttnDownBlock3D",) * 3 + ("DownBlock3D",) and unet.config.sample_size == 64
    schema = state_dict_schema(cfg)
    assert set(unet._pending) == set(sd2d) and all("motion_modules." in k for k in schema if k not in unet._loaded)
    # the motion checkpoint then completes it (load_weights path, util.py:106-121)
    missing, unexpected = unet.load_state_dict({k: v for k, v in sd.items() if "motion_modules." in k}, strict=False)
    assert not missing and not unexpected
    conf["dual_cross_attention"] = True
    with open(os.path.join(root, "config.json"), "w") as f:
        json.dump(conf, f)
    with pytest.raises(NotImplementedError, match="dual_cross_attention"):
        NativeUNet3D.from_pretrained_2d(os.path.join(tmp_path, "sd"), subfolder="unet", unet_additional_kwargs=extra)


def test_load_weights_merges_text_encoder_lora_and_refuses_silent_dreambooth_skips(tmp_path):
    """the reference's convert_lora merges the lora_te_* deltas into pipeline.text_encoder (:66-68,99-107) and its
    DreamBooth branch always replaces VAE and text encoder (util.py:137-144).  load_weights must do the same or fail loudly."""
    import types
    from safetensors.torch import save_file
    from neurons_amd.clip import NativeCLIPTextModel, clip_random_state_dict
    from tiny_configs import tiny_clip_config, tiny_unet_config
    cfg, ccfg = tiny_unet_config(), tiny_clip_config()
    sd = random_state_dict(cfg, _lib.NR_KIND_UNET3D, seed=11)
    tsd = clip_random_state_dict(ccfg, seed=97)
    unet = NativeUNet3D(cfg)
    unet.load_state_dict(sd)
    te = NativeCLIPTextModel(ccfg)
    te.load_state_dict(tsd)
    H = ccfg.hidden_size
    name = "lora_te_text_model_encoder_layers_1_self_attn_q_proj"
    g = torch.Generator().manual_seed(3)
    lora = {name + ".lora_down.weight": torch.randn(4, H, generator=g), name + ".lora_up.weight": torch.randn(H, 4, generator=g),
            name + ".alpha": torch.tensor(4.0)}
    path = os.path.join(tmp_path, "l.safetensors")
    save_file(lora, path)
    pipe = types.SimpleNamespace(unet=unet, text_encoder=te, vae=None)
    W.load_weights(pipe, lora_model_path=path, lora_alpha=0.8)
    key = "text_model.encoder.layers.1.self_attn.q_proj.weight"
    want = tsd[key] + 0.8 * lora[name + ".lora_up.weight"] @ lora[name + ".lora_down.weight"]
    assert torch.allclose(te._pending[key].float(), want, atol=1e-6)
    # a torch text encoder gets the same in-place merge as the reference performs
    lin = torch.nn.Module()
    lin.text_model = torch.nn.Module(); lin.text_model.encoder = torch.nn.Module(); lin.text_model.encoder.layers = torch.nn.ModuleList(
        [torch.nn.Module(), torch.nn.Module()])
    lin.text_model.encoder.layers[1].self_attn = torch.nn.Module()
    lin.text_model.encoder.layers[1].self_attn.q_proj = torch.nn.Linear(H, H, bias=False)
    w0 = lin.text_model.encoder.layers[1].self_attn.q_proj.weight.detach().clone()
    unet2 = NativeUNet3D(cfg)
    unet2.load_state_dict(sd)
    W.load_weights(types.SimpleNamespace(unet=unet2, text_encoder=lin, vae=None), lora_model_path=path, lora_alpha=0.8)
    assert torch.allclose(lin.text_model.encoder.layers[1].self_attn.q_proj.weight, w0 + (want - tsd[key]), atol=1e-6)
    with pytest.raises(ValueError, match="text-encoder tensors"):
        unet3 = NativeUNet3D(cfg)
        unet3.load_state_dict(sd)
        W.load_weights(types.SimpleNamespace(unet=unet3, text_encoder=None, vae=None), lora_model_path=path)
    # DreamBooth: a PyTorch VAE without vae_converter must not be skipped silently
    ck = os.path.join(tmp_path, "db.safetensors")
    save_file({"model.diffusion_model.dummy": torch.zeros(1)}, ck)
    with pytest.raises(ValueError, match="replaces the VAE"):
        W.load_weights(types.SimpleNamespace(unet=unet, text_encoder=te, vae=torch.nn.Identity()), dreambooth_model_path=ck)
