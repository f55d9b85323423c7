"""Weight ingestion for the native networks — what ``animatediff/utils/util.py:92-185`` (``load_weights``) does to
``pipeline.unet`` / ``.vae`` / ``.text_encoder``, expressed on state dicts because the native U-Net has no torch
sub-modules to mutate:

  * motion-module checkpoint filter                     util.py:106-121
  * DreamBooth / LDM checkpoint -> diffusers key names  convert_from_ckpt.py:328-556 (``convert_ldm_unet_checkpoint``)
  * kohya LoRA folded into the base weights             convert_lora_safetensor_to_diffusers.py:50-112 (``convert_lora``)
  * diffusers-style (domain adapter / motion) LoRA      convert_lora_safetensor_to_diffusers.py:27-47 (``load_diffusers_lora``)

LoRA has no runtime cost: ``W += alpha * up @ down`` is applied to the host copy before the engine converts it.
Everything here is host-side dictionary work (CPU tensors), pinned by tests/test_weights.py against the reference's
own functions (key map and merged tensors recorded in tests/golden/weights.json by oracle/gen_golden.py).
"""
from typing import Dict, Iterable, List, Optional, Tuple

import torch

from .unet3d import UNet3DConfig, state_dict_schema

LDM_UNET_PREFIX = "model.diffusion_model."

_RESNET = (("in_layers.0", "norm1"), ("in_layers.2", "conv1"), ("emb_layers.1", "time_emb_proj"), ("out_layers.0", "norm2"),
           ("out_layers.3", "conv2"), ("skip_connection", "conv_shortcut"))


def ldm_unet_key_map(cfg: UNet3DConfig) -> Dict[str, str]:
    """LDM (``model.diffusion_model.`` stripped) -> diffusers names for the IMAGE layers of the SD-1.5 U-Net.
    Block numbering: input_blocks 0 = conv_in, then per level ``layers_per_block`` res(+attn) blocks and one
    downsample; output_blocks per level ``layers_per_block + 1`` blocks with the upsample appended to the last."""
    sch = state_dict_schema(cfg)
    lpb, L = cfg.layers_per_block, len(cfg.block_out_channels)
    m = {"time_embed.0": "time_embedding.linear_1", "time_embed.2": "time_embedding.linear_2", "input_blocks.0.0": "conv_in",
         "out.0": "conv_norm_out", "out.2": "conv_out"}
    for i in range(1, L * (lpb + 1)):
        b, j = (i - 1) // (lpb + 1), (i - 1) % (lpb + 1)
        if j == lpb:
            m[f"input_blocks.{i}.0.op"] = f"down_blocks.{b}.downsamplers.0.conv"
            continue
        for old, new in _RESNET:
            m[f"input_blocks.{i}.0.{old}"] = f"down_blocks.{b}.resnets.{j}.{new}"
        if cfg.down_block_types[b] == "CrossAttnDownBlock3D":
            m[f"input_blocks.{i}.1"] = f"down_blocks.{b}.attentions.{j}"
    for old, new in _RESNET:
        m[f"middle_block.0.{old}"] = f"mid_block.resnets.0.{new}"
        m[f"middle_block.2.{old}"] = f"mid_block.resnets.1.{new}"
    m["middle_block.1"] = "mid_block.attentions.0"
    for i in range(L * (lpb + 1)):
        b, j = i // (lpb + 1), i % (lpb + 1)
        for old, new in _RESNET:
            m[f"output_blocks.{i}.0.{old}"] = f"up_blocks.{b}.resnets.{j}.{new}"
        attn = cfg.up_block_types[b] == "CrossAttnUpBlock3D"
        if attn:
            m[f"output_blocks.{i}.1"] = f"up_blocks.{b}.attentions.{j}"
        if j == lpb and b != L - 1:
            m[f"output_blocks.{i}.{2 if attn else 1}.conv"] = f"up_blocks.{b}.upsamplers.0.conv"
    # expand module prefixes to full parameter names that exist in the schema
    out = {}
    prefixes = sorted(m.items(), key=lambda kv: -len(kv[0]))
    inv = {}
    for old, new in prefixes:
        inv.setdefault(new, old)
    for key in sch:
        if "motion_modules." in key:
            continue
        for new, old in sorted(inv.items(), key=lambda kv: -len(kv[0])):
            if key == new or key.startswith(new + "."):
                out[old + key[len(new):]] = key
                break
    return out


def convert_ldm_unet_checkpoint(checkpoint: Dict[str, torch.Tensor], cfg: UNet3DConfig) -> Dict[str, torch.Tensor]:
    """DreamBooth / SD ``.ckpt`` / ``.safetensors`` state dict -> diffusers-named U-Net state dict (image layers)."""
    km = ldm_unet_key_map(cfg)
    out = {}
    for k, v in checkpoint.items():
        if not k.startswith(LDM_UNET_PREFIX):
            continue
        kk = k[len(LDM_UNET_PREFIX):]
        if kk in km:
            out[km[kk]] = v
    sch = state_dict_schema(cfg)
    for k in list(out):
        want = sch[k]
        if tuple(out[k].shape) != tuple(want):
            if out[k].numel() == int(torch.Size(want).numel()):
                out[k] = out[k].reshape(want)       # conv1x1 <-> linear (convert_from_ckpt.py:203-212 conv_attn_to_linear)
            else:
                raise ValueError(f"{k}: checkpoint shape {tuple(out[k].shape)} does not fit {tuple(want)}")
    return out


def filter_motion_module(state_dict: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """util.py:110-118: keep ``motion_modules.`` parameters, drop the (regenerated) ``pos_encoder.pe`` buffers."""
    sd = state_dict["state_dict"] if "state_dict" in state_dict else state_dict
    return {k: v for k, v in sd.items() if "motion_modules." in k and "pos_encoder.pe" not in k}


def diffusers_lora_deltas(lora_sd: Dict[str, torch.Tensor], alpha: float = 1.0) -> Dict[str, torch.Tensor]:
    """``load_diffusers_lora`` (:27-47): ``...attn1.processor.to_q_lora.down.weight`` -> delta for ``...attn1.to_q.weight``."""
    deltas = {}
    for key in lora_sd:
        if "up." in key:
            continue
        up_key = key.replace(".down.", ".up.")
        model_key = key.replace("processor.", "").replace("_lora", "").replace("down.", "").replace("up.", "")
        model_key = model_key.replace("to_out.", "to_out.0.")
        d = alpha * torch.mm(lora_sd[up_key].float(), lora_sd[key].float())
        deltas[model_key] = deltas[model_key] + d if model_key in deltas else d
    return deltas


def kohya_lora_deltas(lora_sd: Dict[str, torch.Tensor], cfg: UNet3DConfig, alpha: float = 0.6, prefix: str = "lora_unet") \
        -> Tuple[Dict[str, torch.Tensor], List[str]]:
    """``convert_lora`` (:50-112) for the U-Net keys: ``lora_unet_<path with underscores>.lora_down.weight``.  Returns
    (deltas by parameter name, the text-encoder keys that were skipped — those belong to the PyTorch CLIP module)."""
    sch = state_dict_schema(cfg)
    under = {k[:-len(".weight")].replace(".", "_"): k for k in sch if k.endswith(".weight")}
    deltas, skipped, seen = {}, [], set()
    for key in lora_sd:
        if ".alpha" in key or key in seen:
            continue
        if "text" in key:
            skipped.append(key)
            continue
        name = key.split(".")[0].split(prefix + "_")[-1]
        if name not in under:
            raise KeyError(f"LoRA layer {key} has no counterpart in the U-Net")
        up_key = key.replace("lora_down", "lora_up")
        down_key = key.replace("lora_up", "lora_down")
        up, down = lora_sd[up_key].float(), lora_sd[down_key].float()
        if up.dim() == 4:
            d = alpha * torch.mm(up.squeeze(3).squeeze(2), down.squeeze(3).squeeze(2)).unsqueeze(2).unsqueeze(3)
        else:
            d = alpha * torch.mm(up, down)
        target = under[name]
        deltas[target] = deltas[target] + d if target in deltas else d
        seen.update((up_key, down_key))
    return deltas, skipped


def apply_deltas(net, deltas: Dict[str, torch.Tensor]):
    """Fold LoRA deltas into a native network's not-yet-converted host weights."""
    for k, d in deltas.items():
        if k not in net._schema:
            raise KeyError(f"{k} is not a parameter of {type(net).__name__}")
        if k not in net._pending:
            raise RuntimeError(f"{k}: base weight is no longer on the host (LoRA must be merged before the first forward; "
                               "reload the state dict to re-merge)")
        net._pending[k] = net._pending[k].float() + d.reshape(net._pending[k].shape)


def te_lora_deltas(lora_sd: Dict[str, torch.Tensor], te_schema, alpha: float = 0.6, prefix: str = "lora_te") -> Dict[str, torch.Tensor]:
    """The text-encoder half of ``convert_lora`` (:50-112): ``lora_te_text_model_encoder_layers_0_self_attn_q_proj.lora_down.weight``
    -> delta for ``text_model.encoder.layers.0.self_attn.q_proj.weight``.  ``te_schema``: the encoder's parameter names."""
    under = {k[:-len(".weight")].replace(".", "_"): k for k in te_schema if k.endswith(".weight")}
    deltas, seen = {}, set()
    for key in lora_sd:
        if ".alpha" in key or key in seen or "text" not in key:
            continue
        name = key.split(".")[0].split(prefix + "_")[-1]
        if name not in under:
            raise KeyError(f"LoRA layer {key} has no counterpart in the text encoder")
        up_key, down_key = key.replace("lora_down", "lora_up"), key.replace("lora_up", "lora_down")
        d = alpha * torch.mm(lora_sd[up_key].float(), lora_sd[down_key].float())
        target = under[name]
        deltas[target] = deltas[target] + d if target in deltas else d
        seen.update((up_key, down_key))
    return deltas


def _is_native(obj) -> bool:
    return hasattr(obj, "_pending") and hasattr(obj, "_schema")


def load_weights(animation_pipeline, motion_module_path="", motion_module_lora_configs=(), adapter_lora_path="",
                 adapter_lora_scale=1.0, dreambooth_model_path="", lora_model_path="", lora_alpha=0.8,
                 vae_converter=None, text_encoder_converter=None):
    """Drop-in for ``animatediff.utils.util.load_weights`` (util.py:92-185) with a native ``pipeline.unet``.

    DreamBooth checkpoint (:125-144): the reference ALWAYS replaces VAE, U-Net and text encoder.  Here: a native VAE
    (``NativeAutoencoderKL``) takes its ``first_stage_model.*`` tensors directly (it uses the LDM key names), a native CLIP
    (``NativeCLIPTextModel``) its ``cond_stage_model.transformer.*`` tensors; for PyTorch modules pass the reference's
    ``convert_ldm_vae_checkpoint`` / ``convert_ldm_clip_checkpoint`` as ``vae_converter`` / ``text_encoder_converter``.  With
    neither, this raises instead of silently keeping the base VAE / text encoder (the videos would differ from the
    reference's).  Kohya LoRA (:147-160): the ``lora_te_*`` deltas are merged into the text encoder as ``convert_lora`` does."""
    unet = animation_pipeline.unet
    if motion_module_path != "":
        sd = filter_motion_module(torch.load(motion_module_path, map_location="cpu"))
        missing, unexpected = unet.load_state_dict(sd, strict=False)
        assert len(unexpected) == 0
    if dreambooth_model_path != "":
        if dreambooth_model_path.endswith(".safetensors"):
            from safetensors import safe_open
            ckpt = {}
            with safe_open(dreambooth_model_path, framework="pt", device="cpu") as f:
                for key in f.keys():
                    ckpt[key] = f.get_tensor(key)
        else:
            ckpt = torch.load(dreambooth_model_path, map_location="cpu")
        vae = getattr(animation_pipeline, "vae", None)
        te = getattr(animation_pipeline, "text_encoder", None)
        # 1. vae
        if vae_converter is not None:
            vae.load_state_dict(vae_converter(ckpt, vae.config))
        elif vae is not None and hasattr(vae, "load_ldm_state_dict"):
            vae.load_ldm_state_dict({k[len("first_stage_model."):]: v for k, v in ckpt.items() if k.startswith("first_stage_model.")})
        elif vae is not None:
            raise ValueError("load_weights(dreambooth_model_path=...): the reference also replaces the VAE (util.py:137-139); pass "
                             "vae_converter=convert_ldm_vae_checkpoint for a PyTorch VAE, or use NativeAutoencoderKL")
        # 2. unet
        unet.load_state_dict(convert_ldm_unet_checkpoint(ckpt, unet.config), strict=False)
        # 3. text_model
        if text_encoder_converter is not None:
            animation_pipeline.text_encoder = text_encoder_converter(ckpt)
        elif te is not None and _is_native(te):
            pre = "cond_stage_model.transformer."
            te.load_state_dict({k[len(pre):]: v for k, v in ckpt.items() if k.startswith(pre)})
        elif te is not None:
            raise ValueError("load_weights(dreambooth_model_path=...): the reference also replaces the text encoder (util.py:143-144); "
                             "pass text_encoder_converter=convert_ldm_clip_checkpoint for a PyTorch CLIP, or use NativeCLIPTextModel")
    if lora_model_path != "":
        assert lora_model_path.endswith(".safetensors")
        from safetensors.torch import load_file
        lsd = load_file(lora_model_path)
        deltas, te_keys = kohya_lora_deltas(lsd, unet.config, alpha=lora_alpha)
        apply_deltas(unet, deltas)
        if te_keys:
            te = getattr(animation_pipeline, "text_encoder", None)
            if te is not None and _is_native(te):
                apply_deltas(te, te_lora_deltas(lsd, te._schema, alpha=lora_alpha))
            elif isinstance(te, torch.nn.Module):
                params = dict(te.named_parameters())
                for k, d in te_lora_deltas(lsd, [n for n in params], alpha=lora_alpha).items():
                    params[k].data += d.to(params[k].device, params[k].dtype)
            else:
                raise ValueError(f"the LoRA carries {len(te_keys)} text-encoder tensors (lora_te_*) but the pipeline has no text encoder "
                                 "to merge them into (convert_lora merges them, :66-68)")
    if adapter_lora_path != "":
        sd = torch.load(adapter_lora_path, map_location="cpu")
        sd = sd["state_dict"] if "state_dict" in sd else sd
        sd.pop("animatediff_config", "")
        apply_deltas(unet, diffusers_lora_deltas(sd, adapter_lora_scale))
    for c in motion_module_lora_configs:
        sd = torch.load(c["path"], map_location="cpu")
        sd = sd["state_dict"] if "state_dict" in sd else sd
        sd.pop("animatediff_config", "")
        apply_deltas(unet, diffusers_lora_deltas(sd, c["alpha"]))
    return animation_pipeline
