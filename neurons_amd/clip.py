"""CLIP text encoder — host mirror of the ``text_encoder`` object ``_encode_prompt`` calls
(animatediff/pipelines/pipeline_neuroclips.py:153-240: ``self.text_encoder(text_input_ids, attention_mask=None)[0]``),
i.e. transformers ``CLIPTextModel`` (SD-1.5 ``text_encoder/config.json``: ViT-L/14 text tower) up to
``last_hidden_state``.  Every FLOP runs in libneurons_amd.so (kind NR_KIND_CLIP_TEXT); tokenisation stays with the
caller's ``CLIPTokenizer``.  There is no CPU fallback.
"""
import ctypes as C
from dataclasses import dataclass
from typing import Dict

import numpy as np
import torch

from . import _lib
from .unet3d import _on_device, _NativeNet


@dataclass
class CLIPTextConfig:
    """SD-1.5 text_encoder/config.json (openai/clip-vit-large-patch14 text tower)."""
    vocab_size: int = 49408
    hidden_size: int = 768
    intermediate_size: int = 3072
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    max_position_embeddings: int = 77
    hidden_act: str = "quick_gelu"
    layer_norm_eps: float = 1e-5
    use_attention_mask: bool = False     # read by _encode_prompt (:173-176); SD-1.5 does not set it
    in_channels: int = 0                 # (module-like surface of _NativeNet)


def clip_c_config(cfg: CLIPTextConfig) -> _lib.NrNetConfig:
    if cfg.hidden_act != "quick_gelu":
        raise NotImplementedError("only quick_gelu (SD-1.5 text encoder) is built")
    if cfg.use_attention_mask:
        raise NotImplementedError("use_attention_mask text encoders are not built (SD-1.5 passes attention_mask=None)")
    c = _lib.NrNetConfig()
    c.kind = _lib.NR_KIND_CLIP_TEXT
    c.num_levels = 2
    c.block_out_channels[0] = c.block_out_channels[1] = cfg.hidden_size
    c.num_heads = cfg.num_attention_heads
    c.layers_per_block = cfg.num_hidden_layers
    c.cross_attention_dim = cfg.intermediate_size
    c.in_channels = cfg.vocab_size
    c.motion_pe_max_len = cfg.max_position_embeddings
    c.norm_num_groups = 1
    c.norm_eps = cfg.layer_norm_eps
    return c


def clip_state_dict_schema(cfg: CLIPTextConfig) -> Dict[str, tuple]:
    """Parameter names / shapes of ``CLIPTextModel.state_dict()`` (the ``position_ids`` buffer is ignored)."""
    h, i = cfg.hidden_size, cfg.intermediate_size
    k = {"text_model.embeddings.token_embedding.weight": (cfg.vocab_size, h),
         "text_model.embeddings.position_embedding.weight": (cfg.max_position_embeddings, h)}
    for l in range(cfg.num_hidden_layers):
        p = f"text_model.encoder.layers.{l}"
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            k[f"{p}.self_attn.{n}.weight"] = (h, h)
            k[f"{p}.self_attn.{n}.bias"] = (h,)
        for n in ("layer_norm1", "layer_norm2"):
            k[f"{p}.{n}.weight"] = (h,)
            k[f"{p}.{n}.bias"] = (h,)
        k[f"{p}.mlp.fc1.weight"] = (i, h)
        k[f"{p}.mlp.fc1.bias"] = (i,)
        k[f"{p}.mlp.fc2.weight"] = (h, i)
        k[f"{p}.mlp.fc2.bias"] = (h,)
    k["text_model.final_layer_norm.weight"] = (h,)
    k["text_model.final_layer_norm.bias"] = (h,)
    return k


def clip_random_state_dict(cfg: CLIPTextConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    from .synth import randn
    sd = {}
    for name, shape in clip_state_dict_schema(cfg).items():
        z = randn(name, shape, seed)
        if "embedding" in name:
            t = 0.5 * z
        elif name.endswith(".bias"):
            t = (0.1 if "norm" in name else 0.02) * z
        elif len(shape) == 1:
            t = 1.0 + 0.1 * z
        else:
            t = z / (shape[1] ** 0.5)
        sd[name] = t
    return sd


class CLIPTextOutput(tuple):
    """``BaseModelOutputWithPooling``-like: ``out[0]`` and ``out.last_hidden_state`` (pooled output is not computed —
    the pipeline never reads it)."""

    def __new__(cls, last_hidden_state):
        o = super().__new__(cls, (last_hidden_state,))
        o.last_hidden_state = last_hidden_state
        return o


class NativeCLIPTextModel(_NativeNet):
    _kind = _lib.NR_KIND_CLIP_TEXT
    _config_cls = CLIPTextConfig

    def _build_cconf(self, config):
        return clip_c_config(config)

    def _build_schema(self, config):
        return clip_state_dict_schema(config)

    def load_state_dict(self, state_dict, strict=True):
        sd = {k: v for k, v in state_dict.items() if not k.endswith("position_ids")}
        return super().load_state_dict(sd, strict=strict)

    def _on_plan(self):
        b, f, h, w, L = self._plan_key
        dev = self.device
        self._io_ids = torch.empty(b, w, dtype=torch.int32, device=dev)

    @_on_device
    def forward(self, input_ids, attention_mask=None, **kwargs):
        if attention_mask is not None or kwargs:
            raise NotImplementedError("attention_mask / extra arguments are not built (SD-1.5: attention_mask=None)")
        if not input_ids.is_cuda:
            raise RuntimeError("NativeCLIPTextModel.forward: CUDA (ROCm) tensors required; there is no CPU fallback")
        if input_ids.dim() != 2:
            raise ValueError("input_ids must be [batch][seq_len]")
        b, L = input_ids.shape
        if L > self.config.max_position_embeddings:
            raise ValueError(f"sequence length {L} exceeds max_position_embeddings {self.config.max_position_embeddings}")
        outs = []
        for i in range(0, b, 16):
            ids = input_ids[i:i + 16]
            self._ensure_plan(ids.shape[0], 1, 1, L, 0)
            self._io_ids.copy_(ids)
            out = torch.empty(ids.shape[0], L, self.config.hidden_size, dtype=torch.float32, device=input_ids.device)
            _lib.check(_lib.load().nr_clip_text_forward(self._handle(), torch.cuda.current_stream().cuda_stream,
                                                        self._io_ids.data_ptr(), out.data_ptr()))
            outs.append(out)
        return CLIPTextOutput(outs[0] if len(outs) == 1 else torch.cat(outs))

    __call__ = forward
