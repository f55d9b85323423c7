"""CPU: pin oracle/clip_oracle.py against vectors produced by the installed transformers CLIPTextModel (the class
_encode_prompt calls; fixture records the transformers version) and check the host-side surface."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
GOLD = os.path.join(HERE, "golden", "clip_tiny.npz")

from neurons_amd.clip import CLIPTextConfig, NativeCLIPTextModel, clip_random_state_dict, clip_state_dict_schema  # noqa: E402
from oracle import clip_oracle as CO  # noqa: E402
from tiny_configs import tiny_clip_config  # noqa: E402


@torch.no_grad()
def test_oracle_matches_transformers_golden():
    g = np.load(GOLD)
    cfg = tiny_clip_config()
    sd = clip_random_state_dict(cfg, seed=97)
    out = CO.clip_text_forward(sd, torch.from_numpy(g["ids"]), cfg.num_hidden_layers, cfg.num_attention_heads)
    want = torch.from_numpy(g["last_hidden_state"])
    assert (out - want).abs().max().item() <= 2e-4 * want.abs().max().item()
    # causal: changing a later token must not change earlier positions
    ids2 = torch.from_numpy(g["ids"]).clone()
    ids2[:, 40:] = 7
    out2 = CO.clip_text_forward(sd, ids2, cfg.num_hidden_layers, cfg.num_attention_heads)
    assert torch.allclose(out[:, :40], out2[:, :40], atol=1e-5) and not torch.allclose(out[:, 40:], out2[:, 40:], atol=1e-3)


def test_schema_is_the_sd15_text_encoder():
    sch = clip_state_dict_schema(CLIPTextConfig())
    assert len(sch) == 196 and sum(int(np.prod(s)) for s in sch.values()) == 123_060_480     # ViT-L/14 text tower
    assert sch["text_model.encoder.layers.11.mlp.fc1.weight"] == (3072, 768)


def test_product_path_refuses_cpu_and_masks():
    enc = NativeCLIPTextModel(tiny_clip_config())
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        enc(torch.zeros(1, 77, dtype=torch.long))
    with pytest.raises(NotImplementedError):
        enc(torch.zeros(1, 77, dtype=torch.long), attention_mask=torch.ones(1, 77))
    with pytest.raises(NotImplementedError):
        NativeCLIPTextModel(CLIPTextConfig(hidden_act="gelu"))
    assert enc.config.use_attention_mask is False        # what _encode_prompt reads (:173)
    missing, unexpected = enc.load_state_dict({"text_model.embeddings.position_ids": torch.arange(77)[None]}, strict=False)
    assert unexpected == [] and len(missing) == len(clip_state_dict_schema(tiny_clip_config()))
