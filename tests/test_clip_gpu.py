"""GPU: the CLIP text encoder through the C ABI (nr_clip_text_forward) against transformers-generated vectors (tiny
width) and the pinned oracle at SD-1.5 width; then inside NeuroclipsPipeline._encode_prompt with a stand-in tokenizer."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
GOLD = os.path.join(HERE, "golden", "clip_tiny.npz")

from test_engine_gpu import metrics  # noqa: E402


def _tiny():
    from neurons_amd.clip import NativeCLIPTextModel, clip_random_state_dict
    from tiny_configs import tiny_clip_config
    cfg = tiny_clip_config()
    sd = clip_random_state_dict(cfg, seed=97)
    enc = NativeCLIPTextModel(cfg).to("cuda")
    enc.load_state_dict(sd)
    return enc, cfg, sd


def test_tiny_clip_matches_transformers_golden(cuda):
    g = np.load(GOLD)
    enc, _, _ = _tiny()
    ids = torch.from_numpy(g["ids"]).cuda()
    out = enc(ids, attention_mask=None)[0]
    rel, psnr = metrics("tiny CLIP text encoder vs transformers", out, g["last_hidden_state"])
    assert rel < 2.5e-2 and psnr > 35
    assert torch.equal(out, enc(ids.long()).last_hidden_state)
    # causal mask: later tokens do not influence earlier positions (bit-exact: same rows, same arithmetic)
    ids2 = ids.clone()
    ids2[:, 40:] = 7
    out2 = enc(ids2)[0]
    assert torch.equal(out[:, :40], out2[:, :40]) and not torch.equal(out[:, 40:], out2[:, 40:])
    # batch rows are independent
    assert torch.equal(out[1:], enc(ids[1:])[0])


def test_tiny_clip_taps_vs_oracle(cuda):
    from neurons_amd import _lib
    from oracle import clip_oracle as CO
    g = np.load(GOLD)
    enc, cfg, sd = _tiny()
    lib = _lib.load()
    _lib.check(lib.nr_net_set_debug(enc._handle(), 1))
    ids = torch.from_numpy(g["ids"]).cuda()
    enc(ids)
    taps = {}
    with torch.no_grad():
        CO.clip_text_forward({k: v.cuda() for k, v in sd.items()}, ids, cfg.num_hidden_layers, cfg.num_attention_heads, taps=taps)
    n = lib.nr_net_num_taps(enc._h)
    assert n == len(taps)
    for i in range(n):
        name = lib.nr_net_tap_name(enc._h, i).decode()
        ref = taps[name]
        b, L, c = ref.shape
        buf = np.empty(b * L * c, dtype=np.float32)
        rows, cc = C.c_int32(), C.c_int32()
        _lib.check(lib.nr_net_read_tap(enc._h, i, buf.ctypes.data_as(C.c_void_p), buf.size, C.byref(rows), C.byref(cc)))
        rel, _ = metrics(f"tap {name}", torch.from_numpy(buf).reshape(b, L, c), ref)
        assert rel < 3e-2


def test_sd15_width_clip_vs_oracle_and_encode_prompt(cuda):
    from neurons_amd import DDIMScheduler, NeuroclipsPipeline
    from neurons_amd.clip import CLIPTextConfig, NativeCLIPTextModel, clip_state_dict_schema
    from oracle import clip_oracle as CO
    cfg = CLIPTextConfig()
    gen = torch.Generator(device="cuda").manual_seed(21)
    sd = {}
    for k, shape in clip_state_dict_schema(cfg).items():
        z = torch.randn(shape, generator=gen, device="cuda")
        if "embedding" in k:
            z = 0.5 * z
        elif k.endswith(".bias"):
            z = (0.1 if "norm" in k else 0.02) * z
        elif len(shape) == 1:
            z = 1.0 + 0.1 * z
        else:
            z = z / (shape[1] ** 0.5)
        sd[k] = z
    enc = NativeCLIPTextModel(cfg).to("cuda")
    enc.load_state_dict({k: v.cpu() for k, v in sd.items()})
    ids = torch.randint(0, cfg.vocab_size, (2, 77), generator=gen, device="cuda")
    out = enc(ids)[0]
    with torch.no_grad():
        ref = CO.clip_text_forward(sd, ids, 12, 12)
    rel, psnr = metrics("SD-1.5-width CLIP text encoder vs oracle", out, ref)
    assert rel < 2.5e-2 and psnr > 35

    # _encode_prompt (pipeline_neuroclips.py:153-240) with the native text encoder and a stand-in tokenizer
    class Tok:
        model_max_length = 77

        def __call__(self, prompt, padding=None, max_length=None, truncation=None, return_tensors=None):
            prompt = [prompt] if isinstance(prompt, str) else prompt
            rows = []
            for p in prompt:
                t = [49406] + [1 + (ord(ch) % 1000) for ch in p][:75] + [49407]
                rows.append(t + [49407] * (77 - len(t)))
            return type("Enc", (), {"input_ids": torch.tensor(rows), "attention_mask": torch.ones(len(rows), 77)})()

        def batch_decode(self, x):
            return [""]

    pipe = NeuroclipsPipeline(vae=None, text_encoder=enc, tokenizer=Tok(), unet=None,
                              scheduler=DDIMScheduler(beta_start=0.00085, beta_end=0.012, clip_sample=False, steps_offset=1))
    emb = pipe._encode_prompt("a cat on a mat", torch.device("cuda"), 1, True, None)
    assert emb.shape == (2, 77, 768)
    tok = Tok()
    with torch.no_grad():
        want = torch.cat([CO.clip_text_forward(sd, tok("").input_ids.cuda(), 12, 12),
                          CO.clip_text_forward(sd, tok("a cat on a mat").input_ids.cuda(), 12, 12)])       # uncond first (:238)
    rel, psnr = metrics("_encode_prompt with the native text encoder vs oracle", emb, want)
    assert rel < 2.5e-2
