"""Deterministic stand-ins for the three PyTorch modules NeuroclipsPipeline receives from its caller and that are NOT on the native
hot path of the a1 fixture: the CLIP tokenizer, the CLIP text encoder and the VAE decoder.  They are used twice with identical
behaviour: by oracle/gen_golden.py: gen_pipeline_call, where the REFERENCE's own ``NeuroclipsPipeline.__call__``
(animatediff/pipelines/pipeline_neuroclips.py:321-501) is run around them, and by the tests that feed the same objects to
``neurons_amd.NeuroclipsPipeline``.  Pure recipes (neurons_amd.synth), no reference code, no oracle code."""
import types

import torch


class FakeTokenizer:
    """Call surface the pipeline uses (pipeline_neuroclips.py:156-166,211-217): fixed-length ids, one code per character."""
    model_max_length = 77

    def __call__(self, text, padding=None, max_length=None, truncation=None, return_tensors=None):
        texts = [text] if isinstance(text, str) else list(text)
        L = self.model_max_length
        ids = torch.zeros(len(texts), L, dtype=torch.long)
        for i, t in enumerate(texts):
            codes = [ord(c) % 251 + 1 for c in t][:L]
            if codes:
                ids[i, :len(codes)] = torch.tensor(codes)
        return types.SimpleNamespace(input_ids=ids, attention_mask=(ids != 0).long())

    def batch_decode(self, ids):
        return ["".join(chr(int(c)) for c in row if int(c) > 0) for row in ids]


class FakeTextEncoder:
    """``text_encoder(ids, attention_mask=None)[0]`` -> (B, 77, dim): token-table lookup + position term (seeded recipe)."""

    def __init__(self, dim, seed=41):
        from neurons_amd.synth import randn
        self.table = randn("fake.clip.table", (256, dim), seed)
        self.pos = 0.5 * randn("fake.clip.pos", (FakeTokenizer.model_max_length, dim), seed + 1)
        self.config = types.SimpleNamespace()          # no use_attention_mask attribute (as CLIPTextConfig of SD-1.5)
        self.calls = 0

    def to(self, device):
        return self

    def __call__(self, ids, attention_mask=None):
        assert attention_mask is None
        self.calls += 1
        return (self.table.to(ids.device)[ids] + self.pos.to(ids.device),)


class FakeVAE:
    """``vae.decode(z).sample`` (pipeline_neuroclips.py:249): (1, 4, h, w) -> (1, 3, 8h, 8w), a fixed pointwise map + nearest 8x."""

    def __init__(self):
        self.config = types.SimpleNamespace(block_out_channels=(1, 1, 1, 1))      # vae_scale_factor = 2 ** 3 (:150)
        self.calls = 0

    def to(self, device):
        return self

    def decode(self, z):
        assert z.shape[0] == 1 and z.shape[1] == 4, "the reference decodes frame by frame"
        self.calls += 1
        # gentle on purpose: the pipeline hands over latents / 0.18215 (values of order 100 with random-init networks); a steep map would turn
        # the loop's bf16-level latent differences into sign flips of saturated pixels and measure nothing
        img = torch.tanh(z[:, :3] * 0.006 + 0.002 * z[:, 3:4])
        img = img.repeat_interleave(8, dim=-1).repeat_interleave(8, dim=-2)
        return types.SimpleNamespace(sample=img)
