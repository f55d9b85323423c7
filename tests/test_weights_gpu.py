"""GPU end of SURVEY 8(f2): checkpoint FILES -> neurons_amd.weights.load_weights -> the engine's converted bf16 arena -> one forward,
against the oracle on the state dict the reference's own converter / LoRA functions produce from the same inputs.

What the reference does (animatediff/utils/util.py:92-185): motion-module ckpt filtered into the U-Net (:106-121), DreamBooth / LDM
checkpoint converted with convert_ldm_unet_checkpoint (convert_from_ckpt.py:328) and loaded (:125-144), kohya LoRA merged
(convert_lora, convert_lora_safetensor_to_diffusers.py:50-112; :147-160), diffusers-style adapter LoRA merged (load_diffusers_lora, :27-47;
:163-171).  The expected merged tensors are pinned by tests/golden/weights.json (`lora_changed_checksums`: sums of the tensors the reference's
functions changed on the reference's own torch U-Net, recorded by oracle/gen_golden.py: gen_weights) — this test first checks its own
expectation against those checksums, then holds the HIP forward to the oracle on it.  The load is repeated AFTER a plan exists (other LoRA
strength): the engine must rebuild every converted buffer that depends on a reloaded tensor.

Tolerance: the fixture's LoRA factors are unit-variance rank-4 matrices (gen_weights), so at alpha 0.8 the five merged matrices grow ~10x in norm
(a to_q among them: sharply peaked softmax rows) — harsher than the N(0, 1/fan_in) regime the 2.5e-2 per-evaluation bar was stated for
(tools/stress_probe.py shows what that does to ANY bf16 path).  Stated here: rel-L2 <= 4e-2 and PSNR >= 40 dB (measured 3.0e-2 / 47 dB, round 6);
that the merges are IN the weights is shown by the same forward missing the un-merged / stale expectations by more than twice that."""
import json
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from test_engine_gpu import metrics  # noqa: E402
from test_weights import _lora_inputs  # noqa: E402

GOLD = json.load(open(os.path.join(HERE, "golden", "weights.json")))
FWD_REL, FWD_PSNR = 4e-2, 40.0


def _write_checkpoints(tmp_path, cfg, sd):
    """The four files load_weights reads, in the formats the reference reads them."""
    from safetensors.torch import save_file
    from neurons_amd import weights as W
    km = W.ldm_unet_key_map(cfg)                                     # == the reference converter's map (tests/test_weights.py, CPU)
    ldm = {W.LDM_UNET_PREFIX + old: sd[new].contiguous() for old, new in km.items()}
    k = W.LDM_UNET_PREFIX + "input_blocks.1.1.proj_in.weight"        # SD-1.5 checkpoints store proj_in / proj_out as 1x1 convs
    assert ldm[k].dim() == 4
    ldm["first_stage_model.decoder.conv_in.bias"] = torch.zeros(4)   # other sub-models of the checkpoint are not the U-Net's business
    db = os.path.join(tmp_path, "dreambooth.safetensors")
    save_file(ldm, db)
    mm = {"state_dict": {k: v for k, v in sd.items() if "motion_modules." in k}}
    pe_key = [k for k in mm["state_dict"] if k.endswith("attention_blocks.0.to_q.weight")][0].replace("to_q.weight", "pos_encoder.pe")
    mm["state_dict"][pe_key] = torch.zeros(1, 32, 64)                # regenerated buffers the reference filters out (util.py:117)
    mmp = os.path.join(tmp_path, "mm.ckpt")
    torch.save(mm, mmp)
    kohya, dl = _lora_inputs(cfg)
    kohya = {k: v.contiguous() for k, v in kohya.items() if "lora_te_" not in k}   # no text encoder in this pipeline object
    kp = os.path.join(tmp_path, "kohya.safetensors")
    save_file(kohya, kp)
    ap = os.path.join(tmp_path, "adapter.ckpt")
    torch.save({"state_dict": dict(dl)}, ap)
    return db, mmp, kp, ap, kohya, dl


def _expected(sd, cfg, kohya, dl, lora_alpha, adapter_scale):
    """W + alpha * up @ down per target, written out independently of neurons_amd.weights."""
    from neurons_amd.unet3d import state_dict_schema
    out = {k: v.clone().float() for k, v in sd.items()}
    for t in GOLD["kohya_targets"]:
        name = "lora_unet_" + t.replace(".", "_")
        up, down = kohya[name + ".lora_up.weight"].float(), kohya[name + ".lora_down.weight"].float()
        d = lora_alpha * (up.flatten(1) @ down.flatten(1))
        out[t + ".weight"] += d.reshape(out[t + ".weight"].shape)
    for key in dl:
        if ".up." in key:
            continue
        base = key.replace(".down.weight", "").replace("processor.", "").replace("_lora", "").replace("to_out", "to_out.0") + ".weight"
        out[base] += adapter_scale * (dl[key.replace(".down.", ".up.")].float() @ dl[key].float())
    assert set(out) == set(state_dict_schema(cfg))
    return out


def test_checkpoint_files_through_load_weights_into_the_engine(cuda, tmp_path):
    import types
    from neurons_amd import _lib, NativeUNet3D
    from neurons_amd import weights as W
    from neurons_amd.synth import randn
    from neurons_amd.unet3d import random_state_dict
    from oracle import animatediff_oracle as O
    from tiny_configs import tiny_unet_config
    cfg = tiny_unet_config()
    sd = random_state_dict(cfg, _lib.NR_KIND_UNET3D, seed=11)        # the base weights gen_weights merged on
    db, mmp, kp, ap, kohya, dl = _write_checkpoints(str(tmp_path), cfg, sd)

    # the expectation is the reference's: same checksums as the tensors its convert_lora / load_diffusers_lora left on its own U-Net
    want = _expected(sd, cfg, kohya, dl, 0.8, 0.7)
    changed = {k for k in sd if not torch.equal(want[k], sd[k].float())}
    assert changed == set(GOLD["lora_changed_checksums"])
    for k, (s, a) in GOLD["lora_changed_checksums"].items():
        assert abs(float(want[k].double().sum()) - s) <= 1e-4 * max(1.0, abs(a)) and abs(float(want[k].double().abs().sum()) - a) <= 1e-5 * a, k

    unet = NativeUNet3D(cfg).to("cuda")                               # EMPTY: every tensor arrives through the files
    pipe = types.SimpleNamespace(unet=unet, vae=None, text_encoder=None)
    W.load_weights(pipe, motion_module_path=mmp, dreambooth_model_path=db, lora_model_path=kp, lora_alpha=0.8,
                   adapter_lora_path=ap, adapter_lora_scale=0.7)
    assert not [k for k in unet._schema if k not in unet._loaded]    # nothing missing, the pos_encoder.pe entry was dropped

    dev = torch.device("cuda")
    sample = randn("f2.sample", (2, 4, 8, 8, 8), 5).to(dev)
    ctx = randn("f2.ctx", (2, 77, cfg.cross_attention_dim), 6).to(dev)
    ocfg = O.OracleConfig.from_native(cfg)

    def oracle_eps(state):
        with torch.no_grad():
            return O.unet3d_forward({k: v.to(dev) for k, v in state.items()}, ocfg, sample, 481, ctx)

    eps = unet(sample, 481, encoder_hidden_states=ctx).sample
    rel, psnr = metrics("f2: files -> load_weights -> engine vs oracle(reference-merged weights)", eps, oracle_eps(want))
    assert rel < FWD_REL and psnr > FWD_PSNR
    # the merges matter at this tolerance's scale: the un-merged base weights give a visibly different answer
    rel_base, _ = metrics("f2: same forward vs oracle(base weights, no LoRA)", eps, oracle_eps({k: v.float() for k, v in sd.items()}))
    assert rel_base > 2 * rel

    # ---- reload AFTER a plan exists, other strengths: every converted buffer fed by a reloaded tensor is rebuilt ----
    W.load_weights(pipe, motion_module_path=mmp, dreambooth_model_path=db, lora_model_path=kp, lora_alpha=0.3,
                   adapter_lora_path=ap, adapter_lora_scale=1.5)
    eps2 = unet(sample, 481, encoder_hidden_states=ctx).sample
    want2 = _expected(sd, cfg, kohya, dl, 0.3, 1.5)
    rel2, psnr2 = metrics("f2: reload after plan (alpha 0.3 / 1.5) vs oracle", eps2, oracle_eps(want2))
    assert rel2 < FWD_REL and psnr2 > FWD_PSNR
    assert not torch.equal(eps, eps2)
    rel_stale, _ = metrics("f2: second forward vs FIRST expectation (must be worse)", eps2, oracle_eps(want))
    assert rel_stale > 2 * rel2

    # ---- and back: the first configuration again reproduces the first result bit for bit (conversion is deterministic) ----
    W.load_weights(pipe, motion_module_path=mmp, dreambooth_model_path=db, lora_model_path=kp, lora_alpha=0.8,
                   adapter_lora_path=ap, adapter_lora_scale=0.7)
    eps3 = unet(sample, 481, encoder_hidden_states=ctx).sample
    assert torch.equal(eps, eps3)
