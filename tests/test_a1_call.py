"""a1 pinned by the reference's OWN ``NeuroclipsPipeline.__call__`` (animatediff/pipelines/pipeline_neuroclips.py:321-501).

tests/golden/a1_call.npz was produced by oracle/gen_golden.py: gen_pipeline_call, which imports the reference pipeline class, gives it
the reference's tiny U-Net / SparseCtrl, the stand-in tokenizer / text encoder / VAE of tests/fake_modules.py and the oracle's DDIM
restatement as scheduler, calls ``pipe(prompt, ...)`` and records the ``noise`` it draws inside, every callback latent and ``.videos``.
Here the same call is made (CPU) through the oracle's restatement of the harness and (GPU) through neurons_amd.NeuroclipsPipeline with
the SAME stand-in objects, prompt string, latents and ``noise=``: harness facts the builder previously only checked against its own
reading (RNG order :395-418, cond/mask construction :447-458, the low_strength quirk :410-413, uncond-first CFG :238,479, per-frame decode
:242-255) are now checked against the reference function."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
GOLD = os.path.join(HERE, "golden", "a1_call.npz")

from fake_modules import FakeTextEncoder, FakeTokenizer, FakeVAE  # noqa: E402
from tiny_configs import tiny_ctrl_config, tiny_unet_config  # noqa: E402


def _ctx(g, dim):
    """[uncond | text] embeddings exactly as _encode_prompt builds them from the stand-ins (pipeline_neuroclips.py:153-240)."""
    tok, enc = FakeTokenizer(), FakeTextEncoder(dim)
    text = enc(tok(str(g["prompt"])).input_ids)[0]
    uncond = enc(tok("").input_ids)[0]
    return torch.cat([uncond, text])


def _decode_like_reference(latents):
    """decode_latents (:242-255) around the stand-in VAE."""
    vae = FakeVAE()
    b, c, f, h, w = latents.shape
    z = (1 / 0.18215 * latents).permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)
    video = torch.cat([vae.decode(z[i:i + 1]).sample for i in range(z.shape[0])])
    video = video.reshape(b, f, *video.shape[1:]).permute(0, 2, 1, 3, 4)
    return (video / 2 + 0.5).clamp(0, 1)


@torch.no_grad()
@pytest.mark.parametrize("case", ["A", "B"])
def test_oracle_harness_reproduces_the_reference_call(case):
    from neurons_amd import _lib
    from neurons_amd.unet3d import random_state_dict
    from oracle import animatediff_oracle as O
    g = np.load(GOLD)
    ucfg, ccfg = tiny_unet_config(), tiny_ctrl_config()
    usd = random_state_dict(ucfg, _lib.NR_KIND_UNET3D, seed=11)
    csd = random_state_dict(ccfg, _lib.NR_KIND_SPARSECTRL, seed=12)
    N = int(g["steps"])
    assert list(g[f"{case}.timesteps"]) == O.ddim_timesteps(N)
    x_log = {}
    final, _ = O.neuroclips_denoise(usd, O.OracleConfig.from_native(ucfg), csd, O.OracleConfig.from_native(ccfg),
                                    torch.from_numpy(g[f"{case}.latents"]), torch.from_numpy(g[f"{case}.noise"]),
                                    _ctx(g, ucfg.cross_attention_dim), torch.from_numpy(g[f"{case}.cimg"]),
                                    tuple(int(i) for i in g[f"{case}.index"]), N, float(g["guidance"]), x_log=x_log)
    want = torch.from_numpy(g[f"{case}.final_latents"])
    err = (final - want).abs().max().item()
    assert err <= 2e-3 * want.abs().max().item(), err
    err0 = (x_log["after"][0] - torch.from_numpy(g[f"{case}.latents_after_step0"])).abs().max().item()
    assert err0 <= 2e-4 * want.abs().max().item(), err0
    vid = _decode_like_reference(final)[:, :, :, ::8, ::8]
    assert (vid - torch.from_numpy(g[f"{case}.videos_sub"])).abs().max().item() <= 2e-3
    # facts recorded while the reference ran: low_strength 0.0 and 0.3 give the same result (all timesteps always run);
    # negative_prompt="" equals negative_prompt=None
    assert bool(g["quirk_low_strength_0_equals_0p3"]) and bool(g["negative_prompt_empty_str_equals_none"])


def test_encode_prompt_of_the_native_pipeline_matches_the_reference_order():
    """_encode_prompt is plain torch (CLIP stays the caller's module): runs on the CPU.  uncond first, then text (:238)."""
    from neurons_amd import DDIMScheduler, NeuroclipsPipeline
    g = np.load(GOLD)
    dim = tiny_unet_config().cross_attention_dim
    sched = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="linear", steps_offset=1, clip_sample=False)
    pipe = NeuroclipsPipeline(FakeVAE(), FakeTextEncoder(dim), FakeTokenizer(), None, sched, None)
    got = pipe._encode_prompt([str(g["prompt"])], torch.device("cpu"), 1, True, None)
    assert torch.equal(got, _ctx(g, dim))
    assert pipe.vae_scale_factor == 8


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["A", "B"])
def test_native_pipeline_call_matches_the_reference_call(cuda, case):
    from neurons_amd import _lib, DDIMScheduler, NativeSparseCtrl, NativeUNet3D, NeuroclipsPipeline
    from neurons_amd.unet3d import random_state_dict
    from test_engine_gpu import metrics
    g = np.load(GOLD)
    ucfg, ccfg = tiny_unet_config(), tiny_ctrl_config()
    unet, ctrl = NativeUNet3D(ucfg).to(cuda), NativeSparseCtrl(ccfg).to(cuda)
    unet.load_state_dict(random_state_dict(ucfg, _lib.NR_KIND_UNET3D, seed=11))
    ctrl.load_state_dict(random_state_dict(ccfg, _lib.NR_KIND_SPARSECTRL, seed=12))
    sched = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="linear", steps_offset=1, clip_sample=False)
    vae, enc = FakeVAE(), FakeTextEncoder(ucfg.cross_attention_dim)
    pipe = NeuroclipsPipeline(vae=vae, text_encoder=enc, tokenizer=FakeTokenizer(), unet=unet, scheduler=sched, controlnet=ctrl).to(cuda)
    traj = []
    F = int(g["frames"])
    res = pipe(str(g["prompt"]), video_length=F, height=64, width=64, num_inference_steps=int(g["steps"]),
               guidance_scale=float(g["guidance"]), latents=torch.from_numpy(g[f"{case}.latents"]).to(cuda),
               noise=torch.from_numpy(g[f"{case}.noise"]), controlnet_images=torch.from_numpy(g[f"{case}.cimg"]).to(cuda),
               controlnet_image_index=[int(i) for i in g[f"{case}.index"]], low_strength=float(g[f"{case}.low_strength"]),
               callback=lambda i, t, lat: traj.append((i, int(t), lat.clone())), callback_steps=1)
    assert [t for _, t, _ in traj] == list(g[f"{case}.timesteps"])
    assert enc.calls == 2 and vae.calls == F                      # prompt + uncond; one decode per frame (:248-249)
    rel, psnr = metrics(f"a1 case {case}: final latents of NeuroclipsPipeline.__call__ vs the reference's own __call__", traj[-1][2],
                        g[f"{case}.final_latents"])
    assert psnr >= 40.0, psnr
    videos = res.videos
    assert isinstance(videos, torch.Tensor) and tuple(videos.shape) == (1, 3, F, 64, 64) and videos.dtype == torch.float32
    sub = videos[:, :, :, ::8, ::8].float().cpu()
    mse = ((sub - torch.from_numpy(g[f"{case}.videos_sub"])) ** 2).mean().item()
    vpsnr = 10 * np.log10(1.0 / (mse + 1e-20))
    print(f"[a1 case {case}: .videos vs the reference's] psnr={vpsnr:.1f} dB")
    assert vpsnr >= 35.0, vpsnr
