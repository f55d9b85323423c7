"""GPU: the first-stage decoder through the C ABI (nr_vae_decode) against reference-generated vectors (tiny width) and
against the pinned oracle at the SD-VAE width.  The decoder is ~30 bf16 convolutions deep with fp32 accumulation;
the bar is PSNR >= 40 dB on the decoded image (north_star tolerance for the pixel-space outputs)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
GOLD = os.path.join(HERE, "golden", "vae_tiny.npz")

from test_engine_gpu import metrics  # noqa: E402


def _tiny():
    from neurons_amd.vae import NativeVAEDecoder, vae_random_state_dict
    from tiny_configs import tiny_vae_config
    cfg = tiny_vae_config()
    sd = vae_random_state_dict(cfg, seed=91)
    dec = NativeVAEDecoder(cfg).to("cuda")
    dec.load_state_dict(sd)
    return dec, cfg, sd


def test_tiny_decode_first_stage_matches_reference_golden(cuda):
    g = np.load(GOLD)
    dec, _, _ = _tiny()
    z = torch.from_numpy(g["z"]).cuda()
    img = dec.decode_first_stage(z)
    rel, psnr = metrics("tiny VAE decode_first_stage vs reference", img, g["image"])
    assert psnr >= 40.0 and rel < 2.5e-2
    assert torch.equal(img, dec.decode_first_stage(z))                        # deterministic (graph replay)
    # images are independent; a different batch size re-chunks the GroupNorm partial sums (fp32 summation order), so the
    # comparison is "far inside the parity tolerance", not bitwise
    _, p1 = metrics("image 1 alone vs in a batch of 2", dec.decode_first_stage(z[1:]), img[1:])
    assert p1 >= 60.0


def test_tiny_decode_latents_matches_reference_golden(cuda):
    g = np.load(GOLD)
    dec, _, _ = _tiny()
    vid = dec.decode_latents(torch.from_numpy(g["lat"]).cuda())
    assert vid.shape == g["video"].shape and float(vid.min()) >= 0.0 and float(vid.max()) <= 1.0
    rel, psnr = metrics("tiny VAE decode_latents vs reference", vid, g["video"])
    assert psnr >= 40.0


def test_tiny_vae_taps_vs_oracle(cuda):
    from neurons_amd import _lib
    from oracle import vae_oracle as V
    g = np.load(GOLD)
    dec, cfg, sd = _tiny()
    lib = _lib.load()
    _lib.check(lib.nr_net_set_debug(dec._handle(), 1))
    z = torch.from_numpy(g["z"]).cuda()
    dec.decode(z, z_scale=1 / 0.18215)
    taps = {}
    with torch.no_grad():
        V.decode({k: v.cuda() for k, v in sd.items()}, z / 0.18215, len(cfg.ch_mult), cfg.num_res_blocks, taps=taps)
    n = lib.nr_net_num_taps(dec._h)
    assert n == len(taps)
    worst = 0.0
    for i in range(n):
        name = lib.nr_net_tap_name(dec._h, i).decode()
        ref = taps[name]                                   # b c h w
        b, c, h, w = ref.shape
        buf = np.empty(b * h * w * c, dtype=np.float32)
        rows, cc = C.c_int32(), C.c_int32()
        _lib.check(lib.nr_net_read_tap(dec._h, i, buf.ctypes.data_as(C.c_void_p), buf.size, C.byref(rows), C.byref(cc)))
        got = torch.from_numpy(buf).reshape(b, h, w, c).permute(0, 3, 1, 2)
        rel, _ = metrics(f"tap {name}", got, ref)
        worst = max(worst, rel)
    assert worst < 3e-2


def _gpu_sd(cfg, seed):
    from neurons_amd.vae import vae_decoder_state_dict_schema
    g = torch.Generator(device="cuda").manual_seed(seed)
    sd = {}
    for k, shape in vae_decoder_state_dict_schema(cfg).items():
        z = torch.randn(shape, generator=g, device="cuda")
        if k.endswith(".bias"):
            z = (0.1 if ".norm" in k else 0.02) * z
        elif len(shape) == 1:
            z = 1.0 + 0.1 * z
        else:
            z = z / (int(np.prod(shape[1:])) ** 0.5)
        sd[k] = z
    return sd


def test_sd_width_decoder_vs_oracle(cuda):
    """unclip6.yaml / SD-1.5 VAE width (ch 128, mult 1,2,4,4; mid attention with d = 512) at a 16x24 latent."""
    from neurons_amd.synth import randn
    from neurons_amd.vae import NativeVAEDecoder, VAEDecoderConfig
    from oracle import vae_oracle as V
    cfg = VAEDecoderConfig()
    sd = _gpu_sd(cfg, 7)
    dec = NativeVAEDecoder(cfg).to("cuda")
    dec.load_state_dict({k: v.cpu() for k, v in sd.items()})
    z = randn("v.z", (3, 4, 16, 24), 1).cuda()
    img = dec.decode(z, z_scale=1 / 0.18215)
    with torch.no_grad():
        ref = V.decode(sd, z / 0.18215, 4, 2)
    rel, psnr = metrics("SD-width VAE decoder vs oracle", img, ref)
    assert psnr >= 40.0 and rel < 2.5e-2
    # chunked decode: same images; not bit-identical because the igemm picks tile / split-K per M (fp32 summation
    # order differs), so the bar is "far inside the parity tolerance"
    _, p2 = metrics("chunked vs batched decode", dec.decode(z, z_scale=1 / 0.18215, chunk=2), img)
    assert p2 >= 50.0


def test_full_size_clip_decode_properties(cuda):
    """BASELINE config 2 output size: 16 frames of 32x32 latents -> 256x256.  Size-independent properties: frames are
    independent (a 16-frame launch agrees with 4-frame launches far inside tolerance), the unit-range epilogue equals the host
    expression on the raw output, and one frame agrees with the oracle."""
    from neurons_amd.synth import randn
    from neurons_amd.vae import NativeVAEDecoder, VAEDecoderConfig
    from oracle import vae_oracle as V
    cfg = VAEDecoderConfig()
    sd = _gpu_sd(cfg, 9)
    dec = NativeVAEDecoder(cfg).to("cuda")
    dec.load_state_dict({k: v.cpu() for k, v in sd.items()})
    lat = (randn("v.lat", (1, 4, 16, 32, 32), 2) * 0.18215).cuda()
    vid = dec.decode_latents(lat)
    assert vid.shape == (1, 3, 16, 256, 256)
    frames = lat.permute(0, 2, 1, 3, 4).reshape(16, 4, 32, 32)
    raw = dec.decode(frames, z_scale=1 / 0.18215)
    assert torch.equal(vid[0].permute(1, 0, 2, 3), (raw / 2 + 0.5).clamp(0, 1))
    _, p2 = metrics("16-frame vs 4x4-frame decode", dec.decode(frames, z_scale=1 / 0.18215, chunk=4), raw)
    assert p2 >= 50.0
    key = dec.decode_keyframe(frames[:2])                                     # utils.py:343-348 post-scaling
    raw2 = dec.decode(frames[:2], z_scale=1 / 0.18215)                         # same 2-image plan as `key`
    assert torch.allclose(key, (raw2 * 0.8 + 0.2).clamp(0, 1), rtol=0, atol=1e-6)        # fma vs mul+add rounding
    with torch.no_grad():
        ref = V.decode(sd, frames[5:6] / 0.18215, 4, 2)
    rel, psnr = metrics("256x256 frame vs oracle", raw[5:6], ref)
    assert psnr >= 40.0


def test_vae_input_errors(cuda):
    dec, _, _ = _tiny()
    with pytest.raises(ValueError):
        dec.decode(torch.zeros(1, 3, 8, 8, device="cuda"))
    with pytest.raises(RuntimeError, match="multiple of 64"):
        dec.decode(torch.zeros(1, 4, 6, 6, device="cuda"))


def _tiny_enc():
    from neurons_amd.vae import NativeVAEEncoder, vae_random_state_dict
    from tiny_configs import tiny_vae_config
    cfg = tiny_vae_config()
    esd = vae_random_state_dict(cfg, seed=94, encoder=True)
    enc = NativeVAEEncoder(cfg).to("cuda")
    enc.load_state_dict(esd)
    return enc, cfg, esd


def test_tiny_encoder_matches_reference_golden(cuda):
    g = np.load(GOLD)
    enc, _, _ = _tiny_enc()
    img = torch.from_numpy(g["enc_img"]).cuda()
    post = enc.encode(img, in_mul=2.0, in_add=-1.0)                 # vae.encode(2 * x - 1)  scripts/neuroclips_video.py:267
    rel, psnr = metrics("tiny VAE encoder moments vs reference", post.parameters, g["enc_moments"])
    assert psnr >= 40.0 and rel < 2.5e-2
    assert torch.equal(post.parameters, enc.encode(2 * img - 1).parameters) or \
        metrics("fused 2x-1 vs host 2x-1", post.parameters, enc.encode(2 * img - 1).parameters)[1] > 60
    # .sample() * 0.18215 with the reference's noise; .mode()
    lat = post._draw(torch.from_numpy(g["enc_noise"]).cuda(), 0.18215)
    rel, psnr = metrics("latent_dist.sample() * 0.18215 vs reference", lat, g["enc_sample"])
    assert psnr >= 40.0
    rel, psnr = metrics("latent_dist.mode() * 0.18215 vs reference", post.mode(scale=0.18215), g["enc_mode"])
    assert psnr >= 40.0
    # the sampling kernel itself is exact arithmetic on its inputs
    from oracle import vae_oracle as V
    want = V.gaussian_sample(post.parameters, torch.from_numpy(g["enc_noise"]).cuda()) * 0.18215
    assert torch.allclose(lat, want, rtol=1e-5, atol=1e-6)
    # generator-driven draw is reproducible
    g1 = torch.Generator(device="cuda").manual_seed(3)
    g2 = torch.Generator(device="cuda").manual_seed(3)
    assert torch.equal(post.sample(generator=g1), post.sample(generator=g2))


def test_tiny_encoder_taps_vs_oracle(cuda):
    from neurons_amd import _lib
    from oracle import vae_oracle as V
    g = np.load(GOLD)
    enc, cfg, esd = _tiny_enc()
    lib = _lib.load()
    _lib.check(lib.nr_net_set_debug(enc._handle(), 1))
    img = torch.from_numpy(g["enc_img"]).cuda()
    enc.moments(img, 2.0, -1.0)
    taps = {}
    with torch.no_grad():
        V.encode_moments({k: v.cuda() for k, v in esd.items()}, 2 * img - 1, len(cfg.ch_mult), cfg.num_res_blocks, taps=taps)
    n = lib.nr_net_num_taps(enc._h)
    assert n == len(taps)
    worst = 0.0
    for i in range(n):
        name = lib.nr_net_tap_name(enc._h, i).decode()
        ref = taps[name]
        b, c, h, w = ref.shape
        buf = np.empty(b * h * w * c, dtype=np.float32)
        rows, cc = C.c_int32(), C.c_int32()
        _lib.check(lib.nr_net_read_tap(enc._h, i, buf.ctypes.data_as(C.c_void_p), buf.size, C.byref(rows), C.byref(cc)))
        got = torch.from_numpy(buf).reshape(b, h, w, c).permute(0, 3, 1, 2)
        rel, _ = metrics(f"tap {name}", got, ref)
        worst = max(worst, rel)
    assert worst < 3e-2


def test_sd_width_autoencoder_surface_vs_oracle(cuda):
    """The scripts' ``vae`` object (diffusers surface) at SD-VAE width on a 17-image 256x256 batch (16 blurry frames + the
    control image, scripts/neuroclips_video.py:263-283), loaded from DIFFUSERS-named weights."""
    from neurons_amd.vae import NativeAutoencoderKL, VAEDecoderConfig, diffusers_vae_key_map, vae_encoder_state_dict_schema
    from oracle import vae_oracle as V
    cfg = VAEDecoderConfig()
    sd = _gpu_sd(cfg, 11)
    gen = torch.Generator(device="cuda").manual_seed(12)
    for k, shape in vae_encoder_state_dict_schema(cfg).items():
        z = torch.randn(shape, generator=gen, device="cuda")
        sd[k] = (0.1 if ".norm" in k else 0.02) * z if k.endswith(".bias") else (1.0 + 0.1 * z if len(shape) == 1 else z / (int(np.prod(shape[1:])) ** 0.5))
    inv = {v: k for k, v in {**diffusers_vae_key_map(cfg), **diffusers_vae_key_map(cfg, True)}.items()}
    dsd = {}
    for k, v in sd.items():
        dk = inv[k]
        dsd[dk] = (v[:, :, 0, 0] if "attentions.0" in dk and v.dim() == 4 else v).cpu()
    vae = NativeAutoencoderKL(cfg).to("cuda")
    missing, unexpected = vae.load_state_dict(dsd)
    assert not missing and not unexpected
    x = torch.rand(17, 3, 256, 256, generator=gen, device="cuda")
    post = vae.encode(2 * x - 1).latent_dist
    assert post.parameters.shape == (17, 8, 32, 32)
    with torch.no_grad():
        ref = V.encode_moments(sd, 2 * x[3:5] - 1, 4, 2)
    rel, psnr = metrics("SD-width encoder moments vs oracle", post.parameters[3:5], ref)
    assert psnr >= 40.0 and rel < 2.5e-2
    lat = post.sample(generator=gen) * 0.18215
    img = vae.decode(lat[:2] / 0.18215).sample
    assert img.shape == (2, 3, 256, 256) and torch.isfinite(img).all()
