"""CPU: pin oracle/sgm_oracle.py and the sgm host logic against vectors produced by the reference's own sgm classes."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
GOLD = os.path.join(HERE, "golden", "sgm_tiny.npz")

from neurons_amd.sgm import DiscreteDenoiser, LegacyDDPMDiscretization, SGMUNetConfig, sgm_random_state_dict, sgm_state_dict_schema  # noqa: E402
from oracle import sgm_oracle as S  # noqa: E402
from tiny_configs import tiny_sgm_config  # noqa: E402


def _close(name, got, want, tol=2e-4):
    got, want = got.detach().float(), torch.as_tensor(want).float()
    assert got.shape == want.shape
    err, scale = (got - want).abs().max().item(), want.abs().max().item()
    assert err <= tol * scale + 1e-6, f"{name}: {err:.3e} vs scale {scale:.3e}"


@torch.no_grad()
def test_unet_forward_matches_reference():
    g = np.load(GOLD)
    cfg = tiny_sgm_config()
    sd = sgm_random_state_dict(cfg, seed=71)
    eps = S.unet_forward(sd, cfg, torch.from_numpy(g["x"]), torch.from_numpy(g["t"]), torch.from_numpy(g["ctx"]), torch.from_numpy(g["y"]))
    _close("sgm eps", eps, g["eps"])


def test_sigma_tables_match_reference():
    g = np.load(GOLD)
    for n in (38, 50):
        assert np.allclose(S.legacy_ddpm_sigmas(n).numpy(), g[f"sigmas{n}"], rtol=1e-6, atol=0)
        assert np.allclose(LegacyDDPMDiscretization()(n).numpy(), g[f"sigmas{n}"], rtol=1e-6, atol=0)
    s = LegacyDDPMDiscretization()(38)
    assert len(s) == 39 and abs(float(s[0]) - 14.61) < 0.01 and abs(float(s[-2]) - 0.158) < 1e-3 and float(s[-1]) == 0.0
    den = DiscreteDenoiser()
    sq, c_in, idx = den.scalars(float(s[0]))
    assert sq == float(s[0]) and idx == 999 and abs(c_in - 1 / (sq * sq + 1) ** 0.5) < 1e-7


@torch.no_grad()
def test_euler_cfg_loop_matches_reference():
    g = np.load(GOLD)
    cfg = tiny_sgm_config()
    sd = sgm_random_state_dict(cfg, seed=71)
    ctx, y = torch.from_numpy(g["ctx"]), torch.from_numpy(g["y"])
    c = {"crossattn": ctx[1:2], "vector": y[1:2]}
    uc = {"crossattn": ctx[0:1], "vector": y[1:2]}
    final = S.euler_edm_sample(sd, cfg, torch.from_numpy(g["z"]), c, uc, 4, 5.0)
    _close("sgm 4-step loop", final, g["loop_final"], tol=2e-3)


def test_full_size_schema_matches_survey():
    sc = sgm_state_dict_schema(SGMUNetConfig())
    assert abs(sum(int(np.prod(s)) for s in sc.values()) / 1e6 - 2501.3) < 0.1     # 2 501 M parameters


@torch.no_grad()
def test_unclip_recon_harness_matches_reference():
    """a18: the fixture is the output of the reference's OWN utils.unclip_recon (gen_golden.gen_unclip); the oracle's
    restatement of the harness (noised_z, offset noise, uc tokens, uc-first CFG, clamp(x*.8+.2)) must reproduce it."""
    from neurons_amd.vae import vae_random_state_dict
    from oracle import vae_oracle as V
    from tiny_configs import tiny_vae_config
    g = np.load(os.path.join(HERE, "golden", "unclip_tiny.npz"))
    cfg, vcfg = tiny_sgm_config(), tiny_vae_config()
    sd = sgm_random_state_dict(cfg, seed=71)
    vsd = vae_random_state_dict(vcfg, seed=91)
    t = {k: torch.from_numpy(g[k]) for k in ("tokens", "vector_suffix", "z", "uc_tokens", "noise", "offset")}
    out = S.unclip_recon(sd, cfg, t["tokens"], t["vector_suffix"], t["z"], t["uc_tokens"], t["noise"], t["offset"], int(g["num_steps"]),
                         lambda z: V.decode_first_stage(vsd, z, len(vcfg.ch_mult), vcfg.num_res_blocks))
    assert tuple(out.shape) == (1, 3, 768, 768)
    st = int(g["stride"])
    want = torch.from_numpy(g["samples_sub"].astype(np.float32))
    err = (out[:, :, ::st, ::st] - want).abs().max().item()
    assert err <= 2e-3, err                                      # fp16 storage of values in [0, 1]: 5e-4, + fp32 reassociation
    assert abs(out.double().mean().item() - float(g["samples_mean"])) < 1e-4
    assert abs((out.double() ** 2).mean().item() - float(g["samples_sq"])) < 1e-4


# ---- the DiffusionEngine-shaped boundary (neurons_amd.sgm: DiscreteDenoiser / VanillaCFG / EulerEDMSampler generic path) ----
class _OracleModel:
    """OpenAIWrapper call shape (wrappers.py:23-34) around the fp32 oracle network: the FOREIGN model of the CPU boundary tests."""

    def __init__(self, sd, cfg):
        self.sd, self.cfg = sd, cfg

    def __call__(self, x, t, c, **kw):
        return S.unet_forward(self.sd, self.cfg, x, t, c["crossattn"], c["vector"])


@torch.no_grad()
def test_boundary_generic_sampler_with_foreign_denoiser_matches_reference_loop():
    """sampler(denoiser_fn, x, cond=, uc=) with a closure the sampler does not recognise: the reference's loop statement for statement
    (prepare_inputs -> denoiser -> guider -> to_d -> euler_step), DiscreteDenoiser.forward on per-sample sigma tensors."""
    from neurons_amd.sgm import EulerEDMSampler
    g = np.load(GOLD)
    cfg = tiny_sgm_config()
    model = _OracleModel(sgm_random_state_dict(cfg, seed=71), cfg)
    den = DiscreteDenoiser()
    ctx, y = torch.from_numpy(g["ctx"]), torch.from_numpy(g["y"])
    c = {"crossattn": ctx[1:2], "vector": y[1:2]}
    uc = {"crossattn": ctx[0:1], "vector": y[1:2]}
    calls = []

    def denoiser(x, sigma, cc):
        calls.append((tuple(x.shape), tuple(sigma.shape), sorted(cc)))
        return den(model, x, sigma, cc)

    sampler = EulerEDMSampler(num_steps=4, scale=5.0)
    z = torch.from_numpy(g["z"])
    z0 = z.clone()
    final = sampler(denoiser, z, cond=c, uc=uc)
    _close("boundary: generic 4-step loop", final, g["loop_final"], tol=2e-3)
    assert len(calls) == 4 and all(cl == ((2,) + tuple(z.shape[1:]), (2,), ["crossattn", "vector"]) for cl in calls), calls   # CFG-doubled
    assert torch.equal(z, z0)                       # the caller's tensor is not modified
    # tensor API of the denoiser == its host-scalar view
    s = sampler.discretization(4)
    for i in range(4):
        sq, c_in, idx = den.scalars(float(s[i]))
        st = s[i].reshape(1)
        assert int(den.sigma_to_idx(st)) == idx and float(den.possibly_quantize_sigma(st)) == sq
        assert int(den.possibly_quantize_c_noise(den.possibly_quantize_sigma(st))) == idx


@torch.no_grad()
def test_boundary_unclip_recon_call_sequence_on_cpu_engine():
    """The call sequence of utils.unclip_recon (oracle/unclip_harness.py) against an engine-shaped object whose network is the oracle:
    pins ema_scope / sampler.discretization / sampler.num_steps / denoiser(model, x, sigma, c) / sampler(closure, …) / decode_first_stage
    of the boundary classes against the reference's own output (tests/golden/unclip_tiny.npz)."""
    import contextlib
    import types
    from neurons_amd.sgm import EulerEDMSampler
    from neurons_amd.vae import vae_random_state_dict
    from oracle import vae_oracle as V
    from tiny_configs import tiny_vae_config
    from oracle.unclip_harness import call_like_unclip_recon
    g = np.load(os.path.join(HERE, "golden", "unclip_tiny.npz"))
    cfg, vcfg = tiny_sgm_config(), tiny_vae_config()
    vsd = vae_random_state_dict(vcfg, seed=91)
    engine = types.SimpleNamespace(
        ema_scope=contextlib.nullcontext, model=_OracleModel(sgm_random_state_dict(cfg, seed=71), cfg), denoiser=DiscreteDenoiser(),
        sampler=EulerEDMSampler(num_steps=int(g["num_steps"]), scale=5.0),
        decode_first_stage=lambda z: V.decode_first_stage(vsd, z, len(vcfg.ch_mult), vcfg.num_res_blocks))
    t = {k: torch.from_numpy(g[k]) for k in ("tokens", "vector_suffix", "z", "uc_tokens", "noise", "offset")}
    out = call_like_unclip_recon(t["tokens"], engine, t["vector_suffix"], t, num_samples=1, offset_noise_level=0.04, device="cpu")
    st = int(g["stride"])
    want = torch.from_numpy(g["samples_sub"].astype(np.float32))
    err = (out[:, :, ::st, ::st] - want).abs().max().item()
    assert err <= 2e-3, err
    assert abs(out.double().mean().item() - float(g["samples_mean"])) < 1e-4
