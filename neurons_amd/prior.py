"""Sampling loop of the fMRI -> CLIP diffusion prior: ``BrainDiffusionPrior.p_sample_loop`` (model_variants/BrainModel_neurons.py:343-389,
called at recon_keyframe_neurons_enhance.py:364-366 with ``cond_scale=1., timesteps=100``).

BASELINE's north-star keeps the BrainModel / ``PriorNetwork`` forward in PyTorch-ROCm; what moves to HIP is everything of an iteration
BEHIND the network call: ``DiffusionPrior.p_mean_variance`` (x_start from the prediction, optional classifier-free guidance, clamp, the
posterior mean) and the ancestral update, fused into ONE kernel (C ABI ``nr_prior_p_sample_step``) instead of ~12 elementwise torch kernels
per step.  ``BrainDiffusionPrior`` subclasses ``dalle2_pytorch.DiffusionPrior`` (BrainModel_neurons.py:14-17,316); dalle2-pytorch 1.15.6
(requirements.txt:11) is NOT in /root/reference, so the noise schedule and the step are restated from its published algorithm
(``NoiseScheduler``, ``cosine_beta_schedule``, ``p_mean_variance``, ``q_posterior``) and the parity of this module is UNPINNED.

Drop-in use: ``NativePriorSampler.from_prior(model.diffusion_prior).p_sample_loop(shape, text_cond=..., cond_scale=1., timesteps=100)``.
"""
import math
from typing import Optional

import torch

from . import _lib


def cosine_beta_schedule(timesteps: int, s: float = 0.008):
    """dalle2_pytorch.cosine_beta_schedule (Nichol & Dhariwal 2021, eq. 17), fp64, betas clipped to [0, 0.999]."""
    steps = timesteps + 1
    x = torch.linspace(0, timesteps, steps, dtype=torch.float64)
    ac = torch.cos(((x / timesteps) + s) / (1 + s) * math.pi * 0.5) ** 2
    ac = ac / ac[0]
    betas = 1 - (ac[1:] / ac[:-1])
    return torch.clip(betas, 0, 0.999)


def linear_beta_schedule(timesteps: int):
    scale = 1000 / timesteps
    return torch.linspace(scale * 0.0001, scale * 0.02, timesteps, dtype=torch.float64)


class NoiseSchedule:
    """The buffers of ``dalle2_pytorch.NoiseScheduler`` the sampling loop reads (fp64 on the host; the HIP step forms its per-t
    coefficients from (alphas_cumprod[t], alphas_cumprod[t-1], betas[t]) and rounds them to fp32 as ``register_buffer`` does)."""

    def __init__(self, timesteps: int = 100, beta_schedule: str = "cosine"):
        if beta_schedule == "cosine":
            betas = cosine_beta_schedule(timesteps)
        elif beta_schedule == "linear":
            betas = linear_beta_schedule(timesteps)
        else:
            raise NotImplementedError(f"beta_schedule {beta_schedule!r} (the NEURONS prior uses dalle2's default 'cosine')")
        self.num_timesteps = int(timesteps)
        self.betas = betas
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.alphas_cumprod_prev = torch.cat([torch.ones(1, dtype=torch.float64), self.alphas_cumprod[:-1]])

    def triple(self, t: int):
        return float(self.alphas_cumprod[t]), float(self.alphas_cumprod_prev[t]), float(self.betas[t])


class NativePriorSampler:
    """``p_sample_loop`` / ``p_sample_loop_ddpm`` / ``p_sample`` of ``BrainDiffusionPrior`` around ANY network object with dalle2's
    ``forward_with_cond_scale(x, t, cond_scale=, self_cond=, **text_cond)`` (the caller's PyTorch ``PriorNetwork``,
    BrainModel_neurons.py:484-633) or, failing that, a plain ``net(x, t, **text_cond)``."""

    def __init__(self, net, image_embed_dim: int, timesteps: int = 100, beta_schedule: str = "cosine", predict_x_start: bool = True,
                 predict_v: bool = False, sampling_clamp_l2norm: bool = False, sampling_final_clamp_l2norm: bool = False,
                 init_image_embed_l2norm: bool = False, image_embed_scale: Optional[float] = None):
        if sampling_clamp_l2norm or sampling_final_clamp_l2norm or init_image_embed_l2norm:
            raise NotImplementedError("the l2norm clamps of dalle2's DiffusionPrior are off in the NEURONS configuration "
                                      "(recon_keyframe_neurons_enhance.py:232-239 keeps the defaults) and are not built")
        self.net = net
        self.image_embed_dim = image_embed_dim
        self.noise_scheduler = NoiseSchedule(timesteps, beta_schedule)
        self.predict_x_start, self.predict_v = predict_x_start, predict_v
        self.image_embed_scale = image_embed_scale if image_embed_scale is not None else image_embed_dim ** 0.5
        self.mode = 1 if predict_v else (0 if predict_x_start else 2)

    @classmethod
    def from_prior(cls, prior):
        """Build from a ``BrainDiffusionPrior`` / ``dalle2_pytorch.DiffusionPrior`` instance (reads its public attributes)."""
        ns = prior.noise_scheduler
        return cls(prior.net, prior.image_embed_dim, timesteps=int(ns.num_timesteps), predict_x_start=bool(prior.predict_x_start),
                   predict_v=bool(getattr(prior, "predict_v", False)), sampling_clamp_l2norm=bool(getattr(prior, "sampling_clamp_l2norm", False)),
                   sampling_final_clamp_l2norm=bool(getattr(prior, "sampling_final_clamp_l2norm", False)),
                   init_image_embed_l2norm=bool(getattr(prior, "init_image_embed_l2norm", False)), image_embed_scale=prior.image_embed_scale)

    def _network(self, x, times, text_cond, self_cond, cond_scale):
        """(conditional, null) predictions; null is None when cond_scale == 1 (forward_with_cond_scale returns the logits untouched then)."""
        kw = dict(text_cond or {})
        if self_cond is not None:
            kw["self_cond"] = self_cond
        if cond_scale == 1.0:
            if hasattr(self.net, "forward_with_cond_scale"):
                return self.net.forward_with_cond_scale(x, times, cond_scale=1.0, **kw), None
            return self.net(x, times, **kw), None
        # dalle2 forward_with_cond_scale: null_logits = forward(..., text_cond_drop_prob=1., image_cond_drop_prob=1); the combine runs in the kernel
        return self.net(x, times, **kw), self.net(x, times, text_cond_drop_prob=1.0, image_cond_drop_prob=1.0, **kw)

    @torch.no_grad()
    def p_sample(self, x, t, text_cond=None, self_cond=None, clip_denoised=True, cond_scale=1.0, generator=None, noise=None):
        """BrainModel_neurons.py:324-341 for a batch whose entries share one timestep (as the loops call it).  ``noise``: the N(0, 1)
        draw of this step (default: ``torch.randn_like(x)``, as the reference does -- it ignores ``generator`` too, :334-337)."""
        if not x.is_cuda:
            raise RuntimeError("NativePriorSampler.p_sample runs in the HIP kernel nr_prior_p_sample_step: CUDA (ROCm) tensors required")
        tt = int(t.flatten()[0]) if torch.is_tensor(t) else int(t)
        times = torch.full((x.shape[0],), tt, device=x.device, dtype=torch.long)
        pred, null = self._network(x, times, text_cond, self_cond, float(cond_scale))
        pred = pred.to(torch.float32).contiguous()
        null = None if null is None else null.to(torch.float32).contiguous()
        xin = x.to(torch.float32).contiguous()
        if tt > 0:
            noise = (torch.randn_like(xin) if noise is None else noise.to(torch.float32)).contiguous()
        else:
            noise = None                                              # "no noise when t == 0"
        out, x_start = torch.empty_like(xin), torch.empty_like(xin)
        ac, acp, beta = self.noise_scheduler.triple(tt)
        clamp = 1 if (clip_denoised and not self.predict_x_start) else 0
        _lib.check(_lib.load().nr_prior_p_sample_step(
            torch.cuda.current_stream().cuda_stream, pred.data_ptr(), None if null is None else null.data_ptr(), xin.data_ptr(),
            None if noise is None else noise.data_ptr(), out.data_ptr(), x_start.data_ptr(), xin.numel(), float(cond_scale), self.mode, clamp,
            ac, acp, beta))
        return out, x_start

    @torch.no_grad()
    def p_sample_loop_ddpm(self, shape, text_cond, cond_scale=1.0, generator=None, noises=None):
        """BrainModel_neurons.py:363-389.  ``noises``: optional explicit draws [x_T, eps_{T-1}, ..., eps_1] (tests)."""
        device = next(iter(text_cond.values())).device if text_cond else torch.device("cuda")
        if noises is not None:
            x = noises[0].to(device)
        elif generator is None:
            x = torch.randn(shape, device=device)
        else:
            x = torch.randn(shape, device=device, generator=generator)
        x_start = None
        T = self.noise_scheduler.num_timesteps
        for k, i in enumerate(reversed(range(T))):
            self_cond = x_start if getattr(self.net, "self_cond", False) else None
            x, x_start = self.p_sample(x, i, text_cond=text_cond, self_cond=self_cond, cond_scale=cond_scale, generator=generator,
                                       noise=None if noises is None or i == 0 else noises[1 + k])
        return x

    @torch.no_grad()
    def p_sample_loop(self, *args, timesteps=None, **kwargs):
        """BrainModel_neurons.py:343-361: DDPM when ``timesteps`` equals the schedule length (the NEURONS call: 100 of 100)."""
        timesteps = self.noise_scheduler.num_timesteps if timesteps is None else timesteps
        assert timesteps <= self.noise_scheduler.num_timesteps
        if timesteps < self.noise_scheduler.num_timesteps:
            raise NotImplementedError("p_sample_loop_ddim (fewer sampling steps than the schedule) lives in dalle2_pytorch, is not reached by "
                                      "the NEURONS scripts (timesteps=100 of 100) and is not built")
        return self.p_sample_loop_ddpm(*args, **kwargs)      # "PS removed all image_embed_scale instances!" (:359-360): no rescale
