"""DDIM scheduler with the surface the reference pipeline uses.

The reference imports ``DDIMScheduler`` from un-vendored ``diffusers==0.11.1`` (README.md:58;
constructed at scripts/neuroclips_video.py:219 from configs/inference/inference-v3.yaml:16-21; used at
animatediff/pipelines/pipeline_neuroclips.py:317,378-379,423,431,436,483).  That source is not under
/root/reference, so this is a restatement of the published DDIM algorithm (Song et al. 2020, eq. 12, eta
as in diffusers' ``step``) — parity for this class is *unpinned* by any reference test; it is cross-checked
against the in-repo sibling ``animatediff/utils/util.py:211-221`` (``next_step``) in tests/.

Only table construction / coefficient selection happens here (host logic).  The per-element update runs in
the HIP kernel ``nr_cfg_ddim_step`` (include/neurons_amd.h); ``step`` refuses CPU tensors.
"""
from dataclasses import dataclass
from types import SimpleNamespace

import numpy as np
import torch


@dataclass
class DDIMSchedulerOutput:
    prev_sample: torch.Tensor
    pred_original_sample: torch.Tensor = None


class DDIMScheduler:
    order = 1

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 trained_betas=None, clip_sample=True, set_alpha_to_one=True, steps_offset=0,
                 prediction_type="epsilon"):
        if trained_betas is not None:
            betas = torch.as_tensor(trained_betas, dtype=torch.float32)
        elif beta_schedule == "linear":
            # diffusers "linear": linspace in beta (NOT SD's scaled_linear) — SURVEY.md F12
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        elif beta_schedule == "scaled_linear":
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        else:
            raise NotImplementedError(f"{beta_schedule} does is not implemented for {self.__class__}")
        if prediction_type != "epsilon":
            raise NotImplementedError("only epsilon prediction is on the NEURONS path")
        self.betas = betas
        self.alphas = 1.0 - betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                                      beta_schedule=beta_schedule, clip_sample=clip_sample,
                                      set_alpha_to_one=set_alpha_to_one, steps_offset=steps_offset,
                                      prediction_type=prediction_type)
        self._timesteps_host = [int(t) for t in self.timesteps]

    def scale_model_input(self, sample, timestep=None):
        return sample

    def set_timesteps(self, num_inference_steps, device=None):
        self.num_inference_steps = num_inference_steps
        step_ratio = self.config.num_train_timesteps // num_inference_steps
        timesteps = (np.arange(0, num_inference_steps) * step_ratio).round()[::-1].copy().astype(np.int64)
        timesteps = timesteps + self.config.steps_offset
        self._timesteps_host = [int(t) for t in timesteps]          # no device sync inside the loop
        self.timesteps = torch.from_numpy(timesteps).to(device)

    @property
    def timesteps_host(self):
        return list(self._timesteps_host)

    def alpha_pair(self, timestep):
        """(alpha_prod_t, alpha_prod_t_prev) as python floats for integer ``timestep``."""
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        t = int(timestep)
        prev_t = t - self.config.num_train_timesteps // self.num_inference_steps
        a_t = float(self.alphas_cumprod[t])
        a_prev = float(self.alphas_cumprod[prev_t]) if prev_t >= 0 else float(self.final_alpha_cumprod)
        return a_t, a_prev

    def add_noise(self, original_samples, noise, timesteps):
        ac = self.alphas_cumprod.to(device=original_samples.device, dtype=original_samples.dtype)
        timesteps = timesteps.to(original_samples.device)
        sqrt_alpha_prod = ac[timesteps] ** 0.5
        sqrt_alpha_prod = sqrt_alpha_prod.flatten()
        while len(sqrt_alpha_prod.shape) < len(original_samples.shape):
            sqrt_alpha_prod = sqrt_alpha_prod.unsqueeze(-1)
        sqrt_one_minus = (1 - ac[timesteps]) ** 0.5
        sqrt_one_minus = sqrt_one_minus.flatten()
        while len(sqrt_one_minus.shape) < len(original_samples.shape):
            sqrt_one_minus = sqrt_one_minus.unsqueeze(-1)
        return sqrt_alpha_prod * original_samples + sqrt_one_minus * noise

    def step(self, model_output, timestep, sample, eta=0.0, use_clipped_model_output=False, generator=None,
             variance_noise=None, return_dict=True):
        if eta != 0.0:
            raise NotImplementedError("the NEURONS path runs DDIM with eta = 0 (pipeline_neuroclips.py:331)")
        if self.config.clip_sample:
            raise NotImplementedError("clip_sample=True is not on the NEURONS path (inference-v3.yaml:21)")
        if not (model_output.is_cuda and sample.is_cuda):
            raise RuntimeError("DDIMScheduler.step runs in the HIP kernel nr_cfg_ddim_step: CUDA (ROCm) tensors required, "
                               "there is no CPU fallback")
        from . import ops
        a_t, a_prev = self.alpha_pair(timestep)
        prev = ops.cfg_ddim_step(model_output.float().contiguous(), sample.float().contiguous(), 1.0, a_t, a_prev,
                                 do_cfg=False).to(sample.dtype)
        if not return_dict:
            return (prev,)
        return DDIMSchedulerOutput(prev_sample=prev)
