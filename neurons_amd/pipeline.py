"""NeuroclipsPipeline — the denoising loop of ``animatediff/pipelines/pipeline_neuroclips.py`` (:43, ``__call__``
:321-501) with the two networks and the CFG+DDIM update running in libneurons_amd.so.

Kept identical to the reference (so ``scripts/neuroclips_video*.py`` can call it unchanged):
  * call signature and defaults (:322-346), ``check_inputs`` errors (:274-287), ``prepare_latents`` (:289-318)
  * CFG order: uncond first (``cat([uncond, text])`` :238, ``chunk(2)`` :479)
  * RNG order: an unused ``keylatents`` randn draw happens before ``noise = randn_like(latents)`` (:395-405,418)
  * the ``low_strength`` quirk: latents are noised to ``timesteps[0]`` and ALL timesteps are run;
    ``low_strength >= 1`` gives an empty ``latent_timestep`` exactly as in the reference (:410-413)
  * SparseCtrl cond/mask construction (:447-458) — built once, they are step-invariant
Additive extensions (defaults keep reference behaviour): ``text_embeddings=`` (skip the CLIP encoder),
``noise=`` (explicit noise instead of the in-call draw), ``output_type="latent"`` (skip the VAE).
``vae`` may be a ``neurons_amd.vae.NativeVAEDecoder`` (the decode then also runs in HIP) or the
caller's PyTorch ``AutoencoderKL``; CLIP stays a PyTorch module supplied by the caller.
"""
import os
from dataclasses import dataclass
from typing import Callable, List, Optional, Union

import numpy as np
import torch

from . import _lib


@dataclass
class AnimationPipelineOutput:
    videos: Union[torch.Tensor, np.ndarray]


class _NullBar:
    def __init__(self, total=None):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def update(self, n=1):
        pass


def controlnet_group_plan(timesteps, group):
    """Grouped SparseCtrl schedule as plain data (host logic, unit-tested on CPU).  Returns ``(groups, steps)``:
    ``groups[g]`` = the timesteps evaluated by group g (``group`` of them; the tail of the last group repeats the final timestep: those
    samples are evaluated and never consumed) and its event/buffer slot ``g % 2``;
    ``steps[i]`` = ``(launch, g, phase)``: before step i's U-Net forward issue the evaluation of group ``launch`` (``None``: nothing to
    issue), then consume samples ``phase`` of group ``g``.  Group 0 is issued before the loop; group g + 1 is issued at the first step
    of group g, i.e. one group ahead, into the slot whose last reader was step ``g * group - 1``."""
    n = len(timesteps)
    group = max(1, min(int(group), n))
    ngroups = (n + group - 1) // group
    groups = [dict(timesteps=[timesteps[min(g * group + p, n - 1)] for p in range(group)], slot=g % 2) for g in range(ngroups)]
    steps = []
    for i in range(n):
        g, p = divmod(i, group)
        launch = g + 1 if (p == 0 and g + 1 < ngroups) else None
        steps.append((launch, g, p))
    return groups, steps


def controlnet_group_size(num_steps, batch2, frames, lat_h, lat_w, preferred="auto"):
    """DDIM steps per SparseCtrl evaluation under the grouped schedule (host logic, unit-tested on CPU).
    Bounds: at most 64 samples per evaluation (engine limit NR_MAX_BATCH) and 2 Mi level-0 rows (the largest evaluation the full-size tests
    exercise: 8 clips x CFG x 4 steps at 16 f x 32 x 32, BASELINE config 5's 4 clips x CFG x 2 steps at 32 f x 64 x 64).
    An integer ``preferred`` is taken as is (capped).  ``"auto"``: the size in 1..5 with the least total cost
    ``c(G) * ceil(N / G) * G`` -- the tail of the last group is evaluated and never consumed (50 steps at G = 4: 2 of 52 evaluations), and
    c(G) is the measured per-step cost of one evaluation at group size G relative to G >= 4 (5.64 / 4.68 / 4.15 ms at G = 1 / 2 / 4 on
    MI355X, G = 8 no better: tools/ctrl_batch.py); 50, 25 and 10 steps give G = 5 (no tail)."""
    cap = max(1, min(64 // max(1, batch2), (2 << 20) // max(1, batch2 * frames * lat_h * lat_w), num_steps))
    if preferred != "auto":
        return max(1, min(int(preferred), cap))
    rel = {1: 1.36, 2: 1.13, 3: 1.06}
    best, best_cost = 1, None
    for g in range(1, min(cap, 5) + 1):
        cost = rel.get(g, 1.0) * (-(-num_steps // g)) * g
        if best_cost is None or cost < best_cost - 1e-9 or (abs(cost - best_cost) <= 1e-9 and abs(g - 4) < abs(best - 4)):
            best, best_cost = g, cost
    return best


class NeuroclipsPipeline:
    _optional_components = []

    def __init__(self, vae, text_encoder, tokenizer, unet, scheduler, controlnet=None):
        # the reference patches steps_offset != 1 / clip_sample == True configs with a deprecation warning
        # (pipeline_neuroclips.py:64-89); the NEURONS scheduler config already has the patched values
        if getattr(scheduler.config, "steps_offset", 1) != 1:
            scheduler.config.steps_offset = 1
        if getattr(scheduler.config, "clip_sample", False) is True:
            scheduler.config.clip_sample = False
        self.register_modules(vae=vae, text_encoder=text_encoder, tokenizer=tokenizer, unet=unet, scheduler=scheduler,
                              controlnet=controlnet)
        if vae is not None and hasattr(vae, "config") and hasattr(vae.config, "block_out_channels"):
            self.vae_scale_factor = 2 ** (len(vae.config.block_out_channels) - 1)
        else:
            self.vae_scale_factor = 8
        self._progress_bar_config = {}
        self._device = torch.device("cpu")
        self.overlap_controlnet = True   # run SparseCtrl concurrently with the U-Net encoder (nr_denoise_step_forward)
        # issue step i+1's SparseCtrl evaluation during step i's decoder (results identical).  Measured +0.2 % (round 1: 14.00 vs
        # 13.97 frames/s) and +0.5 % (round 2: 16.15 / 16.17 vs 16.08) on MI355X: the CUs are already saturated by the encoder
        # overlap, so off by default (NR_PREFETCH=1 turns it on)
        self.prefetch_controlnet = os.environ.get("NR_PREFETCH") == "1"
        # SparseCtrl evaluations of `controlnet_group` consecutive DDIM steps run as ONE forward on a `group` x larger batch, one group
        # ahead of the U-Net steps that consume them (the network sees timestep, context and condition only, not the latents:
        # set_noisy_sample_input_to_zero).  Same 50 evaluations, better GEMM shapes; 1 = one evaluation per step (nr_denoise_step_forward)
        # "auto" (default): controlnet_group_size picks the size with no wasted tail (5 for 50 / 25 / 10 steps); an integer forces it
        self.controlnet_group = int(os.environ["NR_CTRL_GROUP"]) if os.environ.get("NR_CTRL_GROUP") else "auto"

    # ---- DiffusionPipeline surface used by the scripts (SURVEY §8c "Python harness rows") ----
    def register_modules(self, **kwargs):
        for k, v in kwargs.items():
            setattr(self, k, v)

    def to(self, device):
        self._device = torch.device(device)
        for name in ("vae", "text_encoder", "unet", "controlnet"):
            m = getattr(self, name, None)
            if m is not None and hasattr(m, "to"):
                m.to(self._device)
        return self

    @property
    def device(self):
        return self._device

    @property
    def _execution_device(self):
        return self._device

    def set_progress_bar_config(self, **kwargs):
        self._progress_bar_config = kwargs

    def progress_bar(self, total=None):
        return _NullBar(total)

    def enable_vae_slicing(self):
        if hasattr(self.vae, "enable_slicing"):
            self.vae.enable_slicing()

    def disable_vae_slicing(self):
        if hasattr(self.vae, "disable_slicing"):
            self.vae.disable_slicing()

    # ---- prompt encoding (pipeline_neuroclips.py:153-240); CLIP stays PyTorch ----
    def _encode_prompt(self, prompt, device, num_videos_per_prompt, do_classifier_free_guidance, negative_prompt):
        if self.tokenizer is None or self.text_encoder is None:
            raise ValueError("tokenizer/text_encoder are required to encode prompts; pass text_embeddings= instead")
        batch_size = len(prompt) if isinstance(prompt, list) else 1
        text_inputs = self.tokenizer(prompt, padding="max_length", max_length=self.tokenizer.model_max_length,
                                     truncation=True, return_tensors="pt")
        text_input_ids = text_inputs.input_ids
        use_mask = hasattr(self.text_encoder.config, "use_attention_mask") and self.text_encoder.config.use_attention_mask
        attention_mask = text_inputs.attention_mask.to(device) if use_mask else None
        text_embeddings = self.text_encoder(text_input_ids.to(device), attention_mask=attention_mask)[0]
        bs_embed, seq_len, _ = text_embeddings.shape
        text_embeddings = text_embeddings.repeat(1, num_videos_per_prompt, 1).view(bs_embed * num_videos_per_prompt, seq_len, -1)
        if do_classifier_free_guidance:
            if negative_prompt is None:
                uncond_tokens = [""] * batch_size
            elif type(prompt) is not type(negative_prompt):
                raise TypeError(f"`negative_prompt` should be the same type to `prompt`, but got {type(negative_prompt)} !="
                                f" {type(prompt)}.")
            elif isinstance(negative_prompt, str):
                uncond_tokens = [negative_prompt]
            elif batch_size != len(negative_prompt):
                raise ValueError(f"`negative_prompt`: {negative_prompt} has batch size {len(negative_prompt)}, but `prompt`:"
                                 f" {prompt} has batch size {batch_size}. Please make sure that passed `negative_prompt` matches"
                                 " the batch size of `prompt`.")
            else:
                uncond_tokens = negative_prompt
            max_length = text_input_ids.shape[-1]
            uncond_input = self.tokenizer(uncond_tokens, padding="max_length", max_length=max_length, truncation=True,
                                          return_tensors="pt")
            attention_mask = uncond_input.attention_mask.to(device) if use_mask else None
            uncond_embeddings = self.text_encoder(uncond_input.input_ids.to(device), attention_mask=attention_mask)[0]
            seq_len = uncond_embeddings.shape[1]
            uncond_embeddings = uncond_embeddings.repeat(1, num_videos_per_prompt, 1).view(batch_size * num_videos_per_prompt, seq_len, -1)
            text_embeddings = torch.cat([uncond_embeddings, text_embeddings])
        return text_embeddings

    # ---- VAE decode (pipeline_neuroclips.py:242-255).  A NativeVAEDecoder as ``vae`` decodes all frames in one engine
    # launch with the /2+0.5 clamp fused; any other ``vae`` (the caller's PyTorch AutoencoderKL) goes frame by frame
    # exactly as the reference does ----
    def decode_latents(self, latents):
        from .vae import NativeVAEDecoder
        if isinstance(self.vae, NativeVAEDecoder):
            return self.vae.decode_latents(latents).cpu().float().numpy()
        video_length = latents.shape[2]
        latents = 1 / 0.18215 * latents
        b, c, f, h, w = latents.shape
        latents = latents.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)
        video = []
        for frame_idx in range(latents.shape[0]):
            video.append(self.vae.decode(latents[frame_idx:frame_idx + 1]).sample)
        video = torch.cat(video)
        video = video.reshape(b, video_length, *video.shape[1:]).permute(0, 2, 1, 3, 4)
        video = (video / 2 + 0.5).clamp(0, 1)
        return video.cpu().float().numpy()

    def prepare_extra_step_kwargs(self, generator, eta):
        """Same contract as pipeline_neuroclips.py:257-272: `eta` / `generator` reach scheduler.step only when its signature names them."""
        import inspect
        names = set(inspect.signature(self.scheduler.step).parameters.keys())
        extra = {}
        if "eta" in names:
            extra["eta"] = eta
        if "generator" in names:
            extra["generator"] = generator
        return extra

    def check_inputs(self, prompt, height, width, callback_steps):
        if not isinstance(prompt, str) and not isinstance(prompt, list):
            raise ValueError(f"`prompt` has to be of type `str` or `list` but is {type(prompt)}")
        if height % 8 != 0 or width % 8 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")
        if (callback_steps is None) or (callback_steps is not None and (not isinstance(callback_steps, int) or callback_steps <= 0)):
            raise ValueError(f"`callback_steps` has to be a positive integer but is {callback_steps} of type"
                             f" {type(callback_steps)}.")

    def prepare_latents(self, batch_size, num_channels_latents, video_length, height, width, dtype, device, generator,
                        latents=None):
        shape = (batch_size, num_channels_latents, video_length, height // self.vae_scale_factor, width // self.vae_scale_factor)
        if isinstance(generator, list) and len(generator) != batch_size:
            raise ValueError(f"You have passed a list of generators of length {len(generator)}, but requested an effective batch"
                             f" size of {batch_size}. Make sure the batch size matches the length of the generators.")
        if latents is None:
            if isinstance(generator, list):
                latents = [torch.randn(shape, generator=generator[i], device=device, dtype=dtype) for i in range(batch_size)]
                latents = torch.cat(latents, dim=0).to(device)
            else:
                latents = torch.randn(shape, generator=generator, device=device, dtype=dtype).to(device)
        else:
            if latents.shape != shape:
                raise ValueError(f"Unexpected latents shape, got {latents.shape}, expected {shape}")
            latents = latents.to(device)
        latents = latents * self.scheduler.init_noise_sigma
        return latents

    @torch.no_grad()
    def __call__(self, prompt: Union[str, List[str]], video_length: Optional[int], height: Optional[int] = None,
                 width: Optional[int] = None, num_inference_steps: int = 50, guidance_scale: float = 7.5,
                 negative_prompt: Optional[Union[str, List[str]]] = None, num_videos_per_prompt: Optional[int] = 1,
                 eta: float = 0.0, generator=None, latents: Optional[torch.Tensor] = None,
                 keylatents: Optional[torch.Tensor] = None, output_type: Optional[str] = "tensor", return_dict: bool = True,
                 callback: Optional[Callable[[int, int, torch.Tensor], None]] = None, callback_steps: Optional[int] = 1,
                 controlnet_images: torch.Tensor = None, controlnet_image_index: list = [0],
                 controlnet_conditioning_scale: Union[float, List[float]] = 1.0, low_strength=0.0,
                 text_embeddings: Optional[torch.Tensor] = None, noise: Optional[torch.Tensor] = None, **kwargs):
        height = height or self.unet.config.sample_size * self.vae_scale_factor
        width = width or self.unet.config.sample_size * self.vae_scale_factor
        self.check_inputs(prompt, height, width, callback_steps)
        if eta != 0.0 and hasattr(self.scheduler, "alpha_pair"):
            raise NotImplementedError("eta != 0 is not on the NEURONS path")

        batch_size = 1
        if latents is not None:
            batch_size = latents.shape[0]
        if isinstance(prompt, list):
            batch_size = len(prompt)
        device = self._execution_device
        if device.type != "cuda":
            raise RuntimeError("NeuroclipsPipeline runs on MI355X only: call .to('cuda') first (no CPU fallback)")
        do_classifier_free_guidance = guidance_scale > 1.0

        if text_embeddings is None:
            prompt = prompt if isinstance(prompt, list) else [prompt] * batch_size
            if negative_prompt is not None:
                negative_prompt = negative_prompt if isinstance(negative_prompt, list) else [negative_prompt] * batch_size
            text_embeddings = self._encode_prompt(prompt, device, num_videos_per_prompt, do_classifier_free_guidance, negative_prompt)
        text_embeddings = text_embeddings.to(device)
        want = batch_size * num_videos_per_prompt * (2 if do_classifier_free_guidance else 1)
        if text_embeddings.shape[0] != want:
            raise ValueError(f"text_embeddings batch {text_embeddings.shape[0]} != {want}")

        self.scheduler.set_timesteps(num_inference_steps, device=device)
        timesteps = self.scheduler.timesteps
        # This package's DDIMScheduler exposes host-side tables (timesteps_host / alpha_pair): CFG + the DDIM update then run as ONE HIP kernel.
        # Any other scheduler object (the diffusers surface the reference caller constructs, scripts/neuroclips_video.py:219) is driven exactly
        # as the reference drives it -- scale_model_input / step(...).prev_sample (pipeline_neuroclips.py:436,483) -- with the CFG combine in HIP.
        own_scheduler = hasattr(self.scheduler, "alpha_pair") and hasattr(self.scheduler, "timesteps_host")
        timesteps_host = self.scheduler.timesteps_host if own_scheduler else [int(v) for v in torch.as_tensor(timesteps).tolist()]
        extra_step_kwargs = {} if own_scheduler else self.prepare_extra_step_kwargs(generator, eta)

        num_channels_latents = self.unet.in_channels
        latents = self.prepare_latents(batch_size * num_videos_per_prompt, num_channels_latents, video_length, height, width,
                                       text_embeddings.dtype, device, generator, latents)
        # drawn and never used by the reference — kept for RNG-order parity (:395-405)
        keylatents = self.prepare_latents(batch_size * num_videos_per_prompt, num_channels_latents, video_length, height,
                                          width, text_embeddings.dtype, device, generator, keylatents)
        del keylatents
        latents_dtype = latents.dtype

        init_timestep = min(int(num_inference_steps * low_strength), num_inference_steps)
        t_start = max(num_inference_steps - init_timestep, 0)
        steps = self.scheduler.timesteps[:t_start]
        latent_timestep = steps[:1].repeat(batch_size)
        if noise is None:
            noise = torch.randn_like(latents)
        else:
            noise = noise.to(device=device, dtype=latents.dtype)
        latents = self.scheduler.add_noise(latents, noise, latent_timestep)
        latents = latents.to(torch.float32).contiguous()      # DDIM state stays fp32 for the whole loop

        use_ctrl = (getattr(self, "controlnet", None) is not None) and (controlnet_images is not None)
        if use_ctrl:
            assert controlnet_images.dim() == 5
            controlnet_images = controlnet_images.to(latents.device)
            cond_shape = list(controlnet_images.shape)
            cond_shape[2] = video_length
            controlnet_cond = torch.zeros(cond_shape, device=latents.device)
            mask_shape = list(cond_shape)
            mask_shape[1] = 1
            controlnet_conditioning_mask = torch.zeros(mask_shape, device=latents.device)
            assert controlnet_images.shape[2] >= len(controlnet_image_index)
            controlnet_cond[:, :, controlnet_image_index] = controlnet_images[:, :, :len(controlnet_image_index)]
            controlnet_conditioning_mask[:, :, controlnet_image_index] = 1
        import contextlib
        cleanup = contextlib.ExitStack()
        if use_ctrl and hasattr(self.controlnet, "set_condition_frames"):
            # the frames that carry a condition are known here: no tensor scan, no dependence on autograd version counters
            # (torch.inference_mode tensors have none).  The reset is registered BEFORE anything else can raise: a plan error or an
            # out-of-memory in forward_async must not leave this call's frame list on the controlnet object (it overrides the tensor
            # scan of later direct forwards)
            cleanup.callback(self.controlnet.set_condition_frames, None)
            self.controlnet.set_condition_frames([int(i) % video_length for i in controlnet_image_index])

        with cleanup, self.progress_bar(total=num_inference_steps) as progress_bar:
            lib = _lib.load()
            n_lat = latents.numel()
            for net in (self.unet, getattr(self, "controlnet", None)):
                if hasattr(net, "set_clip_samples"):       # deterministic-batch mode plans per clip: a clip is 2 samples with guidance, 1 without
                    net.set_clip_samples(2 if do_classifier_free_guidance else 1)
            fused = use_ctrl and hasattr(self.unet, "forward_with_controlnet") and \
                getattr(self.controlnet, "set_noisy_sample_input_to_zero", False) and self.overlap_controlnet
            # grouped schedule: G steps per SparseCtrl evaluation (controlnet_group_size)
            b2 = latents.shape[0] * (2 if do_classifier_free_guidance else 1)
            G = controlnet_group_size(len(timesteps_host), b2, latents.shape[2], latents.shape[3], latents.shape[4], self.controlnet_group) \
                if (fused and hasattr(self.controlnet, "forward_async")) else 1
            self.last_controlnet_group = G
            pending = {}

            def launch_group(g):
                ts = []
                for tt in groups[g]["timesteps"]:
                    ts += [float(tt)] * b2
                pending[g] = self.controlnet.forward_async(ts, ctx_group, controlnet_cond, controlnet_conditioning_mask,
                                                           controlnet_conditioning_scale, slot=groups[g]["slot"])
            if G > 1:
                groups, plan = controlnet_group_plan(list(timesteps_host), G)
                ctx_group = text_embeddings.repeat(G, 1, 1)
                launch_group(0)
            # the loop below builds the U-Net input as cat([latents] * 2) with one timestep (:435): the native U-Net may evaluate what precedes
            # the first cross-attention once per pair (exact; nr_net_set_cfg_pair_identical).  Own scheduler only: its scale_model_input is the
            # identity, a foreign one is not known to treat the two halves alike
            cfg_pair = bool(do_classifier_free_guidance and own_scheduler)
            for i, t in enumerate(timesteps_host):
                latent_model_input = torch.cat([latents] * 2) if do_classifier_free_guidance else latents
                latent_model_input = self.scheduler.scale_model_input(latent_model_input, t if own_scheduler else timesteps[i])
                down_res = mid_res = None
                if G > 1:
                    launch, g, p_ = plan[i]
                    if launch is not None:
                        launch_group(launch)         # runs beside this group's U-Net steps; its buffers were last read in step i - 1
                        pending.pop(g - 1, None)
                    noise_pred = self.unet.forward_after(self.controlnet, groups[g]["slot"], pending[g], p_ * b2, latent_model_input, t,
                                                         text_embeddings, cfg_pair_identical=cfg_pair).sample
                elif fused:
                    # same two network evaluations (:460-475), issued as one library call that overlaps them
                    # the next step's SparseCtrl evaluation (independent of the latents) is issued early
                    t_next = timesteps_host[i + 1] if (i + 1 < len(timesteps_host) and self.prefetch_controlnet) else None
                    noise_pred = self.unet.forward_with_controlnet(
                        self.controlnet, latent_model_input, t, text_embeddings, controlnet_cond,
                        controlnet_conditioning_mask, controlnet_conditioning_scale, next_timestep=t_next, cfg_pair_identical=cfg_pair).sample
                elif use_ctrl:
                    zc = {"zero_copy": True} if hasattr(self.controlnet, "_out_bufs") else {}    # consumed before the next call
                    down_res, mid_res = self.controlnet(
                        latent_model_input, t, encoder_hidden_states=text_embeddings, controlnet_cond=controlnet_cond,
                        conditioning_mask=controlnet_conditioning_mask, conditioning_scale=controlnet_conditioning_scale,
                        guess_mode=False, return_dict=False, **zc)
                if not fused and G == 1:
                    noise_pred = self.unet(latent_model_input, t, encoder_hidden_states=text_embeddings,
                                           down_block_additional_residuals=down_res,
                                           mid_block_additional_residual=mid_res).sample
                if own_scheduler:
                    # CFG combine + DDIM step fused in one HIP kernel (reference: :478-483)
                    a_t, a_prev = self.scheduler.alpha_pair(t)
                    new_latents = torch.empty_like(latents)
                    _lib.check(lib.nr_cfg_ddim_step(torch.cuda.current_stream().cuda_stream, noise_pred.data_ptr(),
                                                    latents.data_ptr(), new_latents.data_ptr(), n_lat, float(guidance_scale),
                                                    1 if do_classifier_free_guidance else 0, a_t, a_prev))
                    latents = new_latents
                else:
                    noise_pred = noise_pred.float().contiguous()
                    if do_classifier_free_guidance:
                        combined = torch.empty_like(latents)
                        _lib.check(lib.nr_cfg_combine(torch.cuda.current_stream().cuda_stream, noise_pred.data_ptr(), combined.data_ptr(),
                                                      n_lat, float(guidance_scale)))
                        noise_pred = combined
                    latents = self.scheduler.step(noise_pred, timesteps[i], latents, **extra_step_kwargs).prev_sample
                    latents = latents.to(torch.float32).contiguous()
                progress_bar.update()
                if callback is not None and i % callback_steps == 0:
                    callback(i, t, latents)

        latents = latents.to(latents_dtype)
        if output_type == "latent":
            video = latents
        else:
            video = self.decode_latents(latents)
            if output_type == "tensor":
                video = torch.from_numpy(video)
        if not return_dict:
            return video
        return AnimationPipelineOutput(videos=video)
