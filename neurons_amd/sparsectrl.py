"""NativeSparseCtrl — drop-in for ``animatediff.models.sparse_controlnet.SparseControlNetModel`` (inference).

Forward signature / return value follow sparse_controlnet.py:450-465,576-581: returns
``(down_block_res_samples: list of 12 tensors, mid_block_res_sample)`` with logical shape ``(b, C, f, h, w)``.
Physically the tensors are channels-last bf16 views of buffers written by ``nr_sparsectrl_forward`` so that
``NativeUNet3D`` consumes them with zero copies; any other consumer sees ordinary (non-contiguous) tensors.
"""
import ctypes as C
import os
from dataclasses import dataclass, replace
from typing import List, Optional

import torch

from . import _lib
from .unet3d import _on_device, UNet3DConfig, _NativeNet, tensor_version


@dataclass
class SparseControlNetOutput:
    down_block_res_samples: List[torch.Tensor]
    mid_block_res_sample: torch.Tensor


def controlnet_config_from_unet(unet_config: UNet3DConfig, controlnet_additional_kwargs: Optional[dict] = None) -> UNet3DConfig:
    """``SparseControlNetModel.from_unet`` (sparse_controlnet.py:316-345): copies the U-Net geometry, then applies
    ``controlnet_additional_kwargs`` (configs/inference/sparsectrl/latent_condition.yaml)."""
    kw = dict(controlnet_additional_kwargs or {})
    mm = dict(num_attention_heads=8, num_transformer_block=1, attention_block_types=("Temporal_Self",),
              temporal_position_encoding=True, temporal_position_encoding_max_len=32, temporal_attention_dim_div=1)
    mm.update(kw.pop("motion_module_kwargs", {}))
    mm["attention_block_types"] = tuple(mm["attention_block_types"])
    cfg = replace(unet_config, motion_module_kwargs=mm,
                  use_motion_module=kw.pop("use_motion_module", True),
                  motion_module_resolutions=tuple(kw.pop("motion_module_resolutions", (1, 2, 4, 8))),
                  motion_module_mid_block=kw.pop("motion_module_mid_block", False),
                  motion_module_type=kw.pop("motion_module_type", "Vanilla"),
                  conditioning_channels=kw.pop("conditioning_channels", 3),
                  set_noisy_sample_input_to_zero=kw.pop("set_noisy_sample_input_to_zero", False),
                  use_simplified_condition_embedding=kw.pop("use_simplified_condition_embedding", False),
                  concate_conditioning_mask=kw.pop("concate_conditioning_mask", True))
    if kw:
        raise TypeError(f"unexpected controlnet kwargs: {sorted(kw)}")
    return cfg


class NativeSparseCtrl(_NativeNet):
    _kind = _lib.NR_KIND_SPARSECTRL

    def __init__(self, config: Optional[UNet3DConfig] = None, **kwargs):
        super().__init__(config, **kwargs)
        self.use_simplified_condition_embedding = self.config.use_simplified_condition_embedding  # neuroclips_video.py:278
        self.set_noisy_sample_input_to_zero = self.config.set_noisy_sample_input_to_zero
        self._out_bufs = None

    @classmethod
    def from_unet(cls, unet, controlnet_additional_kwargs: Optional[dict] = None):
        return cls(controlnet_config_from_unet(unet.config, controlnet_additional_kwargs))

    # ---- identical-frame evaluation (C ABI nr_sparsectrl_set_condition_frames) ---------------------------------------------------------
    def set_condition_frames(self, frames):
        """Explicit frame list for the identical-frame evaluation: the caller states which frames carry a condition (the pipeline knows its
        ``controlnet_image_index``), so no tensor is scanned.  ``None`` returns to deriving the list from the tensors."""
        self._cframes_explicit = None if frames is None else tuple(sorted({int(i) for i in frames}))

    def _sync_condition_frames(self, controlnet_cond, conditioning_mask):
        """Frames whose condition or mask is not all zero.  Every other frame is exactly zero, which is what makes the engine's shortcut exact;
        with the noisy sample NOT zeroed frames differ anyway: off.  Source of the list, in order: NR_CTRL_DEDUP=0 (off, checked on every
        call) -> ``set_condition_frames`` (explicit, the pipeline's path) -> a scan of the tensors (one small reduction + host read), cached
        on (data_ptr, in-place version, shape).  The cache cannot see writes that do not bump the version counter (``cond.data[...] = x``, a
        native kernel writing through ``data_ptr()`` into a reused buffer): such callers pass the list explicitly or call
        ``invalidate_condition_frames()``.  Inference-mode tensors carry no version counter: they are scanned on every call."""
        if os.environ.get("NR_CTRL_DEDUP", "1") == "0" or not self.config.set_noisy_sample_input_to_zero:
            frames = None
        elif getattr(self, "_cframes_explicit", None) is not None:
            f = controlnet_cond.shape[2]
            frames = tuple(sorted({i % f for i in self._cframes_explicit}))
        else:
            vc, vm = tensor_version(controlnet_cond), tensor_version(conditioning_mask)
            key = (controlnet_cond.data_ptr(), vc, conditioning_mask.data_ptr(), vm, tuple(controlnet_cond.shape)) \
                if isinstance(vc, int) and isinstance(vm, int) else None          # no counter ("Inference tensors do not track version counter")
            if key is not None and getattr(self, "_cframes_key", None) == key:
                frames = self._cframes_scanned
            else:
                nz = (controlnet_cond != 0).flatten(3).any(-1).any(1).any(0) | (conditioning_mask != 0).flatten(3).any(-1).any(1).any(0)
                frames = tuple(int(i) for i in torch.nonzero(nz).flatten().tolist())
                self._cframes_key = key
                self._cframes_scanned = frames
                self._cframes_ref = (controlnet_cond, conditioning_mask) if key is not None else None    # keeps data_ptr from being reused
        if getattr(self, "_cframes", "unset") != frames:
            if frames is None:
                _lib.check(_lib.load().nr_sparsectrl_set_condition_frames(self._handle(), None, -1))
            else:
                arr = (C.c_int32 * max(1, len(frames)))(*frames)
                _lib.check(_lib.load().nr_sparsectrl_set_condition_frames(self._handle(), arr, len(frames)))
            self._cframes = frames
            self._plan_key = None           # the launch plan depends on the frame list

    def invalidate_condition_frames(self):
        """Forget the cached scan (after an out-of-band write into a condition tensor that is passed again)."""
        self._cframes_key = None
        self._cframes_ref = None

    def _on_plan(self):
        b, f, h, w, L = self._plan_key
        dev = self.device
        lib = _lib.load()
        n = int(lib.nr_net_num_residuals(self._h))
        bufs = []
        for i in range(n + 1):
            cc, hh, ww = C.c_int32(), C.c_int32(), C.c_int32()
            _lib.check(lib.nr_net_residual_shape(self._h, i, C.byref(cc), C.byref(hh), C.byref(ww)))
            bufs.append(torch.empty(b, f, hh.value, ww.value, cc.value, dtype=torch.bfloat16, device=dev))
        self._out_bufs = bufs
        self._out_ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in bufs[:n]])
        self._io_ctx = torch.empty(b, L, self.config.cross_attention_dim, dtype=torch.float32, device=dev)
        self._io_sample = torch.empty(b, self.config.in_channels, f, h, w, dtype=torch.float32, device=dev)
        self._io_cond = None

    @_on_device
    def forward(self, sample, timestep, encoder_hidden_states, controlnet_cond, conditioning_mask=None,
                conditioning_scale: float = 1.0, class_labels=None, attention_mask=None, cross_attention_kwargs=None,
                guess_mode: bool = False, return_dict: bool = True, zero_copy: bool = False):
        """``SparseControlNetModel.forward`` (sparse_controlnet.py:450-581).  Like the reference it returns FRESH tensors: the
        residuals are cloned out of the engine's persistent output buffers, which the next forward on this handle overwrites.
        ``zero_copy=True`` (used by the pipeline, which consumes the residuals before the next call) returns strided views of
        those buffers instead: valid only until the next forward."""
        if class_labels is not None or attention_mask is not None:
            raise NotImplementedError("class_labels / attention_mask are not used on the NEURONS path")
        if guess_mode:
            raise NotImplementedError("guess_mode is not used on the NEURONS path (pipeline_neuroclips.py:466)")
        if conditioning_mask is None:
            raise ValueError("conditioning_mask is required (concate_conditioning_mask=True)")
        if not sample.is_cuda:
            raise RuntimeError("NativeSparseCtrl.forward: CUDA (ROCm) tensors required; there is no CPU fallback")
        b, c, f, h, w = sample.shape
        ctx = encoder_hidden_states
        if b % ctx.shape[0] != 0:
            raise ValueError("encoder_hidden_states batch must divide the sample batch")
        if ctx.shape[0] != b:     # sparse_controlnet.py:491
            ctx = ctx.repeat(b // ctx.shape[0], 1, 1)
        cb = controlnet_cond.shape[0]
        if controlnet_cond.dim() != 5 or conditioning_mask.dim() != 5:
            raise ValueError("controlnet_cond / conditioning_mask must be (b, c, f, h, w)")
        if controlnet_cond.shape[1] != self.config.conditioning_channels or tuple(controlnet_cond.shape[2:]) != (f, h, w):
            raise ValueError(f"controlnet_cond shape {tuple(controlnet_cond.shape)} does not match sample {tuple(sample.shape)}")
        if b % cb != 0 or conditioning_mask.shape[0] != cb:
            raise ValueError("controlnet_cond batch must divide the sample batch (it is broadcast over the CFG halves)")
        self._sync_condition_frames(controlnet_cond, conditioning_mask)
        self._ensure_plan(b, f, h, w, ctx.shape[1])
        lib = _lib.load()
        self._set_context(ctx)
        if self._io_cond is None or self._io_cond.shape[0] != cb:
            self._io_cond = torch.empty(cb, self.config.conditioning_channels, f, h, w, dtype=torch.float32, device=sample.device)
            self._io_mask = torch.empty(cb, 1, f, h, w, dtype=torch.float32, device=sample.device)
        self._cond_key = None       # the fused step (unet3d.forward_with_controlnet) must re-stage its condition
        self._io_cond.copy_(controlnet_cond)
        self._io_mask.copy_(conditioning_mask)
        sample_ptr = None
        if not self.config.set_noisy_sample_input_to_zero:
            self._io_sample.copy_(sample)
            sample_ptr = self._io_sample.data_ptr()
        ts = self._timesteps_host(timestep, b)
        stream = torch.cuda.current_stream().cuda_stream
        n = len(self._out_bufs) - 1
        _lib.check(lib.nr_sparsectrl_forward(self._h, stream, sample_ptr, ts, self._io_ctx.data_ptr(), ctx.shape[1],
                                             self._io_cond.data_ptr(), self._io_mask.data_ptr(), cb,
                                             float(conditioning_scale), self._out_ptrs, self._out_bufs[n].data_ptr()))
        # logical (b, C, f, h, w) views over the channels-last buffers
        views = [(t if zero_copy else t.clone()).permute(0, 4, 1, 2, 3) for t in self._out_bufs]
        down, mid = views[:n], views[n]
        if not return_dict:
            return (down, mid)
        return SparseControlNetOutput(down_block_res_samples=down, mid_block_res_sample=mid)

    __call__ = forward

    # ---- grouped schedule (C ABI nr_sparsectrl_forward_async; pipeline.py owns the schedule) ---------------------------------------
    def _residual_set(self, slot):
        """Output buffers of evaluation slot 0 / 1 (slot 0 = the buffers of the synchronous forward)."""
        if slot == 0:
            return self._out_bufs, self._out_ptrs
        if getattr(self, "_out_bufs2_plan", None) != self._plan_key:
            self._out_bufs2 = [torch.empty_like(t) for t in self._out_bufs]
            n = len(self._out_bufs2) - 1
            self._out_ptrs2 = (C.c_void_p * n)(*[t.data_ptr() for t in self._out_bufs2[:n]])
            self._out_bufs2_plan = self._plan_key
        return self._out_bufs2, self._out_ptrs2

    @_on_device
    def forward_async(self, timesteps, encoder_hidden_states, controlnet_cond, conditioning_mask, conditioning_scale: float = 1.0,
                      slot: int = 0, frames=None):
        """Evaluate SparseCtrl for ``len(timesteps)`` samples (several DDIM steps x the CFG batch) on the engine's own stream, without
        joining the caller's stream.  Valid only with ``set_noisy_sample_input_to_zero`` (the evaluation does not see the latents).
        Returns the slot's channels-last output buffers ``[(b, f, h, w, C)] * n + [mid]``; consumers wait through
        ``NativeUNet3D.forward_after(..., slot=slot)``."""
        if not self.config.set_noisy_sample_input_to_zero:
            raise NotImplementedError("forward_async requires set_noisy_sample_input_to_zero=True")
        ctx = encoder_hidden_states
        b = len(timesteps)
        cb, _, f, h, w = controlnet_cond.shape
        if ctx.shape[0] != b or b % cb != 0 or conditioning_mask.shape[0] != cb:
            raise ValueError("forward_async: context batch must equal len(timesteps); the condition batch must divide it")
        self._sync_condition_frames(controlnet_cond, conditioning_mask)
        self._ensure_plan(b, f, h, w, ctx.shape[1])
        self._set_context(ctx)
        if self._io_cond is None or self._io_cond.shape[0] != cb:
            self._io_cond = torch.empty(cb, self.config.conditioning_channels, f, h, w, dtype=torch.float32, device=ctx.device)
            self._io_mask = torch.empty(cb, 1, f, h, w, dtype=torch.float32, device=ctx.device)
            self._cond_key = None
        ckey = (controlnet_cond.data_ptr(), tensor_version(controlnet_cond), conditioning_mask.data_ptr(), tensor_version(conditioning_mask),
                tuple(controlnet_cond.shape), self._plan_key)
        if getattr(self, "_cond_key", None) != ckey:      # staged once per clip: an evaluation in flight may be reading it
            self._io_cond.copy_(controlnet_cond)
            self._io_mask.copy_(conditioning_mask)
            self._cond_key = ckey
            self._cond_ref = (controlnet_cond, conditioning_mask)
        bufs, ptrs = self._residual_set(slot)
        ts = (C.c_float * b)(*[float(t) for t in timesteps])
        n = len(bufs) - 1
        _lib.check(_lib.load().nr_sparsectrl_forward_async(
            self._h, torch.cuda.current_stream().cuda_stream, ts, self._io_ctx.data_ptr(), ctx.shape[1], self._io_cond.data_ptr(),
            self._io_mask.data_ptr(), cb, float(conditioning_scale), ptrs, bufs[n].data_ptr(), int(slot)))
        return bufs
