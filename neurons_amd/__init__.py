"""neurons_amd — MI355X-native denoising hot path of xmed-lab/NEURONS (video reconstruction).

Host side mirrors the reference's Python interface for this path (same names, arguments, errors):
  NativeUNet3D        ~ animatediff/models/unet.py:38        UNet3DConditionModel
  NativeSparseCtrl    ~ animatediff/models/sparse_controlnet.py:85  SparseControlNetModel
  DDIMScheduler       ~ diffusers==0.11.1 DDIMScheduler (call sites pipeline_neuroclips.py:378-483)
  NeuroclipsPipeline  ~ animatediff/pipelines/pipeline_neuroclips.py:43
  sgm.NativeSGMUNet   ~ generative_models/sgm/modules/diffusionmodules/openaimodel.py:472 (unCLIP keyframes)
  vae.NativeVAEDecoder ~ generative_models/sgm/modules/diffusionmodules/model.py:612 Decoder (= AutoencoderKL.decode)
All arithmetic of the two networks runs in libneurons_amd.so (hand-written HIP for gfx950) behind the
C ABI in include/neurons_amd.h; PyTorch supplies device memory, streams and torch.distributed only.
"""
from .scheduler import DDIMScheduler  # noqa: F401
from .unet3d import NativeUNet3D, UNet3DConfig  # noqa: F401
from .sparsectrl import NativeSparseCtrl  # noqa: F401
from .pipeline import NeuroclipsPipeline  # noqa: F401
