"""Dump per-launch timings (HIP events) of one U-Net and one SparseCtrl forward at BASELINE config 2 shapes.
Usage (GPU box): python tools/per_op_profile.py gpurun_out/ops_unet.csv gpurun_out/ops_ctrl.csv"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import gpu_random_state_dict  # noqa: E402
from neurons_amd import _lib, NativeSparseCtrl, NativeUNet3D  # noqa: E402
from neurons_amd.sparsectrl import controlnet_config_from_unet  # noqa: E402
from neurons_amd.unet3d import UNet3DConfig, state_dict_schema  # noqa: E402

dev = torch.device("cuda", 0)
ucfg = UNet3DConfig()
ccfg = controlnet_config_from_unet(ucfg, dict(set_noisy_sample_input_to_zero=True, use_simplified_condition_embedding=True,
                                              conditioning_channels=4,
                                              motion_module_kwargs=dict(attention_block_types=["Temporal_Self"],
                                                                        temporal_position_encoding_max_len=32)))
unet, ctrl = NativeUNet3D(ucfg).to(dev), NativeSparseCtrl(ccfg).to(dev)
unet.load_state_dict({k: v.cpu() for k, v in gpu_random_state_dict(state_dict_schema(ucfg, 0), 1, dev).items()})
ctrl.load_state_dict({k: v.cpu() for k, v in gpu_random_state_dict(state_dict_schema(ccfg, 1), 2, dev).items()})
F, L = 16, 32
x = torch.randn(2, 4, F, L, L, device=dev)
ctx = torch.randn(2, 77, 768, device=dev)
cond = torch.zeros(1, 4, F, L, L, device=dev)
mask = torch.zeros(1, 1, F, L, L, device=dev)
for _ in range(2):
    down, mid = ctrl(x, 500, encoder_hidden_states=ctx, controlnet_cond=cond, conditioning_mask=mask, return_dict=False)
    unet(x, 500, encoder_hidden_states=ctx, down_block_additional_residuals=down, mid_block_additional_residual=mid)
torch.cuda.synchronize()
for net, path in ((unet, sys.argv[1]), (ctrl, sys.argv[2])):
    os.environ["NR_PROFILE_CSV"] = path
    for _ in range(2):
        p = net.profile_last()
    print(path, {k: round(v["ms"], 3) for k, v in p.items()})
