"""Dump per-launch timings (HIP events) of one U-Net and one SparseCtrl forward at BASELINE config 2 shapes (default), or at
another shape: [clips frames latent [ctrl_group]] (config 4: 8 16 32 4; config 5 per clip: 1 32 64).
Usage (GPU box): python tools/per_op_profile.py gpurun_out/ops_unet.csv gpurun_out/ops_ctrl.csv [clips frames latent [ctrl_group]]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import gpu_random_state_dict  # noqa: E402
from neurons_amd import _lib, NativeSparseCtrl, NativeUNet3D  # noqa: E402
from neurons_amd.sparsectrl import controlnet_config_from_unet  # noqa: E402
from neurons_amd.unet3d import UNet3DConfig, state_dict_schema  # noqa: E402

dev = torch.device("cuda", 0)
B, F, L = (int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (1, 16, 32)
G = int(sys.argv[6]) if len(sys.argv) > 6 else 1          # SparseCtrl evaluated for G steps at once (the grouped schedule's batch)
ucfg = UNet3DConfig()
if F > 24:
    ucfg.motion_module_kwargs = dict(ucfg.motion_module_kwargs, temporal_position_encoding_max_len=32)
ccfg = controlnet_config_from_unet(ucfg, dict(set_noisy_sample_input_to_zero=True, use_simplified_condition_embedding=True,
                                              conditioning_channels=4,
                                              motion_module_kwargs=dict(attention_block_types=["Temporal_Self"],
                                                                        temporal_position_encoding_max_len=32)))
unet, ctrl = NativeUNet3D(ucfg).to(dev), NativeSparseCtrl(ccfg).to(dev)
unet.load_state_dict({k: v.cpu() for k, v in gpu_random_state_dict(state_dict_schema(ucfg, 0), 1, dev).items()})
ctrl.load_state_dict({k: v.cpu() for k, v in gpu_random_state_dict(state_dict_schema(ccfg, 1), 2, dev).items()})
x = torch.randn(2 * B, 4, F, L, L, device=dev)
ctx = torch.randn(2 * B, 77, 768, device=dev)
cond = torch.zeros(B, 4, F, L, L, device=dev)
cond[:, :, 0] = torch.randn(B, 4, L, L, device=dev)
mask = torch.zeros(B, 1, F, L, L, device=dev)
mask[:, :, 0] = 1
for _ in range(2):
    down, mid = ctrl(x, 500, encoder_hidden_states=ctx, controlnet_cond=cond, conditioning_mask=mask, return_dict=False)
    unet(x, 500, encoder_hidden_states=ctx, down_block_additional_residuals=down, mid_block_additional_residual=mid)
if G > 1:      # the profile of the grouped evaluation: the same network on G x the CFG batch
    xg, cg = torch.cat([x] * G), torch.cat([ctx] * G)
    for _ in range(2):
        ctrl(xg, 500, encoder_hidden_states=cg, controlnet_cond=cond, conditioning_mask=mask, return_dict=False)
torch.cuda.synchronize()
for net, path in ((unet, sys.argv[1]), (ctrl, sys.argv[2])):
    os.environ["NR_PROFILE_CSV"] = path
    for _ in range(2):
        p = net.profile_last()
    print(path, {k: round(v["ms"], 3) for k, v in p.items()})
