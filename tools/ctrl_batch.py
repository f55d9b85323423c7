"""SparseCtrl forward time vs batch (2 = one DDIM step's CFG pair, 4 = two steps evaluated together: the network does not see the latents)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import gpu_random_state_dict  # noqa: E402
from neurons_amd import NativeSparseCtrl  # noqa: E402
from neurons_amd.sparsectrl import controlnet_config_from_unet  # noqa: E402
from neurons_amd.unet3d import UNet3DConfig, state_dict_schema  # noqa: E402

dev = torch.device("cuda", 0)
ucfg = UNet3DConfig()
ccfg = controlnet_config_from_unet(ucfg, dict(set_noisy_sample_input_to_zero=True, use_simplified_condition_embedding=True, conditioning_channels=4,
                                              motion_module_kwargs=dict(attention_block_types=["Temporal_Self"], temporal_position_encoding_max_len=32)))
sd = {k: v.cpu() for k, v in gpu_random_state_dict(state_dict_schema(ccfg, 1), 2, dev).items()}
F, L = 16, 32
for B in (2, 4, 6, 8):
    ctrl = NativeSparseCtrl(ccfg).to(dev)
    ctrl.load_state_dict(sd)
    x = torch.randn(B, 4, F, L, L, device=dev)
    ctx = torch.randn(B, 77, 768, device=dev)
    cond = torch.zeros(1, 4, F, L, L, device=dev)
    mask = torch.zeros(1, 1, F, L, L, device=dev)
    ts = torch.full((B,), 500.0, device=dev)
    for _ in range(3):
        ctrl(x, ts, encoder_hidden_states=ctx, controlnet_cond=cond, conditioning_mask=mask, return_dict=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        ctrl(x, ts, encoder_hidden_states=ctx, controlnet_cond=cond, conditioning_mask=mask, return_dict=False)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    print(f"SparseCtrl batch {B}: {ms:.3f} ms per forward = {ms / (B // 2):.3f} ms per DDIM step", flush=True)
    del ctrl
    torch.cuda.empty_cache()
