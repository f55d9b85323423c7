"""Tiny full-topology configurations shared by the golden-vector generator (oracle/gen_golden.py, build container only)
and by the tests that replay those vectors.  Pure configuration: no reference import, no oracle import."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def tiny_unet_config():
    from neurons_amd.unet3d import UNet3DConfig
    return UNet3DConfig(sample_size=8, block_out_channels=(64, 64, 128, 128), cross_attention_dim=64)


def tiny_ctrl_config():
    from neurons_amd.sparsectrl import controlnet_config_from_unet
    return controlnet_config_from_unet(tiny_unet_config(), dict(
        set_noisy_sample_input_to_zero=True, use_simplified_condition_embedding=True, conditioning_channels=4,
        use_motion_module=True, motion_module_resolutions=[1, 2, 4, 8], motion_module_mid_block=False,
        motion_module_type="Vanilla",
        motion_module_kwargs=dict(num_attention_heads=8, num_transformer_block=1, attention_block_types=["Temporal_Self"],
                                  temporal_position_encoding=True, temporal_position_encoding_max_len=32,
                                  temporal_attention_dim_div=1)))


def tiny_sgm_config():
    from neurons_amd.sgm import SGMUNetConfig
    return SGMUNetConfig(model_channels=64, channel_mult=(1, 2, 4), num_res_blocks=2, attention_resolutions=(4, 2),
                         num_head_channels=32, transformer_depth=(1, 2, 3), context_dim=128, adm_in_channels=64)


def tiny_vae_config():
    from neurons_amd.vae import VAEDecoderConfig
    return VAEDecoderConfig(ch=64, ch_mult=(1, 1, 2, 2), num_res_blocks=2)


def tiny_clip_config():
    from neurons_amd.clip import CLIPTextConfig
    return CLIPTextConfig(vocab_size=1000, hidden_size=128, intermediate_size=512, num_hidden_layers=3, num_attention_heads=4,
                          max_position_embeddings=77)
