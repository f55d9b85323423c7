"""ctypes binding of libneurons_amd.so (the C ABI declared in include/neurons_amd.h).

The reference is pure Python, so this module *is* the FFI stub a maintainer would add.
The product path has no CPU fallback: if the shared library is missing or no HIP device is present,
loading / creating a network raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# NR_LIB_VARIANT=pk loads libneurons_amd_pk.so: the SAME sources built WITH packed fp32 VALU ops (make -C neurons_amd/csrc pk), the
# other arm of the two-stream determinism A/B (tools/race_gn.py, profiles/r03_race_*); never the product library
LIB_PATH = os.path.join(_HERE, "libneurons_amd" + ("_" + os.environ["NR_LIB_VARIANT"] if os.environ.get("NR_LIB_VARIANT") else "") + ".so")

NR_KIND_UNET3D = 0
NR_KIND_SPARSECTRL = 1
NR_KIND_SGM_UNET = 2
NR_KIND_VAE_DECODER = 3
NR_KIND_VAE_ENCODER = 4
NR_KIND_CLIP_TEXT = 5
NR_KIND_LEAF_TRANSFORMER3D = 6
NR_KIND_LEAF_TEMPORAL = 7
NR_MAX_LEVELS = 4


class NrNetConfig(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("in_channels", C.c_int32),
        ("out_channels", C.c_int32),
        ("num_levels", C.c_int32),
        ("block_out_channels", C.c_int32 * NR_MAX_LEVELS),
        ("down_block_has_attn", C.c_int32 * NR_MAX_LEVELS),
        ("up_block_has_attn", C.c_int32 * NR_MAX_LEVELS),
        ("layers_per_block", C.c_int32),
        ("num_heads", C.c_int32),
        ("cross_attention_dim", C.c_int32),
        ("norm_num_groups", C.c_int32),
        ("norm_eps", C.c_float),
        ("use_motion_module", C.c_int32),
        ("motion_num_heads", C.c_int32),
        ("motion_num_attention_blocks", C.c_int32),
        ("motion_pe_max_len", C.c_int32),
        ("motion_module_mid_block", C.c_int32),
        ("conditioning_channels", C.c_int32),
        ("set_noisy_sample_input_to_zero", C.c_int32),
        ("transformer_depth", C.c_int32 * NR_MAX_LEVELS),
        ("num_head_channels", C.c_int32),
        ("adm_in_channels", C.c_int32),
    ]


NR_PROF_KINDS = 5
NR_PROF_NAMES = ("igemm", "groupnorm", "layernorm", "attention", "other")


class NrProfile(C.Structure):
    _fields_ = [("ms", C.c_double * NR_PROF_KINDS), ("flops", C.c_double * NR_PROF_KINDS),
                ("bytes", C.c_double * NR_PROF_KINDS), ("launches", C.c_int32 * NR_PROF_KINDS)]


# every symbol include/neurons_amd.h declares: name -> (restype, argtypes)
_VP = C.c_void_p
_FP = C.POINTER(C.c_float)
_I32 = C.c_int32
_I64 = C.c_int64
SYMBOLS = {
    "nr_net_create": (_I32, [C.POINTER(NrNetConfig), C.POINTER(_VP)]),
    "nr_net_destroy": (None, [_VP]),
    "nr_last_error": (C.c_char_p, []),
    "nr_net_load_tensor": (_I32, [_VP, C.c_char_p, _VP, C.POINTER(_I64), _I32]),
    "nr_net_plan": (_I32, [_VP, _I32, _I32, _I32, _I32, _I32]),
    "nr_net_release_host_weights": (_I32, [_VP]),
    "nr_net_invalidate_context": (_I32, [_VP]),
    "nr_net_set_graph": (_I32, [_VP, _I32]),
    "nr_net_workspace_bytes": (_I64, [_VP]),
    "nr_net_weight_bytes": (_I64, [_VP]),
    "nr_net_num_residuals": (_I32, [_VP]),
    "nr_net_residual_shape": (_I32, [_VP, _I32, C.POINTER(_I32), C.POINTER(_I32), C.POINTER(_I32)]),
    "nr_unet3d_forward": (_I32, [_VP, _VP, _VP, _FP, _VP, _I32, C.POINTER(_VP), _VP, _VP]),
    "nr_sparsectrl_forward": (_I32, [_VP, _VP, _VP, _FP, _VP, _I32, _VP, _VP, _I32, C.c_float, C.POINTER(_VP), _VP]),
    "nr_denoise_step_forward": (_I32, [_VP, _VP, _VP, _VP, _FP, _VP, _I32, _VP, _VP, _I32, C.c_float, C.POINTER(_VP), _VP, _VP, _FP]),
    "nr_sparsectrl_forward_async": (_I32, [_VP, _VP, _FP, _VP, _I32, _VP, _VP, _I32, C.c_float, C.POINTER(_VP), _VP, _I32]),
    "nr_unet3d_forward_after": (_I32, [_VP, _VP, _I32, _VP, _VP, _FP, _VP, _I32, C.POINTER(_VP), _VP, _VP]),
    "nr_sgm_unet_forward": (_I32, [_VP, _VP, _VP, C.c_float, _FP, _VP, _I32, _VP, _VP]),
    "nr_vae_decode": (_I32, [_VP, _VP, _VP, C.c_float, C.c_float, C.c_float, _I32, _VP]),
    "nr_clip_text_forward": (_I32, [_VP, _VP, _VP, _VP]),
    "nr_vae_encode": (_I32, [_VP, _VP, _VP, C.c_float, C.c_float, _VP]),
    "nr_gaussian_sample": (_I32, [_VP, _VP, _VP, _VP, _I32, _I32, _I32, C.c_float]),
    "nr_edm_cfg_euler_step": (_I32, [_VP, _VP, _VP, _VP, _I64, C.c_float, C.c_float, C.c_float, C.c_float]),
    "nr_prior_p_sample_step": (_I32, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _I64, C.c_float, _I32, _I32, C.c_double, C.c_double, C.c_double]),
    "nr_cfg_ddim_step": (_I32, [_VP, _VP, _VP, _VP, _I64, C.c_float, _I32, C.c_double, C.c_double]),
    "nr_cfg_combine": (_I32, [_VP, _VP, _VP, _I64, C.c_float]),
    "nr_net_export_manifest": (_I64, [_VP, _VP, _I64, C.POINTER(_I64)]),
    "nr_net_export_weights": (_I32, [_VP, _VP, _VP, _I64]),
    "nr_net_import_weights": (_I32, [_VP, _VP, C.c_char_p, _I64, _VP, _I64]),
    "nr_net_profile_last": (_I32, [_VP, _VP, C.POINTER(NrProfile)]),
    "nr_net_set_attention_fp8": (_I32, [_VP, _I32]),
    "nr_net_set_deterministic_batch": (_I32, [_VP, _I32]),
    "nr_net_set_clip_samples": (_I32, [_VP, _I32]),
    "nr_net_set_cfg_pair_identical": (_I32, [_VP, _I32]),
    "nr_sparsectrl_set_condition_frames": (_I32, [_VP, C.POINTER(_I32), _I32]),
    "nr_net_set_debug": (_I32, [_VP, _I32]),
    "nr_net_num_taps": (_I32, [_VP]),
    "nr_net_tap_name": (C.c_char_p, [_VP, _I32]),
    "nr_net_read_tap": (_I32, [_VP, _I32, _VP, _I64, C.POINTER(_I32), C.POINTER(_I32)]),
    "nr_g8p_set_mode": (None, [_I32]),
    "nr_g8p_set_phases": (None, [_I32]),
    "nr_ff_set_waves": (None, [_I32]),
    "nr_net_num_ops": (_I32, [_VP]),
    "nr_net_op_desc": (C.c_char_p, [_VP, _I32]),
    "nr_leaf_forward": (_I32, [_VP, _VP, _VP, _VP, _I32, _VP]),
    "nr_op_fm_cache_clear": (None, []),
    "nr_op_gemm": (_I32, [_VP, _VP, _I32, _VP, _VP, _VP, _I32, _VP, _I32, _I32, _I32, _I32, _I32]),
    "nr_op_gemm2": (_I32, [_VP, _VP, _I32, _I32, _VP, _I32, _I32, _VP, _VP, _VP, _I32, _VP, _I32, _I32, _I32]),
    "nr_op_ln_gemm": (_I32, [_VP, _VP, _I32, _VP, _VP, _VP, C.c_float, _VP, _I32, _VP, _I32, _I32, _I32, _I32, _I32, _I32]),
    "nr_op_gemm_ex": (_I32, [_VP, _VP, _I32, _VP, _VP, _VP, C.c_float, _VP, _I32, _I32, _I32, _VP, _I32, _VP, _I32, _I32, _I32, _I32, _I32, _I32,
                             C.c_float]),
    "nr_op_conv3x3": (_I32, [_VP, _VP, _I32, _VP, _I32, _I32, _I32, _I32, _I32, _I32, _VP, _VP, _VP, _I32, _VP, _VP, _I32]),
    "nr_op_conv3x3_tap_inner": (_I32, [_VP, _VP, _I32, _I32, _I32, _I32, _VP, _VP, _VP, _I32, _VP, _VP, _I32]),
    "nr_op_groupnorm": (_I32, [_VP, _VP, _I32, _VP, _I32, _I32, _I32, _I32, _VP, _VP, C.c_float, _I32, _VP, _VP]),
    "nr_op_layernorm": (_I32, [_VP, _VP, _VP, _I32, _I32, _VP, _VP, C.c_float, _VP, _I32, _I32]),
    "nr_op_attention": (_I32, [_VP, _I32, _VP, _VP, _VP, _I32, _I32, _I32, _I32, _I32, _I32, _I32]),
    "nr_op_ff_fused": (_I32, [_VP, _VP, _VP, _VP, _I32, _I32, _VP, _VP, _VP, _VP, _VP, _VP, C.c_float]),
    "nr_op_tattn_fused": (_I32, [_VP, _VP, _I32, _I32, _VP, _VP, _VP, _VP, _VP, _VP, _VP, C.c_float]),
    "nr_op_xattn_fused": (_I32, [_VP, _VP, _I32, _I32, _I32, _VP, _VP, _VP, _I32, _I32, _I32, _VP, _VP, _VP, C.c_float]),
    "nr_op_tattn_fused_frames": (_I32, [_VP, _VP, _I32, _I32, _I32, _VP, _VP, _VP, _VP, _VP, _VP, _VP, C.c_float]),
    "nr_op_tattn_head": (_I32, [_VP, _VP, _VP, _I32, _I32, _I32, _VP, _VP, _VP, _VP, C.c_float]),
    "nr_op_xattn_head": (_I32, [_VP, _VP, _VP, _I32, _I32, _I32, _I32, _VP, _VP, _VP, _VP, _I32, _I32, _I32, C.c_float]),
}

_lib = None


def load():
    """Load the shared library and bind every declared symbol.  Raises if it is missing: the HIP
    extension is the product, there is nothing to fall back to."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C neurons_amd/csrc).  neurons_amd has no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status):
    if status != 0:
        msg = load().nr_last_error()
        raise RuntimeError(f"neurons_amd: status {status}: {msg.decode() if msg else ''}")
