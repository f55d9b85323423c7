"""Host-side sanitizer run of the engine (SURVEY 5 asks for one): the planner, the first-fit arena, every
weight conversion, the manifest parser, the graph-slot LRU, reload invalidation and the error paths of libneurons_amd's C ABI, compiled
HOST-ONLY with AddressSanitizer + UndefinedBehaviorSanitizer against a stand-in HIP runtime whose device memory is host heap memory
(tests/sanitize/).  No GPU: kernel launches are no-ops; what is checked is that the host code never reads or writes outside its buffers,
never uses freed plan objects and never hits undefined behaviour, on the tiny full-topology configurations of every network kind plus the
full-width C = 320 leaf modules (row-panel / fused-kernel planning and stream packing)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
SAN = os.path.join(HERE, "sanitize")


def _cfg_words(cconf):
    raw = bytes(cconf)
    return [int(x) for x in np.frombuffer(raw, dtype=np.int32)]


def _write_schema(path):
    from neurons_amd import _lib
    from neurons_amd.clip import clip_c_config, clip_state_dict_schema
    from neurons_amd.ops import NativeLeaf  # noqa: F401  (schema helpers live beside it)
    from neurons_amd.sgm import sgm_c_config, sgm_state_dict_schema
    from neurons_amd.unet3d import _motion_keys, _transformer_keys, make_c_config, state_dict_schema
    from neurons_amd.vae import vae_c_config, vae_decoder_state_dict_schema, vae_encoder_c_config, vae_encoder_state_dict_schema
    from tiny_configs import tiny_clip_config, tiny_ctrl_config, tiny_sgm_config, tiny_unet_config, tiny_vae_config
    nets = []
    ucfg, ccfg = tiny_unet_config(), tiny_ctrl_config()
    nets.append(("tiny_unet", make_c_config(ucfg, _lib.NR_KIND_UNET3D), state_dict_schema(ucfg, _lib.NR_KIND_UNET3D)))
    nets.append(("tiny_ctrl", make_c_config(ccfg, _lib.NR_KIND_SPARSECTRL), state_dict_schema(ccfg, _lib.NR_KIND_SPARSECTRL)))
    for name, kind, keys, width in (("leaf_temporal", _lib.NR_KIND_LEAF_TEMPORAL, _motion_keys("m", 320, 2), 320),
                                    ("leaf_transformer", _lib.NR_KIND_LEAF_TRANSFORMER3D, _transformer_keys("m", 320, 768), 320),
                                    ("leaf_transformer640", _lib.NR_KIND_LEAF_TRANSFORMER3D, _transformer_keys("m", 640, 768), 640),
                                    ("leaf_temporal640", _lib.NR_KIND_LEAF_TEMPORAL, _motion_keys("m", 640, 2), 640)):
        c = _lib.NrNetConfig()
        c.kind = kind
        c.in_channels = c.out_channels = width
        c.num_levels = 1
        c.block_out_channels[0] = width
        c.num_heads, c.cross_attention_dim, c.norm_num_groups, c.norm_eps = 8, 768, 32, 1e-5
        c.use_motion_module, c.motion_num_heads, c.motion_num_attention_blocks, c.motion_pe_max_len = 1, 8, 2, 24
        nets.append((name, c, keys))
    scfg, vcfg, tcfg = tiny_sgm_config(), tiny_vae_config(), tiny_clip_config()
    nets.append(("tiny_sgm", sgm_c_config(scfg), sgm_state_dict_schema(scfg)))
    nets.append(("tiny_vae_dec", vae_c_config(vcfg), vae_decoder_state_dict_schema(vcfg)))
    nets.append(("tiny_vae_enc", vae_encoder_c_config(vcfg), vae_encoder_state_dict_schema(vcfg)))
    nets.append(("tiny_clip", clip_c_config(tcfg), clip_state_dict_schema(tcfg)))
    assert C.sizeof(_lib.NrNetConfig) % 4 == 0
    with open(path, "w") as f:
        for name, cconf, schema in nets:
            f.write("N " + name + " " + " ".join(str(w) for w in _cfg_words(cconf)) + "\n")
            for k, shape in schema.items():
                f.write(f"T {k} {len(shape)} " + " ".join(str(int(d)) for d in shape) + "\n")


def test_planner_and_c_abi_host_logic_under_asan_ubsan(tmp_path):
    clang = "/opt/rocm/lib/llvm/bin/clang++"
    if not os.path.exists(clang):
        pytest.skip("ROCm clang not present")
    out = str(tmp_path / "build")
    r = subprocess.run(["make", "-C", SAN, "-j4", f"OUT={out}"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    schema = str(tmp_path / "schema.txt")
    _write_schema(schema)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
               LSAN_OPTIONS="suppressions=" + os.path.join(SAN, "lsan.supp"))
    env.pop("NR_DETERMINISTIC_BATCH", None)
    r = subprocess.run([os.path.join(out, "planner_dryrun"), schema], capture_output=True, text=True, timeout=900, env=env)
    print(r.stdout[-2000:])
    assert r.returncode == 0, r.stdout[-2000:] + "\n" + r.stderr[-6000:]
    assert "planner dry-run OK" in r.stdout
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-6000:]
