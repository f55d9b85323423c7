"""CPU-only tests: host logic of the drop-in layer, the C-ABI surface, the state-dict schema, the scheduler."""
import ctypes
import math
import os
import re
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from neurons_amd import _lib  # noqa: E402
from neurons_amd.scheduler import DDIMScheduler  # noqa: E402
from neurons_amd.unet3d import UNet3DConfig, random_state_dict, state_dict_schema  # noqa: E402


def test_library_exports_every_declared_symbol():
    """The C-ABI shared library loads (no GPU needed) and exports every function include/neurons_amd.h declares."""
    hdr = open(os.path.join(ROOT, "include", "neurons_amd.h")).read()
    declared = set(re.findall(r"\b(nr_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"nr_status", "nr_stream"}
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"libneurons_amd.so does not export {name}"
    assert declared == set(_lib.SYMBOLS), (declared ^ set(_lib.SYMBOLS))


def test_panel_kernel_rule_serves_the_shapes_it_was_measured_on():
    """Host logic of the register-panel form of lin160.hip (no GPU call): the LayerNorm-folded wide projections (GEGLU N = 8 C, q|k|v N = 3 C; K = C) are routed
    to it at C = 640 from 4096 rows of one clip and at C = 1280 from 2048 rows -- the headline's 16 x 16 and 8 x 8 levels, configs 4 / 5, SparseCtrl's grouped rows --
    and NOT at the sgm keyframe model's 2048-row C = 640 level (measured slower there, profiles/r06_lin160_panel_ab.txt), on narrow projections (N < 3 K), at
    other widths, or on the <= 1024-row shapes of the GEGLU variant."""
    lib = ctypes.CDLL(_lib.LIB_PATH)
    rule = lib.nr_lin160_panel_rule
    rule.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
    rule.restype = ctypes.c_int
    if os.environ.get("NR_LIN160") == "0" or os.environ.get("NR_LIN160_PANEL") == "0":
        pytest.skip("panel kernel switched off by the environment")
    for M, N, K in ((8192, 5120, 640), (8192, 1920, 640), (65536, 5120, 640), (262144, 1920, 640), (40960, 5120, 640), (4096, 5120, 640),
                    (2048, 10240, 1280), (2048, 3840, 1280), (16384, 10240, 1280), (10240, 3840, 1280)):
        assert rule(M, N, K) == 1, (M, N, K)
    for M, N, K in ((2048, 5120, 640), (2048, 1920, 640), (512, 10240, 1280), (1024, 10240, 1280), (8192, 640, 640), (2048, 1280, 1280), (8192, 1280, 640),
                    (32768, 2560, 320), (8192, 5100, 640), (2048, 10250, 1280)):
        assert rule(M, N, K) == 0, (M, N, K)


def test_ctypes_structs_mirror_the_header():
    """nr_net_config / nr_profile in include/neurons_amd.h and their ctypes mirrors must agree field by field (name, order,
    array length); a drift would silently shift every later field across the ABI."""
    import ctypes as C
    import re
    hdr = open(os.path.join(ROOT, "include", "neurons_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    body = re.search(r"typedef struct nr_net_config \{(.*?)\} nr_net_config;", hdr, flags=re.S).group(1)
    fields = re.findall(r"(int32_t|float)\s+(\w+)(?:\[(\w+)\])?\s*;", body)
    assert len(fields) == len(_lib.NrNetConfig._fields_) >= 22
    for (ctype, name, arr), (pname, ptype) in zip(fields, _lib.NrNetConfig._fields_):
        assert name == pname, (name, pname)
        base = C.c_float if ctype == "float" else C.c_int32
        if arr:
            n = _lib.NR_MAX_LEVELS if arr == "NR_MAX_LEVELS" else int(arr)
            assert ptype._length_ == n and ptype._type_ is base, name
        else:
            assert ptype is base, name
    for const in ("NR_KIND_UNET3D", "NR_KIND_SPARSECTRL", "NR_KIND_SGM_UNET", "NR_KIND_VAE_DECODER", "NR_KIND_VAE_ENCODER", "NR_KIND_CLIP_TEXT",
                  "NR_MAX_LEVELS"):
        assert int(re.search(rf"#define {const} (\d+)", hdr).group(1)) == getattr(_lib, const), const


def test_create_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from neurons_amd import NativeUNet3D
    net = NativeUNet3D(UNet3DConfig(block_out_channels=(64, 64, 128, 128), cross_attention_dim=64))
    with pytest.raises(RuntimeError, match="no CPU fallback|MI355X|HIP"):
        net.to("cpu")
    with pytest.raises(RuntimeError):
        net(torch.zeros(2, 4, 8, 8, 8), 1, torch.zeros(2, 77, 64))


def test_schema_matches_reference_parameter_counts():
    """1 206 tensors / 1 276.7 M parameters (U-Net) and 496.7 M (SparseCtrl): SURVEY.md F4, §7."""
    from neurons_amd.sparsectrl import controlnet_config_from_unet
    u = state_dict_schema(UNet3DConfig(), _lib.NR_KIND_UNET3D)
    assert len(u) == 1206
    assert abs(sum(int(np.prod(s)) for s in u.values()) / 1e6 - 1276.66) < 0.05
    ccfg = controlnet_config_from_unet(UNet3DConfig(), dict(set_noisy_sample_input_to_zero=True,
                                                            use_simplified_condition_embedding=True, conditioning_channels=4,
                                                            motion_module_kwargs=dict(attention_block_types=["Temporal_Self"],
                                                                                      temporal_position_encoding_max_len=32)))
    c = state_dict_schema(ccfg, _lib.NR_KIND_SPARSECTRL)
    assert abs(sum(int(np.prod(s)) for s in c.values()) / 1e6 - 496.73) < 0.05
    assert "controlnet_down_blocks.11.weight" in c and "controlnet_mid_block.bias" in c and "up_blocks.0.resnets.0.conv1.weight" not in c


def test_load_state_dict_errors_mirror_torch():
    from neurons_amd import NativeUNet3D
    cfg = UNet3DConfig(block_out_channels=(64, 64, 128, 128), cross_attention_dim=64)
    net = NativeUNet3D(cfg)
    sd = random_state_dict(cfg, seed=1)
    bad = dict(sd)
    bad.pop("conv_in.weight")
    with pytest.raises(RuntimeError, match="Missing key"):
        net.load_state_dict(bad)
    missing, unexpected = net.load_state_dict(bad, strict=False)
    assert missing == ["conv_in.weight"] and unexpected == []
    bad = dict(sd)
    bad["conv_in.weight"] = torch.zeros(3, 3)
    with pytest.raises(RuntimeError, match="size mismatch"):
        net.load_state_dict(bad)
    extra = dict(sd)
    extra["down_blocks.0.motion_modules.0.temporal_transformer.transformer_blocks.0.attention_blocks.0.pos_encoder.pe"] = torch.zeros(1, 24, 64)
    net.load_state_dict(extra)      # non-persistent PE buffer is accepted and ignored (animatediff/utils/util.py:116)


def test_unsupported_configs_are_rejected():
    from neurons_amd import NativeUNet3D
    with pytest.raises(ValueError, match="does not exist"):
        NativeUNet3D(UNet3DConfig(down_block_types=("Foo", "DownBlock3D", "DownBlock3D", "DownBlock3D")))
    with pytest.raises(NotImplementedError):
        NativeUNet3D(UNet3DConfig(use_inflated_groupnorm=False))


# ---- scheduler (diffusers 0.11.1 DDIM restated; cross-checked against in-repo siblings) -----------------
def _sched():
    return DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="linear", steps_offset=1, clip_sample=False)


def test_ddim_timesteps_match_survey_a2():
    s = _sched()
    s.set_timesteps(25)
    assert s.timesteps_host[0] == 961 and s.timesteps_host[-1] == 1 and len(s.timesteps_host) == 25
    s.set_timesteps(50)
    assert s.timesteps_host[0] == 981 and s.timesteps_host[1] == 961 and s.timesteps_host[-1] == 1
    assert s.init_noise_sigma == 1.0 and s.order == 1
    x = torch.randn(2, 3)
    assert s.scale_model_input(x, 5) is x


def test_alphas_match_cumprod_mechanics_of_in_repo_discretizer():
    """generative_models/sgm/modules/diffusionmodules/discretizer.py:51-55 builds alphas_cumprod as
    cumprod(1 - betas); 'linear' here is linspace in beta, not in sqrt(beta)."""
    s = _sched()
    betas = np.linspace(0.00085, 0.012, 1000, dtype=np.float64)
    ac = np.cumprod(1.0 - betas)
    assert np.allclose(s.alphas_cumprod.numpy(), ac, rtol=2e-5)
    s.set_timesteps(50)
    a_t, a_prev = s.alpha_pair(981)
    assert math.isclose(a_t, ac[981], rel_tol=2e-5) and math.isclose(a_prev, ac[961], rel_tol=2e-5)
    a_t, a_prev = s.alpha_pair(1)
    assert a_prev == 1.0      # set_alpha_to_one: prev timestep < 0 -> final_alpha_cumprod = 1


def test_ddim_update_is_inverse_of_in_repo_next_step():
    """animatediff/utils/util.py:211-221 (``next_step``, DDIM *inversion*) uses the same closed form in the other
    direction: stepping t -> t' with eps held fixed and back must return the start (exactly, up to fp32)."""
    from oracle import animatediff_oracle as O
    ac = O.ddim_alphas_cumprod()
    x = torch.randn(1, 4, 2, 4, 4, generator=torch.Generator().manual_seed(0))
    eps = torch.randn(1, 4, 2, 4, 4, generator=torch.Generator().manual_seed(1))
    t, n = 501, 50
    prev = O.ddim_step(eps, t, x, ac, n)              # t -> t - 20
    # next_step formula (util.py:217-220) from timestep t-20 back to t
    a_t, a_next = ac[t - 20], ac[t]
    x0 = (prev - (1 - a_t) ** 0.5 * eps) / a_t ** 0.5
    back = a_next ** 0.5 * x0 + (1 - a_next) ** 0.5 * eps
    assert torch.allclose(back, x, atol=2e-5)


def test_add_noise_and_low_strength_quirk():
    s = _sched()
    s.set_timesteps(10)
    x, n = torch.ones(1, 4, 2, 2, 2), torch.full((1, 4, 2, 2, 2), 2.0)
    t = s.timesteps[:1]
    y = s.add_noise(x, n, t)
    a = s.alphas_cumprod[int(t)]
    assert torch.allclose(y, a.sqrt() * x + (1 - a).sqrt() * n)
    # low_strength >= 1 -> empty latent_timestep (pipeline_neuroclips.py:410-413)
    init_timestep = min(int(10 * 1.0), 10)
    t_start = max(10 - init_timestep, 0)
    assert s.timesteps[:t_start][:1].numel() == 0


def test_scheduler_step_refuses_cpu():
    s = _sched()
    s.set_timesteps(10)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        s.step(torch.zeros(1, 4), 901, torch.zeros(1, 4))
    with pytest.raises(NotImplementedError):
        s.step(torch.zeros(1, 4), 901, torch.zeros(1, 4), eta=0.5)


def test_pipeline_requires_gpu_and_validates_inputs():
    from neurons_amd import NativeUNet3D, NeuroclipsPipeline
    cfg = UNet3DConfig(sample_size=8, block_out_channels=(64, 64, 128, 128), cross_attention_dim=64)
    pipe = NeuroclipsPipeline(None, None, None, NativeUNet3D(cfg), _sched(), None)
    with pytest.raises(ValueError, match="divisible by 8"):
        pipe("", video_length=8, height=60, width=64)
    with pytest.raises(ValueError, match="callback_steps"):
        pipe("", video_length=8, height=64, width=64, callback_steps=0)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pipe("", video_length=8, height=64, width=64, text_embeddings=torch.zeros(2, 77, 64))


def test_synth_is_deterministic_and_normal():
    from neurons_amd.synth import randn
    a, b = randn("x.y", (257, 3), 5), randn("x.y", (257, 3), 5)
    assert torch.equal(a, b) and not torch.equal(a, randn("x.z", (257, 3), 5))
    z = randn("big", (200000,), 1)
    assert abs(z.mean().item()) < 0.01 and abs(z.std().item() - 1) < 0.01


def test_controlnet_group_plan_invariants():
    """Grouped SparseCtrl schedule (pipeline.controlnet_group_plan): every step consumes an evaluation that was issued earlier, at most
    one group ahead, into a slot no pending group still owns; group timesteps are the steps' timesteps (tail repeated)."""
    from neurons_amd.pipeline import controlnet_group_plan
    for n in (1, 2, 3, 7, 10, 25, 38, 50):
        ts = [1000 - 13 * i for i in range(n)]
        for G in (1, 2, 3, 4, 8, 64):
            groups, steps = controlnet_group_plan(ts, G)
            g_eff = max(1, min(G, n))
            assert len(steps) == n and len(groups) == (n + g_eff - 1) // g_eff
            issued = {0}                                  # group 0 goes out before the loop
            live_slots = {groups[0]["slot"]: 0}
            for i, (launch, g, p) in enumerate(steps):
                assert (g, p) == divmod(i, g_eff)
                if launch is not None:
                    assert launch == g + 1 and p == 0 and launch not in issued
                    # its slot was last read by group g - 1 (finished with step i - 1), never by the group being consumed now
                    assert groups[launch]["slot"] != groups[g]["slot"]
                    assert live_slots.get(groups[launch]["slot"], launch - 2) == launch - 2 or launch - 2 < 0
                    live_slots[groups[launch]["slot"]] = launch
                    issued.add(launch)
                assert g in issued and live_slots[groups[g]["slot"]] == g
                assert groups[g]["timesteps"][p] == ts[i]
            assert issued == set(range(len(groups)))
            for g, grp in enumerate(groups):
                assert len(grp["timesteps"]) == g_eff and grp["slot"] == g % 2
                assert grp["timesteps"] == [ts[min(g * g_eff + p, n - 1)] for p in range(g_eff)]


def test_controlnet_group_size_rule():
    """pipeline.controlnet_group_size: "auto" avoids the wasted tail (50 / 25 / 10 steps -> 5), respects the 64-sample and 2 Mi-row caps,
    never exceeds the step count; an integer is taken as is (capped)."""
    from neurons_amd.pipeline import controlnet_group_size as gs
    assert gs(50, 2, 16, 32, 32) == 5 and gs(25, 2, 16, 32, 32) == 5 and gs(10, 2, 8, 8, 8) == 5
    assert gs(6, 2, 16, 32, 32) == 3 and gs(6, 16, 16, 32, 32) == 3          # B = 1 and B = 8 (C4): two full groups of 3
    assert gs(50, 16, 16, 32, 32) == 4                                       # 64 // 16: the sample cap
    assert gs(50, 8, 32, 64, 64) == 2                                        # BASELINE config 5: the 2 Mi-row cap
    assert gs(1, 2, 16, 32, 32) == 1 and gs(3, 2, 16, 32, 32) == 3
    for forced in (1, 2, 3, 4, 8):
        assert gs(50, 2, 16, 32, 32, forced) == forced
    assert gs(50, 16, 16, 32, 32, 8) == 4 and gs(2, 2, 16, 32, 32, 4) == 2
    for n in range(1, 60):
        g = gs(n, 2, 16, 32, 32)
        assert 1 <= g <= min(5, n)


def test_sgm_sampler_recognises_only_the_canonical_denoiser_closure():
    """the fused HIP path of EulerEDMSampler is taken for utils.unclip_recon's EXACT closure (utils.py:337-338), for the
    explicit opt-in ``engine.native_denoiser()`` and for native objects passed directly; every variation of the closure (post-processing,
    changed sigma / cond, extra inputs, bound methods, partials, closures that merely reference an engine) is called through the generic
    loop, i.e. gets the reference's ``sampler(denoiser, ...)`` contract."""
    import functools
    from neurons_amd import sgm
    from tiny_configs import tiny_sgm_config, tiny_vae_config

    def build():                       # the engine must be a closure variable, as `diffusion_engine` is inside utils.unclip_recon
        diffusion_engine = sgm.NativeDiffusionEngine(tiny_sgm_config(), tiny_vae_config(), num_steps=4)

        def denoiser(x, sigma, c):
            return diffusion_engine.denoiser(diffusion_engine.model, x, sigma, c)

        def scaled(x, sigma, c):
            return diffusion_engine.denoiser(diffusion_engine.model, x, sigma, c) * 1.0

        def other_sigma(x, sigma, c):
            return diffusion_engine.denoiser(diffusion_engine.model, x, sigma * 1, c)

        def extra_inputs(x, sigma, c):
            return diffusion_engine.denoiser(diffusion_engine.model, x, sigma, c, foo=1)

        def swapped_cond(x, sigma, c):
            return diffusion_engine.denoiser(diffusion_engine.model, x, sigma, dict(c))

        def default_arg(x, sigma, c=None):
            return diffusion_engine.denoiser(diffusion_engine.model, x, sigma, c)

        return diffusion_engine, denoiser, [scaled, other_sigma, extra_inputs, swapped_cond, default_arg]

    eng, canonical, variants = build()
    net = eng.model.diffusion_model
    rec = eng.sampler._native_network
    assert rec(canonical) is net
    assert rec(eng.native_denoiser()) is net
    assert rec(eng) is net and rec(eng.model) is net and rec(net) is net
    for v in variants:
        assert rec(v) is None, v.__name__
    assert rec(functools.partial(canonical)) is None
    assert rec(eng.decode_first_stage) is None                 # a bound method of the engine
    assert rec(lambda x, sigma, c: eng.denoiser(eng.model, x, sigma, c) + 0) is None
    other = sgm.NativeDiffusionEngine(tiny_sgm_config(), tiny_vae_config(), num_steps=4)
    assert rec(other.native_denoiser()) is other.model.diffusion_model        # a tagged denoiser names its own engine
    eng.denoiser = lambda *a: None                               # a swapped-out denoiser is no longer the native one
    assert rec(canonical) is None


def _disassemble_shipped_library():
    """ISA text of every gfx950 code object inside libneurons_amd.so (llvm-objcopy / llvm-objdump of the ROCm image; no GPU)."""
    import shutil
    import struct
    import subprocess
    import tempfile
    from neurons_amd import _lib
    llvm = "/opt/rocm/lib/llvm/bin"
    objcopy, objdump = os.path.join(llvm, "llvm-objcopy"), os.path.join(llvm, "llvm-objdump")
    if not (os.path.exists(objcopy) and os.path.exists(objdump)):
        pytest.skip("ROCm llvm tools not present")
    tmp = tempfile.mkdtemp(prefix="nr_isa_")
    out = []
    try:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.run([objcopy, "--dump-section", f".hip_fatbin={fat}", _lib.LIB_PATH, os.path.join(tmp, "copy.so")], check=True)
        d = open(fat, "rb").read()
        magic, pos = b"__CLANG_OFFLOAD_BUNDLE__", 0
        while True:
            i = d.find(magic, pos)
            if i < 0:
                break
            nb, o = struct.unpack_from("<Q", d, i + 24)[0], i + 32
            for _ in range(nb):
                off, size, tl = struct.unpack_from("<QQQ", d, o)
                o += 24
                triple = d[o:o + tl].decode()
                o += tl
                if "gfx950" in triple and size:
                    co = os.path.join(tmp, f"co{len(out)}.o")
                    open(co, "wb").write(d[i + off:i + off + size])
                    out.append(subprocess.run([objdump, "-d", "--mcpu=gfx950", co], check=True, capture_output=True, text=True).stdout)
            pos = i + 24
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


def test_shipped_library_contains_no_packed_fp32_valu_ops():
    """The two-stream corruption of round 2 was bisected to the packed-fp32 code generation of norm.o (profiles/r03_race_*) and
    is NOT root-caused; the product's protection is the library-wide `-packed-fp32-ops` target feature (csrc/Makefile).  A toolchain or
    flag change that re-introduces v_pk_{add,mul,fma}_f32 would give no compile-time signal, so disassemble every gfx950 code object of
    the shipped .so and look.  Needs llvm-objcopy / llvm-objdump of the ROCm image, no GPU."""
    from neurons_amd import _lib
    asms = _disassemble_shipped_library()
    mfma = sum(a.count("v_mfma_") for a in asms)
    packed = sum(a.count(op) for a in asms for op in ("v_pk_add_f32", "v_pk_mul_f32", "v_pk_fma_f32"))
    assert len(asms) >= 8 and mfma > 1000, (len(asms), mfma)            # the disassembly really is the library's kernels
    assert packed == 0, f"{packed} packed fp32 VALU instructions in {_lib.LIB_PATH}: was the -packed-fp32-ops flag dropped?"


def test_mfma_overlap_scanner_flags_the_miscompiled_chain_and_nothing_else():
    """tools/check_mfma_overlap.py on the exact instruction pair hipcc 7.2 emitted for attention.hip's ONES instantiation (wrong sums on
    MI355X) and on the harmless look-alike (partial overlap with a VALU-zeroed SrcC, ffpanel.hip)."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from check_mfma_overlap import scan
    bad = """
	v_mfma_f32_16x16x32_bf16 v[0:3], v[86:89], v[18:21], v[2:5]
	v_mfma_f32_16x16x32_bf16 v[10:13], v[82:85], v[22:25], v[10:13]
	v_mfma_f32_16x16x32_bf16 v[2:5], v[90:93], v[22:25], v[0:3]     // 000000001234: D3B50002
"""
    hits = scan(bad)
    assert len(hits) == 1 and hits[0][1].startswith("v_mfma_f32_16x16x32_bf16 v[2:5]"), hits
    ok = """
	v_mov_b32_e32 v46, 0
	v_mov_b32_e32 v47, 0
	v_mov_b32_e32 v48, 0
	v_mov_b32_e32 v49, 0
	v_mfma_f32_16x16x32_bf16 v[48:51], v[0:3], v[112:115], v[46:49]
	v_mfma_f32_16x16x32_bf16 v[48:51], v[0:3], v[112:115], v[48:51]
	v_mfma_f32_16x16x32_bf16 v[14:17], v[14:17], v[58:61], v[24:27]
"""
    assert scan(ok) == []


def test_shipped_library_has_no_partially_overlapping_mfma_accumulator_chain():
    """The miscompile of DESIGN 3e gives no compile-time signal either: scan the shipped ISA for MFMA -> MFMA chains whose destination
    partially overlaps the SrcC the previous MFMA produced."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from check_mfma_overlap import scan
    asms = _disassemble_shipped_library()
    assert len(asms) >= 8
    hits = [h for a in asms for h in scan(a)]
    assert not hits, hits[:5]


def test_native_networks_accept_every_call_the_reference_script_makes_before_the_pipeline():
    """scripts/neuroclips_video.py touches the two network objects between construction and NeuroclipsPipeline(...):
    :122-123 `unet.config.<attr> = ...`, :125 `SparseControlNetModel.from_unet(unet, controlnet_additional_kwargs=...)`,
    :137-138 `controlnet.load_state_dict(sd)` / `.to(device)`, :212-214 `enable_xformers_memory_efficient_attention()` on both when
    xformers imports (attention.py:228-254), plus the diffusers ModelMixin switches defined next to it (unet.py:251-318
    `set_attention_slice`, `_set_gradient_checkpointing`).  The memory switches are documented no-ops here (one attention path, no autograd):
    they must exist, return None and leave the object usable — an AttributeError on a box that has xformers would break "called unchanged"."""
    from neurons_amd import NativeSparseCtrl, NativeUNet3D
    unet = NativeUNet3D(UNet3DConfig(block_out_channels=(64, 64, 128, 128), cross_attention_dim=64))
    unet.config.num_attention_heads = 8                                 # :122
    unet.config.projection_class_embeddings_input_dim = None            # :123
    ctrl = NativeSparseCtrl.from_unet(unet, controlnet_additional_kwargs=dict(
        set_noisy_sample_input_to_zero=True, use_simplified_condition_embedding=True, conditioning_channels=4))
    assert ctrl.use_simplified_condition_embedding is True              # read at :278
    for net in (unet, ctrl):
        before = dict(vars(net))
        assert net.enable_xformers_memory_efficient_attention() is None                 # :213-214
        assert net.enable_xformers_memory_efficient_attention(attention_op=None) is None
        assert net.disable_xformers_memory_efficient_attention() is None
        assert net.set_use_memory_efficient_attention_xformers(True) is None
        for s in ("auto", "max", 4, [4, 4], None):
            assert net.set_attention_slice(s) is None                                   # unet.py:251
        with pytest.raises(ValueError):
            net.set_attention_slice("half")
        assert net.enable_gradient_checkpointing() is None and net.disable_gradient_checkpointing() is None
        assert net._set_gradient_checkpointing(None, value=True) is None                # unet.py:316
        assert net.eval() is net and net.requires_grad_(False) is net and net.train(False) is net
        with pytest.raises(RuntimeError):
            net.train()
        assert dict(vars(net)) == before                                                 # nothing changed
