"""Single-kernel entry points of libneurons_amd.so on torch tensors (device memory + streams only).

Used by tests/ to check each HIP kernel against the oracle; the product path is the whole-network
engine (neurons_amd/unet3d.py, sparsectrl.py), which calls the same kernels from C++.
All activations are channels-last bf16 ``[nimg, H, W, C]`` / token-major ``[M, C]``.
"""
import torch

from . import _lib


def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _chk_bf16(*ts):
    for t in ts:
        if t is not None:
            assert t.is_cuda and t.dtype == torch.bfloat16 and t.is_contiguous(), (t.dtype, t.shape)


def _chk_f32(*ts):
    for t in ts:
        if t is not None:
            assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous(), (t.dtype, t.shape)


def gemm(a, w, bias=None, res=None, geglu=False):
    """out[M, N(or N/2)] = a[M, K] @ w[N, K]^T (+bias) (+res) ; geglu expects the value/gate-interleaved weight."""
    _chk_bf16(a, w, res)
    _chk_f32(bias)
    M, K = a.shape
    N = w.shape[0]
    out = torch.empty(M, N // 2 if geglu else N, dtype=torch.bfloat16, device=a.device)
    lib = _lib.load()
    _lib.check(lib.nr_op_gemm(_stream(), _ptr(a), K, _ptr(w), _ptr(bias), _ptr(res), out.shape[1], _ptr(out),
                              out.shape[1], M, N, K, 1 if geglu else 0))
    return out


def gemm2(a0, a1, w, bias=None, res=None):
    """out[M, N] = cat([a0, a1], dim=1) @ w[N, c0 + c1]^T (+bias) (+res) without materialising the concat (engine: two-source igemm)."""
    _chk_bf16(a0, a1, w, res)
    _chk_f32(bias)
    M, c0 = a0.shape
    c1 = a1.shape[1]
    N = w.shape[0]
    out = torch.empty(M, N, dtype=torch.bfloat16, device=a0.device)
    _lib.check(_lib.load().nr_op_gemm2(_stream(), _ptr(a0), c0, c0, _ptr(a1), c1, c1, _ptr(w), _ptr(bias), _ptr(res), N, _ptr(out), N, M, N))
    return out


def g8p_mode(mode):
    """0: never use the 256-row ping-pong kernel (gemm8p.hip), 1: shipped heuristic, 2: whenever the shape is supported (tests, A/B)."""
    _lib.load().nr_g8p_set_mode(int(mode))


def g8p_phases(n):
    """2 (default where instantiated) or 4 phases per k-tile of the ping-pong kernel (A/B)."""
    _lib.load().nr_g8p_set_phases(int(n))


def ff_waves(n):
    """8 (default) or 4 waves per workgroup of the fused FeedForward kernel (A/B; bit-identical results)."""
    _lib.load().nr_ff_set_waves(int(n))


def ln_gemm(a, w, gamma, beta, bias=None, res=None, eps=1e-5, act=0, geglu=False):
    """out = Linear(LayerNorm(a)) with the LayerNorm folded into the GEMM (engine: ln_linear).  The folding of gamma / beta
    into (w_scaled, ln_c, bias_folded) is done here on the host exactly as engine.hip's w_ln_linear does.  ``geglu``: w is a GEGLU projection
    [value(inner) | gate(inner)] (rows re-ordered here to the engine's 16 / 16 interleave); out = value * gelu(gate), inner columns."""
    _chk_bf16(a, res)
    M, K = a.shape
    if geglu:
        w, bias = geglu_permute(w, bias)
    N = w.shape[0]
    wf = w.float()
    ws = (wf * gamma.float()[None]).to(torch.bfloat16).contiguous()
    c = ws.float().sum(dim=1).contiguous()
    b = (wf.double() @ beta.double()).float()
    if bias is not None:
        b = b + bias.float()
    b = b.contiguous()
    No = N // 2 if geglu else N
    out = torch.empty(M, No, dtype=torch.bfloat16, device=a.device)
    lib = _lib.load()
    _lib.check(lib.nr_op_ln_gemm(_stream(), _ptr(a), K, _ptr(ws), _ptr(c), _ptr(b), float(eps), _ptr(res), No, _ptr(out), No, M, N, K, 1 if geglu else 0,
                                 int(act)))
    return out


def gemm_ex(a, w, bias=None, ln=None, eps=1e-5, rowvec=None, rowvec_div=1, rowvec_mod=0, res=None, act=0, out_scale=1.0):
    """out = (Linear([LayerNorm](a)) + rowvec[(m // rowvec_div) % rowvec_mod]) * out_scale [quick_gelu] + res.  ``ln`` = (gamma, beta)
    folds the LayerNorm into the weights on the host exactly as engine.hip's w_ln_linear does; ``w`` is fp32 or bf16 [N, K]."""
    _chk_bf16(a, res)
    _chk_f32(rowvec)
    M, K = a.shape
    N = w.shape[0]
    wf = w.float()
    ln_c = None
    b = None if bias is None else bias.float().contiguous()
    if ln is not None:
        gamma, beta = ln
        ws = (wf * gamma.float()[None]).to(torch.bfloat16).contiguous()
        ln_c = ws.float().sum(dim=1).contiguous()
        b = (wf.double() @ beta.double()).float()
        if bias is not None:
            b = b + bias.float()
        b = b.contiguous()
    else:
        ws = w.to(torch.bfloat16).contiguous()
    out = torch.empty(M, N, dtype=torch.bfloat16, device=a.device)
    lib = _lib.load()
    _lib.check(lib.nr_op_gemm_ex(_stream(), _ptr(a), K, _ptr(ws), _ptr(b), _ptr(ln_c), float(eps), _ptr(rowvec), int(rowvec_div),
                                 int(rowvec_mod), 0 if rowvec is None else rowvec.shape[1], _ptr(res), N, _ptr(out), N, M, N, K, 0, int(act),
                                 float(out_scale)))
    return out


def geglu_permute(w, b):
    """Reorder a GEGLU projection (rows [value(inner) | gate(inner)]) into 16-value/16-gate interleave."""
    inner = w.shape[0] // 2
    n = torch.arange(2 * inner)
    q, j = n // 32, n % 32
    src = torch.where(j < 16, q * 16 + j, inner + q * 16 + (j - 16))
    return w[src].contiguous(), (None if b is None else b[src].contiguous())


def conv3x3(x0, w, bias=None, x1=None, stride=1, ups=False, rowvec=None, rowvec_div=1, res=None, tap_inner=False):
    """x0/x1: [nimg, H, W, C] bf16; w: [Cout, 3, 3, Cin] bf16 (tap-major); returns [nimg, OH, OW, Cout].
    tap_inner: run the engine's ResnetBlock form (stride 1, one source): the weight is re-arranged here to [Cout][Cin/64][3][3][64]."""
    _chk_bf16(x0, x1, w, res)
    _chk_f32(bias, rowvec)
    nimg, H, W, c0 = x0.shape
    if tap_inner:
        if x1 is not None or stride != 1 or ups or c0 % 64 != 0:
            raise ValueError("tap_inner: stride 1, no upsample, one source, Cin % 64 == 0")
        Cout = w.shape[0]
        wt = w.reshape(Cout, 9, c0 // 64, 64).permute(0, 2, 1, 3).contiguous()
        out = torch.empty(nimg, H, W, Cout, dtype=torch.bfloat16, device=x0.device)
        _lib.check(_lib.load().nr_op_conv3x3_tap_inner(_stream(), _ptr(x0), c0, nimg, H, W, _ptr(wt), _ptr(bias), _ptr(rowvec), rowvec_div,
                                                        _ptr(res), _ptr(out), Cout))
        return out
    c1 = 0 if x1 is None else x1.shape[3]
    Cout = w.shape[0]
    OH, OW = (2 * H, 2 * W) if ups else (H, W)
    if stride == 2:
        OH, OW = (OH - 1) // 2 + 1, (OW - 1) // 2 + 1
    out = torch.empty(nimg, OH, OW, Cout, dtype=torch.bfloat16, device=x0.device)
    lib = _lib.load()
    _lib.check(lib.nr_op_conv3x3(_stream(), _ptr(x0), c0, _ptr(x1), c1, nimg, H, W, stride, 1 if ups else 0, _ptr(w),
                                 _ptr(bias), _ptr(rowvec), rowvec_div, _ptr(res), _ptr(out), Cout))
    return out


def groupnorm(x0, gamma, beta, groups=32, eps=1e-5, silu=False, x1=None):
    _chk_bf16(x0, x1)
    _chk_f32(gamma, beta)
    nimg, H, W, c0 = x0.shape
    c1 = 0 if x1 is None else x1.shape[3]
    out = torch.empty(nimg, H, W, c0 + c1, dtype=torch.bfloat16, device=x0.device)
    ws = torch.empty(nimg * (H * W) * groups * 2 + 64, dtype=torch.float32, device=x0.device)
    lib = _lib.load()
    _lib.check(lib.nr_op_groupnorm(_stream(), _ptr(x0), c0, _ptr(x1), c1, nimg, H * W, groups, _ptr(gamma), _ptr(beta),
                                   eps, 1 if silu else 0, _ptr(ws), _ptr(out)))
    return out


def layernorm(x, gamma, beta, eps=1e-5, pe=None, pe_hw=1, pe_F=1):
    _chk_bf16(x)
    _chk_f32(gamma, beta, pe)
    M, Cc = x.shape
    out = torch.empty_like(x)
    lib = _lib.load()
    _lib.check(lib.nr_op_layernorm(_stream(), _ptr(x), _ptr(out), M, Cc, _ptr(gamma), _ptr(beta), eps, _ptr(pe), pe_hw, pe_F))
    return out


def attention_self(qkv, heads, fp8=False):
    """qkv: [nimg, L, 3C] fused; returns [nimg, L, C].  fp8: OCP e4m3 MFMA operands (L >= 48; BASELINE config 5)."""
    _chk_bf16(qkv)
    nimg, L, C3 = qkv.shape
    Cc = C3 // 3
    out = torch.empty(nimg, L, Cc, dtype=torch.bfloat16, device=qkv.device)
    lib = _lib.load()
    _lib.check(lib.nr_op_attention(_stream(), 8 if fp8 else 0, _ptr(qkv), None, _ptr(out), nimg, L, L, Cc, heads, 1, 1))
    return out


def attention_cross(q, kv, heads, kv_div, fp8=False):
    """q: [nimg, L, C]; kv: [nb, Lk, 2C] fused; image n attends to kv[n // kv_div].  fp8 as in attention_self."""
    _chk_bf16(q, kv)
    nimg, L, Cc = q.shape
    Lk = kv.shape[1]
    out = torch.empty_like(q)
    lib = _lib.load()
    _lib.check(lib.nr_op_attention(_stream(), 9 if fp8 else 1, _ptr(q), _ptr(kv), _ptr(out), nimg, L, Lk, Cc, heads, 1, kv_div))
    return out


def attention_temporal(qkv, heads, frames):
    """qkv: [(b f), hw, 3C] fused; attention runs over f for every (b, pixel); returns [(b f), hw, C]."""
    _chk_bf16(qkv)
    nimg, hw, C3 = qkv.shape
    Cc = C3 // 3
    out = torch.empty(nimg, hw, Cc, dtype=torch.bfloat16, device=qkv.device)
    lib = _lib.load()
    _lib.check(lib.nr_op_attention(_stream(), 2, _ptr(qkv), None, _ptr(out), nimg, hw, frames, Cc, heads, frames, 1))
    return out


def cfg_ddim_step(eps, x, guidance_scale, alpha_prod_t, alpha_prod_t_prev, do_cfg=True):
    _chk_f32(eps, x)
    out = torch.empty_like(x)
    lib = _lib.load()
    _lib.check(lib.nr_cfg_ddim_step(_stream(), _ptr(eps), _ptr(x), _ptr(out), x.numel(), float(guidance_scale),
                                    1 if do_cfg else 0, float(alpha_prod_t), float(alpha_prod_t_prev)))
    return out


def cfg_combine(eps, guidance_scale):
    """eps fp32 [2B, ...] (uncond half first) -> [B, ...] = e_u + g (e_t - e_u)  (pipeline_neuroclips.py:478-480), HIP nr_cfg_combine"""
    _chk_f32(eps)
    if eps.shape[0] % 2:
        raise ValueError("cfg_combine: the batch must hold the unconditional and the text half")
    out = torch.empty((eps.shape[0] // 2,) + tuple(eps.shape[1:]), dtype=torch.float32, device=eps.device)
    _lib.check(_lib.load().nr_cfg_combine(_stream(), _ptr(eps), _ptr(out), out.numel(), float(guidance_scale)))
    return out


def ff_fused(t, x, gamma, beta, w1, b1, w2, b2, wpo, bpo, eps=1e-5, reuse_stream=False):
    """out = x + proj_out(t + FF(t)),  FF(t) = net.2(GEGLU(net.0(LayerNorm(t))))  — the tail of a (temporal) transformer at C = 320 in ONE
    launch (ffpanel.hip).  The weight conversion is done here on the host exactly as engine.hip does it: value/gate interleave of net.0
    (w_geglu / b_geglu), net.2 + proj_out folded into Wc = [Wpo | Wpo Wff2], bc = bpo + Wpo bff2 (w_fold_ff_proj).
    t, x: [M, C] bf16; w1 [8C, C], b1 [8C], w2 [C, 4C], b2 [C], wpo [C, C], bpo [C] fp32."""
    _chk_bf16(t, x)
    M, C = t.shape
    w1p, b1p = geglu_permute(w1.to(torch.bfloat16), b1.float())
    wc = torch.cat([wpo.float(), wpo.float() @ w2.float()], dim=1).to(torch.bfloat16).contiguous()
    bc = (bpo.float() + wpo.float() @ b2.float()).contiguous()
    out = torch.empty(M, C, dtype=torch.bfloat16, device=t.device)
    _lib.check(_lib.load().nr_op_ff_fused(_stream(), _ptr(t), _ptr(x), _ptr(out), M, C, None if reuse_stream else _ptr(w1p), _ptr(gamma.float().contiguous()),
                                          _ptr(beta.float().contiguous()), _ptr(b1p), _ptr(wc), _ptr(bc), float(eps)))
    return out


def temporal_pe_table(frames, C, device):
    """PositionalEncoding table (motion_module.py:225-239): pe[pos, 0::2] = sin(pos * div), pe[pos, 1::2] = cos(pos * div)."""
    import math
    position = torch.arange(frames, device=device, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, C, 2, device=device, dtype=torch.float32) * (-math.log(10000.0) / C))
    pe = torch.zeros(frames, C, device=device)
    pe[:, 0::2] = torch.sin(position * div)
    pe[:, 1::2] = torch.cos(position * div)
    return pe


def tattn_fused(t, nbatch, hw, gamma, beta, wq, wk, wv, wo, bo, eps=1e-5, reuse_stream=False, frames=16):
    """t <- t + to_out(temporal self-attention(LayerNorm(t) + pe)) over the ``frames`` (16 or 32) frames of every pixel, C = 320, 8 heads, ONE
    launch (tattn.hip).  t: [nbatch * frames * hw, 320] bf16 in "(b f) (h w) c" row order, updated IN PLACE and returned; w*: [320, 320] fp32 or bf16."""
    _chk_bf16(t)
    C, F = 320, int(frames)
    assert t.shape == (nbatch * F * hw, C)
    gb = (beta.float()[None] + temporal_pe_table(F, C, t.device)).contiguous()
    ws = [w.to(torch.bfloat16).contiguous() for w in (wq, wk, wv, wo)]
    _lib.check(_lib.load().nr_op_tattn_fused_frames(_stream(), _ptr(t), nbatch, F, hw, None if reuse_stream else _ptr(ws[0]), _ptr(ws[1]),
                                                    _ptr(ws[2]), _ptr(ws[3]), _ptr(gamma.float().contiguous()), _ptr(gb),
                                                    _ptr(bo.float().contiguous()), float(eps)))
    return t


def tattn_head(t, nbatch, hw, gamma, beta, wq, wk, wv, eps=1e-5, reuse_stream=False):
    """a = temporal self-attention(LayerNorm(t) + pe) over the 16 frames of every pixel, BEFORE to_out, at C = 640 or 1280 (8 heads), ONE launch
    (tattnw.hip).  t: [nbatch * 16 * hw, C] bf16 in "(b f) (h w) c" row order; w*: [C, C]; returns a (same shape, bf16).  The LayerNorm fold is
    prepared here the way the engine prepares it (w_ln_linear / pe_projection): W' = bf16(gamma * W), c = row sums of W', b' = W beta, rv = pe W^T."""
    _chk_bf16(t)
    C, F = t.shape[1], 16
    assert C in (640, 1280) and t.shape[0] == nbatch * F * hw
    w = torch.cat([x.float() for x in (wq, wk, wv)], 0)                                  # [3C][C]
    wf = (w * gamma.float()[None, :]).to(torch.bfloat16).contiguous()
    lnc = wf.float().sum(1).contiguous()
    bias = (w.double() @ beta.double()).float().contiguous()
    rv = (temporal_pe_table(F, C, t.device).double() @ w.double().t()).float().contiguous()   # [16][3C]
    a = torch.empty_like(t)
    _lib.check(_lib.load().nr_op_tattn_head(_stream(), _ptr(t), _ptr(a), nbatch, hw, C, None if reuse_stream else _ptr(wf), _ptr(lnc), _ptr(bias),
                                            _ptr(rv), float(eps)))
    return a


def xattn_fused(t, nimg, hw, img_per_ctx, gamma, beta, wq, wo, bo, kv, Lk, eps=1e-5, reuse_streams=False):
    """t <- t + to_out(cross-attention(LayerNorm(t), context K | V)) at C = 320, 8 heads, in ONE launch (xattn.hip).  t: [nimg * hw, 320] bf16 in
    "(b f) (h w) c" row order, updated IN PLACE and returned; image i attends to context i // img_per_ctx; kv: [nctx * Lk, 640] bf16 (K | V columns:
    the fused to_k | to_v projection of the context); wq, wo: [320, 320]."""
    _chk_bf16(t, kv)
    C = 320
    assert t.shape == (nimg * hw, C) and kv.shape[1] == 2 * C and kv.shape[0] % Lk == 0
    nctx = kv.shape[0] // Lk
    ws = [w.to(torch.bfloat16).contiguous() for w in (wq, wo)]
    _lib.check(_lib.load().nr_op_xattn_fused(_stream(), _ptr(t), nimg, hw, img_per_ctx, None if reuse_streams else _ptr(ws[0]), _ptr(ws[1]), _ptr(kv),
                                             kv.shape[1], Lk, nctx, _ptr(gamma.float().contiguous()), _ptr(beta.float().contiguous()),
                                             _ptr(bo.float().contiguous()), float(eps)))
    return t


def xattn_head(t, nimg, hw, img_per_ctx, gamma, beta, wq, kv, Lk, eps=1e-5, reuse_streams=False):
    """a = cross-attention(LayerNorm(t), context K | V) BEFORE to_out at C = 640 or 1280 (8 heads), ONE launch (xattnw.hip).  t: [nimg * hw, C] bf16 in
    "(b f) (h w) c" row order; image i attends to context i // img_per_ctx; kv: [nctx * Lk, 2C] bf16 (K | V columns); wq: [C, C].  The LayerNorm fold is
    prepared here the way the engine prepares it (w_ln_linear)."""
    _chk_bf16(t, kv)
    C = t.shape[1]
    assert C in (640, 1280) and t.shape[0] == nimg * hw and kv.shape[1] == 2 * C and kv.shape[0] % Lk == 0
    nctx = kv.shape[0] // Lk
    w = wq.float()
    wf = (w * gamma.float()[None, :]).to(torch.bfloat16).contiguous()
    lnc = wf.float().sum(1).contiguous()
    bias = (w.double() @ beta.double()).float().contiguous()
    a = torch.empty_like(t)
    _lib.check(_lib.load().nr_op_xattn_head(_stream(), _ptr(t), _ptr(a), nimg, hw, img_per_ctx, C, None if reuse_streams else _ptr(wf), _ptr(lnc), _ptr(bias),
                                            _ptr(kv), kv.shape[1], Lk, nctx, float(eps)))
    return a


# ---------------------------------------------------------------------------------------------------
# Leaf-module handles (test hooks): ONE reference module planned as a network of its own, so the reference classes' own
# outputs (tests/golden/leaf_ops.npz) can be replayed at the row counts where the engine picks its fused kernels.
# ---------------------------------------------------------------------------------------------------
class NativeLeaf:
    """``kind='transformer3d'``: Transformer3DModel.forward (attention.py:95-142); ``kind='temporal'``: VanillaTemporalModule.forward
    (motion_module.py:79-86,134-158).  ``load_state_dict`` takes the module's own key names; ``forward`` takes / returns the
    reference's fp32 ``b c f h w`` tensors; ``op_descriptions()`` lists the launch plan (which kernel serves which layer)."""

    def __init__(self, kind, channels=320, heads=8, cross_attention_dim=768, norm_num_groups=32, num_attention_blocks=2, pe_max_len=24):
        import ctypes as C
        from .unet3d import _motion_keys, _transformer_keys
        assert kind in ("transformer3d", "temporal")
        self.kind = kind
        c = _lib.NrNetConfig()
        c.kind = _lib.NR_KIND_LEAF_TRANSFORMER3D if kind == "transformer3d" else _lib.NR_KIND_LEAF_TEMPORAL
        c.in_channels = c.out_channels = channels
        c.num_levels = 1
        c.block_out_channels[0] = channels
        c.num_heads = heads
        c.cross_attention_dim = cross_attention_dim
        c.norm_num_groups = norm_num_groups
        c.norm_eps = 1e-5
        c.use_motion_module = 1
        c.motion_num_heads = heads
        c.motion_num_attention_blocks = num_attention_blocks
        c.motion_pe_max_len = pe_max_len
        self._cconf = c
        self._schema = _transformer_keys("m", channels, cross_attention_dim) if kind == "transformer3d" else _motion_keys("m", channels, num_attention_blocks)
        self._h = C.c_void_p()
        self._plan_key = None
        _lib.check(_lib.load().nr_net_create(C.byref(c), C.byref(self._h)))

    def load_state_dict(self, sd):
        import ctypes as C
        import numpy as np
        lib = _lib.load()
        want = {k[2:]: v for k, v in self._schema.items()}
        missing = [k for k in want if k not in sd]
        assert not missing, missing
        for k, shape in want.items():
            v = sd[k]
            assert tuple(v.shape) == tuple(shape), (k, tuple(v.shape), shape)
            a = np.ascontiguousarray(v.detach().to("cpu", torch.float32).numpy())
            _lib.check(lib.nr_net_load_tensor(self._h, ("m." + k).encode(), a.ctypes.data_as(C.c_void_p), (C.c_int64 * a.ndim)(*a.shape), a.ndim))
        self._plan_key = None

    def forward(self, x, encoder_hidden_states=None, graph=True):
        _chk_f32(x, encoder_hidden_states)
        b, c, f, h, w = x.shape
        ctx_len = 0 if encoder_hidden_states is None else encoder_hidden_states.shape[1]
        lib = _lib.load()
        key = (b, f, h, w, ctx_len)
        if key != self._plan_key:
            _lib.check(lib.nr_net_plan(self._h, b, f, h, w, ctx_len))
            self._plan_key = key
        _lib.check(lib.nr_net_set_graph(self._h, 1 if graph else 0))
        _lib.check(lib.nr_net_invalidate_context(self._h))
        out = torch.empty_like(x)
        _lib.check(lib.nr_leaf_forward(self._h, _stream(), _ptr(x), _ptr(encoder_hidden_states), ctx_len, _ptr(out)))
        return out

    __call__ = forward

    def op_descriptions(self):
        lib = _lib.load()
        return [lib.nr_net_op_desc(self._h, i).decode() for i in range(lib.nr_net_num_ops(self._h))]

    def __del__(self):
        try:
            if self._h:
                _lib.load().nr_net_destroy(self._h)
                self._h = None
        except Exception:
            pass
