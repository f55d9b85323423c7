// GroupNorm(32 groups, per frame-image) [+SiLU] and LayerNorm [+temporal positional encoding] on
// channels-last bf16 activations.  HBM-bound kernels: 16-byte vector loads, fp32 statistics.
//
// Reference semantics:
//   InflatedGroupNorm / nn.GroupNorm on "(b f) c h w"   animatediff/models/resnet.py:21-29,
//       attention.py:61,105, motion_module.py:109,142, unet.py:245,468  (biased variance, eps inside sqrt)
//   nn.LayerNorm(dim) eps=1e-5                            attention.py:189,206,212; motion_module.py:201,207
//   PositionalEncoding add after the temporal LayerNorm   motion_module.py:241-243,274-278
//
// GroupNorm is two launches: gn_stats (per image x pixel-chunk partial sums, deterministic: no
// float atomics) and gn_apply (re-reduces the partials, folds gamma/beta/mean/rstd into a per-channel
// scale+shift held in LDS, streams the pixels).  Both accept a channel-concat of two sources so the
// up-block skip concat (unet_blocks.py:634,740) is never materialised for the norm.
#include "common.h"
#include <cstdlib>


namespace {

// grid (nchunk, nimg), block 256.  dynamic LDS: 2*PL*C floats, PL = pixel lanes (see below)
__global__ __launch_bounds__(256) void gn_stats_kernel(NrGnParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int C = p.c0 + p.c1;
  const int CP = C >> 3;                         // 16-byte chunks per pixel
  const int PL = CP <= 256 ? 256 / CP : 1;       // pixel lanes: thread = (pixel lane, channel chunk)
  float* chs = lds;                              // [PL][C] per-channel sums
  float* chq = lds + (size_t)PL * C;             // [PL][C] per-channel sums of squares
  const int img = blockIdx.y, chunk = blockIdx.x;
  const int pbeg = chunk * p.pix_per_blk;
  const int pend = min(p.hw, pbeg + p.pix_per_blk);
  const int tid = threadIdx.x;
  const int pl = CP <= 256 ? tid / CP : 0;
  if (pl < PL) {
    for (int cc = (CP <= 256 ? tid - pl * CP : tid); cc < CP; cc += 256) {
      const int c = cc << 3;
      const bf16* src; int ld;
      if (c < p.c0) { src = p.x0 + c; ld = p.ld0; } else { src = p.x1 + (c - p.c0); ld = p.ld1; }
      float s[8], q[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
      for (int px = pbeg + pl; px < pend; px += PL) {
        const bf16x8 v = *(const bf16x8*)(src + ((size_t)img * p.hw + px) * ld);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float f = (float)v[e]; s[e] += f; q[e] += f * f; }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) { chs[pl * C + c + e] = s[e]; chq[pl * C + c + e] = q[e]; }
    }
  }
  __syncthreads();
  // group reduce (fixed order => deterministic): 8 threads per group
  const int cg = C / p.groups;
  const int g = tid >> 3, sub = tid & 7;
  if (g < p.groups) {
    float s = 0.f, q = 0.f;
    for (int i = sub; i < cg * PL; i += 8) {
      const int l = i / cg, ci = i - l * cg;
      s += chs[l * C + g * cg + ci]; q += chq[l * C + g * cg + ci];
    }
    s += __shfl_xor(s, 1, 64); q += __shfl_xor(q, 1, 64);
    s += __shfl_xor(s, 2, 64); q += __shfl_xor(q, 2, 64);
    s += __shfl_xor(s, 4, 64); q += __shfl_xor(q, 4, 64);
    if (sub == 0) {
      float* o = p.partial + (((size_t)img * p.nchunk + chunk) * p.groups + g) * 2;
      o[0] = s; o[1] = q;
    }
  }
}

// grid (nchunk, nimg), block 256.  dynamic LDS: 2*C + 128 floats.
// ALL LDS of this kernel is dynamic.  With a static __shared__ array next to the dynamic region, hipGraph-captured launches were
// observed to get a workgroup LDS allocation that is short by the static part: the tail of `sh` then overlaps the LDS of whatever
// workgroup the CU places behind it.  Harmless while the network's launches run one after another, but with SparseCtrl and the
// U-Net encoder overlapped on two streams it gave run-to-run different results (2-3 % of the elements of a GroupNorm output, first
// seen at down_blocks.0.resnets.0; tools/race_taps.py).  Same rule as gemm.hip's LayerNorm exchange buffer.
__global__ __launch_bounds__(256) void gn_apply_kernel(NrGnParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int C = p.c0 + p.c1;
  const int CP = C >> 3;
  float* sc = lds;       // [C] scale
  float* sh = lds + C;   // [C] shift
  float* gmean = lds + 2 * C;        // [64]
  float* grstd = lds + 2 * C + 64;   // [64]
  const int img = blockIdx.y, chunk = blockIdx.x;
  const int tid = threadIdx.x;
  const int cg = C / p.groups;
  if (p.finalized) {
    if (tid < p.groups) {
      const float* o = p.partial + (size_t)p.nimg * p.nchunk * p.groups * 2 + ((size_t)img * p.groups + tid) * 2;
      gmean[tid] = o[0];
      grstd[tid] = o[1];
    }
  } else if (tid < p.groups) {
    float s = 0.f, q = 0.f;
    for (int k = 0; k < p.nchunk; ++k) {
      const float* o = p.partial + (((size_t)img * p.nchunk + k) * p.groups + tid) * 2;
      s += o[0]; q += o[1];
    }
    const float inv = 1.0f / ((float)cg * (float)p.hw);
    const float mean = s * inv;
    float var = q * inv - mean * mean;
    var = fmaxf(var, 0.f);
    gmean[tid] = mean;
    grstd[tid] = rsqrtf(var + p.eps);
  }
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    const int g = c / cg;
    const float a = p.gamma[c] * grstd[g];
    sc[c] = a;
    sh[c] = p.beta[c] - gmean[g] * a;
  }
  __syncthreads();
  const int pbeg = chunk * p.pix_per_blk;
  const int pend = min(p.hw, pbeg + p.pix_per_blk);
  if (CP <= 256) {
    // thread = (pixel lane, fixed 16-byte channel chunk): scale / shift of its 8 channels live in registers, no integer
    // division in the streaming loop, four independent loads in flight per thread
    const int PL = 256 / CP;
    const int pl = tid / CP, cc = tid - pl * CP;
    if (pl < PL) {
      const int c = cc << 3;
      float a8[8], b8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { a8[e] = sc[c + e]; b8[e] = sh[c + e]; }
      const bf16* src; int ld;
      if (c < p.c0) { src = p.x0 + c; ld = p.ld0; } else { src = p.x1 + (c - p.c0); ld = p.ld1; }
      const size_t img_row = (size_t)img * p.hw;
      bf16* dst = p.out + c;
      int px = pbeg + pl;
      for (; px + 3 * PL < pend; px += 4 * PL) {
        bf16x8 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *(const bf16x8*)(src + (img_row + px + u * PL) * ld);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float f = (float)v[u][e] * a8[e] + b8[e];
            if (p.silu) f = silu_f(f);
            o[e] = (bf16)f;
          }
          nr_store16(dst + (img_row + px + u * PL) * p.ldo, o);
        }
      }
      for (; px < pend; px += PL) {
        const bf16x8 v = *(const bf16x8*)(src + (img_row + px) * ld);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float f = (float)v[e] * a8[e] + b8[e];
          if (p.silu) f = silu_f(f);
          o[e] = (bf16)f;
        }
        nr_store16(dst + (img_row + px) * p.ldo, o);
      }
    }
    return;
  }
  const int total = (pend - pbeg) * CP;
  for (int idx = tid; idx < total; idx += 256) {
    const int px = pbeg + idx / CP;
    const int c = (idx % CP) << 3;
    const bf16* src; int ld;
    if (c < p.c0) { src = p.x0 + c; ld = p.ld0; } else { src = p.x1 + (c - p.c0); ld = p.ld1; }
    const size_t row = (size_t)img * p.hw + px;
    const bf16x8 v = *(const bf16x8*)(src + row * ld);
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float f = (float)v[e] * sc[c + e] + sh[c + e];
      if (p.silu) f = silu_f(f);
      o[e] = (bf16)f;
    }
    nr_store16(p.out + row * p.ldo + c, o);
  }
}

// Large images (VAE decoder levels: 10^4..10^6 pixels per image) use many small pixel chunks so the streaming
// kernels fill the chip; this kernel folds the per-chunk partials to mean / rstd once per (image, group) in a fixed
// order (deterministic) instead of every apply block re-reducing thousands of partials.  grid (groups, nimg).
__global__ __launch_bounds__(256) void gn_finalize_kernel(NrGnParams p) {
  __shared__ float red[2][4];
  const int g = blockIdx.x, img = blockIdx.y, tid = threadIdx.x;
  float s = 0.f, q = 0.f;
  for (int k = tid; k < p.nchunk; k += 256) {
    const float* o = p.partial + (((size_t)img * p.nchunk + k) * p.groups + g) * 2;
    s += o[0]; q += o[1];
  }
  s = wave_sum(s); q = wave_sum(q);
  if ((tid & 63) == 0) { red[0][tid >> 6] = s; red[1][tid >> 6] = q; }
  __syncthreads();
  if (tid == 0) {
    s = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    q = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    const int cg = (p.c0 + p.c1) / p.groups;
    const float inv = 1.0f / ((float)cg * (float)p.hw);
    const float mean = s * inv;
    float* o = p.partial + (size_t)p.nimg * p.nchunk * p.groups * 2 + ((size_t)img * p.groups + g) * 2;
    o[0] = mean;
    o[1] = rsqrtf(fmaxf(q * inv - mean * mean, 0.f) + p.eps);
  }
}

// Small-image GroupNorm(+SiLU) in ONE launch (the 8x8 and 4x4 levels, where the two-kernel version is pure launch
// latency): one workgroup per (image, group).  The group's hw x cg elements (<= 48 bf16 pairs per thread) are read
// once into registers, reduced with a fixed-order wave/LDS tree (deterministic), normalised and written.
template <int MAXP>
__global__ __launch_bounds__(256) void gn_fused_small_kernel(NrGnParams p) {
  __shared__ float red[2][4];
  const int C = p.c0 + p.c1;
  const int cg = C / p.groups;
  const int hp = cg >> 1;                       // bf16 pairs per pixel of this group
  const int img = blockIdx.y, g = blockIdx.x;
  const int cbase = g * cg;
  const int total = p.hw * hp;
  const int tid = threadIdx.x;
  bf16x2 v[MAXP];
  float s = 0.f, q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXP; ++i) {
    const int idx = tid + 256 * i;
    if (idx < total) {
      const int px = idx / hp, c = cbase + ((idx - px * hp) << 1);
      const bf16* src = c < p.c0 ? p.x0 + ((size_t)img * p.hw + px) * p.ld0 + c
                                 : p.x1 + ((size_t)img * p.hw + px) * p.ld1 + (c - p.c0);
      v[i] = *(const bf16x2*)src;
      const float a = (float)v[i][0], b = (float)v[i][1];
      s += a + b; q += a * a + b * b;
    }
  }
  s = wave_sum(s); q = wave_sum(q);
  if ((tid & 63) == 0) { red[0][tid >> 6] = s; red[1][tid >> 6] = q; }
  __syncthreads();
  s = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
  q = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  const float inv = 1.0f / ((float)cg * (float)p.hw);
  const float mean = s * inv;
  const float rstd = rsqrtf(fmaxf(q * inv - mean * mean, 0.f) + p.eps);
#pragma unroll
  for (int i = 0; i < MAXP; ++i) {
    const int idx = tid + 256 * i;
    if (idx < total) {
      const int px = idx / hp, c = cbase + ((idx - px * hp) << 1);
      float a = ((float)v[i][0] - mean) * rstd * p.gamma[c] + p.beta[c];
      float b = ((float)v[i][1] - mean) * rstd * p.gamma[c + 1] + p.beta[c + 1];
      if (p.silu) { a = silu_f(a); b = silu_f(b); }
      bf16x2 o; o[0] = (bf16)a; o[1] = (bf16)b;
      *(bf16x2*)(p.out + ((size_t)img * p.hw + px) * p.ldo + c) = o;
    }
  }
}

// GroupNorm(+SiLU) with ONE HBM read and ONE write (the 32x32 ... 4x4 levels of the denoisers): one 512- or 1024-thread workgroup per
// (image, set of GS consecutive groups).  The set's hw x (GS*cg) slab (<= 16 B x NV x 512 = 240 KiB of registers per CU) is
// read once as 16-byte chunks, reduced deterministically (fixed-order LDS tree, no atomics), normalised and written.
// Thread t owns chunk column c = t % cpp of pixels plane + NPL*i, so its 8 channels, their (at most two, cg >= 8) groups
// and gamma/beta are fixed: the statistics use v_dot2_f32_bf16 on the raw pairs (sum / sum of squares of all 8 elements, and
// of the elements of the column's first group selected by a bit mask), the apply is one fma per element.
// Workgroups of one image are placed on ONE XCD (its L2 then serves the partially used lines of the narrow slabs).
template <int NV, int T>
__global__ __launch_bounds__(T) void gn_slab_kernel(NrGnParams p, int GS) {
  __shared__ float red[T][4];
  __shared__ float colsum[64][4];
  __shared__ float gstat[16][2];
  const int C = p.c0 + p.c1;
  const int cg = C / p.groups;
  const int cpp = GS * cg / 8;                  // 16-byte chunks per pixel of this slab (<= 64)
  const int NPL = T / cpp;                      // pixel lanes
  const int sets = p.groups / GS;
  int bid = blockIdx.x;
  {
    const int nb = gridDim.x;
    if ((nb & 7) == 0) bid = (bid & 7) * (nb >> 3) + (bid >> 3);     // XCD x gets the contiguous range [x*nb/8, (x+1)*nb/8)
  }
  const int img = bid / sets, set = bid - img * sets;
  const int tid = threadIdx.x;
  const int c = tid % cpp, plane = tid / cpp;
  const bool active = plane < NPL;
  const int ch0 = set * GS * cg + c * 8;        // first of this thread's 8 channels
  const bf16* src; int ld;
  if (ch0 < p.c0) { src = p.x0 + ch0; ld = p.ld0; } else { src = p.x1 + (ch0 - p.c0); ld = p.ld1; }
  src += (size_t)img * p.hw * ld;
  bf16x8 v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int px = plane + NPL * i;
    if (active && px < p.hw) v[i] = *(const bf16x8*)(src + (size_t)px * ld);
    else v[i] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
  }
  // elements [0, nlo) of the chunk belong to the column's first group, the rest to the next one
  const int g_first = (c * 8) / cg;
  const int nlo = min(8, (g_first + 1) * cg - c * 8);
  unsigned mask[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) mask[j] = (2 * j < nlo ? 0xFFFFu : 0u) | (2 * j + 1 < nlo ? 0xFFFF0000u : 0u);
  const bf16x2 one2 = {(bf16)1.0f, (bf16)1.0f};
  float s_all = 0.f, q_all = 0.f, s_lo = 0.f, q_lo = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const unsigned* w = reinterpret_cast<const unsigned*>(&v[i]);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned um = w[j] & mask[j];
      const bf16x2 x = __builtin_bit_cast(bf16x2, w[j]);
      const bf16x2 xm = __builtin_bit_cast(bf16x2, um);
      s_all = __builtin_amdgcn_fdot2_f32_bf16(x, one2, s_all, false);
      q_all = __builtin_amdgcn_fdot2_f32_bf16(x, x, q_all, false);
      s_lo = __builtin_amdgcn_fdot2_f32_bf16(xm, one2, s_lo, false);
      q_lo = __builtin_amdgcn_fdot2_f32_bf16(xm, xm, q_lo, false);
    }
  }
  red[tid][0] = s_lo; red[tid][1] = q_lo; red[tid][2] = s_all - s_lo; red[tid][3] = q_all - q_lo;
  __syncthreads();
  // column sums over the pixel lanes: 8 threads per column, fixed order
  {
    const int col = tid >> 3, r = tid & 7;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    if (col < cpp)
      for (int pl = r; pl < NPL; pl += 8) {
        const float* e = red[col + cpp * pl];
        a[0] += e[0]; a[1] += e[1]; a[2] += e[2]; a[3] += e[3];
      }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      a[k] += __shfl_xor(a[k], 1, 64); a[k] += __shfl_xor(a[k], 2, 64); a[k] += __shfl_xor(a[k], 4, 64);
    }
    if (col < cpp && r == 0) { colsum[col][0] = a[0]; colsum[col][1] = a[1]; colsum[col][2] = a[2]; colsum[col][3] = a[3]; }
  }
  __syncthreads();
  if (tid < GS) {
    float s = 0.f, q = 0.f;
    for (int cc = 0; cc < cpp; ++cc) {
      const int gf = (cc * 8) / cg;
      if (gf == tid) { s += colsum[cc][0]; q += colsum[cc][1]; }
      else if (gf + 1 == tid) { s += colsum[cc][2]; q += colsum[cc][3]; }      // zero when the chunk does not straddle
    }
    const float inv = 1.0f / ((float)cg * (float)p.hw);
    const float mean = s * inv;
    gstat[tid][0] = mean;
    gstat[tid][1] = rsqrtf(fmaxf(q * inv - mean * mean, 0.f) + p.eps);
  }
  __syncthreads();
  if (!active) return;
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int g = e < nlo ? g_first : g_first + 1;
    sc[e] = gstat[g][1] * p.gamma[ch0 + e];
    sh[e] = p.beta[ch0 + e] - gstat[g][0] * sc[e];
  }
  bf16* dst = p.out + (size_t)img * p.hw * p.ldo + ch0;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int px = plane + NPL * i;
    if (px < p.hw) {
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float f = (float)v[i][e] * sc[e] + sh[e];
        if (p.silu) f = silu_f(f);
        o[e] = (bf16)f;
      }
      nr_store16(dst + (size_t)px * p.ldo, o);
    }
  }
}

// LayerNorm over the last dim C (C % 8 == 0).  TPR threads cooperate on one row (TPR in {8,16,32,64}, each
// thread holds <= MAXV 16-byte chunks), 256/TPR rows per block: every lane is busy and has 2-3 loads in
// flight even at C = 320 (640-byte rows).  Two-pass statistics in registers (mean, then centred variance).
// pe: optional [pe_len][C] fp32 table added after the affine; frame index = (row / pe_hw) % pe_F.
template <int TPR, int MAXV>
__global__ __launch_bounds__(256) void layernorm_kernel(const bf16* __restrict__ x, int ldx, bf16* __restrict__ out,
                                                        int ldo, int M, int C, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps,
                                                        const float* __restrict__ pe, int pe_hw, int pe_F) {
  constexpr int RPB = 256 / TPR;
  const int sub = threadIdx.x % TPR;
  const int row = blockIdx.x * RPB + threadIdx.x / TPR;
  const bool rok = row < M;
  const int rowc = rok ? row : M - 1;
  const int CP = C >> 3;
  float v[MAXV][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int cc = sub + TPR * i;
    if (cc < CP) {
      const bf16x8 t = *(const bf16x8*)(x + (size_t)rowc * ldx + cc * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) { v[i][e] = (float)t[e]; s += v[i][e]; }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
    }
  }
#pragma unroll
  for (int o = TPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float mean = s / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int cc = sub + TPR * i;
    if (cc < CP) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mean; q += d * d; }
    }
  }
#pragma unroll
  for (int o = TPR / 2; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
  const float rstd = rsqrtf(q / (float)C + eps);
  if (!rok) return;
  const float* perow = pe ? pe + (size_t)((row / pe_hw) % pe_F) * C : nullptr;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int cc = sub + TPR * i;
    if (cc < CP) {
      const f32x4 g0 = *(const f32x4*)(gamma + cc * 8), g1 = *(const f32x4*)(gamma + cc * 8 + 4);
      const f32x4 b0 = *(const f32x4*)(beta + cc * 8), b1 = *(const f32x4*)(beta + cc * 8 + 4);
      float f[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        f[e] = (v[i][e] - mean) * rstd * g0[e] + b0[e];
        f[4 + e] = (v[i][4 + e] - mean) * rstd * g1[e] + b1[e];
      }
      if (perow) {
        const f32x4 p0 = *(const f32x4*)(perow + cc * 8), p1 = *(const f32x4*)(perow + cc * 8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { f[e] += p0[e]; f[4 + e] += p1[e]; }
      }
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (bf16)f[e];
      nr_store16(out + (size_t)row * ldo + cc * 8, o);
    }
  }
}

template <int TPR, int MAXV>
void launch_ln(const bf16* x, int ldx, bf16* out, int ldo, int M, int C, const float* gamma, const float* beta, float eps,
               const float* pe, int pe_hw, int pe_F, hipStream_t stream) {
  constexpr int RPB = 256 / TPR;
  hipLaunchKernelGGL((layernorm_kernel<TPR, MAXV>), dim3((M + RPB - 1) / RPB), dim3(256), 0, stream, x, ldx, out, ldo, M, C,
                     gamma, beta, eps, pe, pe_hw, pe_F);
}

}  // namespace

extern "C" int nr_gn_workspace_floats(int nimg, int hw, int groups, int* pix_per_blk_out, int* nchunk_out) {
  // aim for >= ~16 chunks per image but at least 8 and at most 128 pixels per block (large images: many chunks,
  // folded by gn_finalize); small batches (sgm U-Net: 2 images, VAE keyframe: 1) shrink the chunks until the streaming
  // kernels have >= 256 blocks
  int ppb = (hw + 15) / 16;
  if (ppb < 8) ppb = 8;
  if (ppb > 128) ppb = 128;
  if (ppb > hw) ppb = hw;
  while (ppb > 2 && (long long)nimg * ((hw + ppb - 1) / ppb) < 256) ppb = (ppb + 1) / 2;
  const int nchunk = (hw + ppb - 1) / ppb;
  if (pix_per_blk_out) *pix_per_blk_out = ppb;
  if (nchunk_out) *nchunk_out = nchunk;
  return nimg * nchunk * groups * 2 + nimg * groups * 2;
}

static int gn_launch_impl(NrGnParams* pp, hipStream_t stream, bool count_only, int* nlaunch);
extern "C" int nr_launch_groupnorm(NrGnParams* pp, hipStream_t stream) { int n = 0; return gn_launch_impl(pp, stream, false, &n); }
// kernels nr_launch_groupnorm enqueues for this shape (1 slab / per-group kernel, 2-3 for the chunked passes): launch accounting only
extern "C" int nr_groupnorm_launches(const NrGnParams* pp) {
  NrGnParams q = *pp;
  int n = 0;
  return gn_launch_impl(&q, nullptr, true, &n) == 0 ? n : 1;
}
#define GNL(...) do { ++*nlaunch; if (!count_only) hipLaunchKernelGGL(__VA_ARGS__); } while (0)
static int gn_launch_impl(NrGnParams* pp, hipStream_t stream, bool count_only, int* nlaunch) {
  NrGnParams p = *pp;
  const int C = p.c0 + p.c1;
  const int pn = p.plan_nimg > 0 && p.plan_nimg < p.nimg ? p.plan_nimg : p.nimg;   // images the variant choices are made for (common.h)
  if (C % 8 != 0 || C % p.groups != 0 || p.groups > 64) return 1;
  if (p.x1 && p.c0 % 8 != 0) return 2;
  {
    // register-resident slab version: one read + one write (see gn_slab_kernel).  Needs >= 8 channels per group (a 16-byte chunk
    // then touches at most two groups), 16-byte aligned slabs and concat split, and a slab of <= 512 x 32 chunks.
    const int cg = C / p.groups;
    static const bool noslab = getenv("NR_GN_SLAB") != nullptr && atoi(getenv("NR_GN_SLAB")) == 0;
    if (!noslab && cg >= 8 && cg % 2 == 0 && p.c0 % 8 == 0 && p.ld0 % 8 == 0 && (p.c1 == 0 || p.ld1 % 8 == 0) && p.ldo % 8 == 0) {
      int GS = 0;
      for (int g = 1; g <= 8 && g <= p.groups; g *= 2) {            // smallest set whose span is a whole number of chunks
        if (p.groups % g == 0 && (g * cg) % 8 == 0 && g * cg / 8 <= 64) { GS = g; break; }
      }
      // widen the slab (better line use) while the chip stays filled and the registers suffice
      while (GS && GS * 2 <= 8 && p.groups % (GS * 2) == 0 && GS * 2 * cg / 8 <= 64 &&
             (long long)pn * (p.groups / (GS * 2)) >= 256 &&
             ((long long)p.hw + (512 / (GS * 2 * cg / 8)) - 1) / (512 / (GS * 2 * cg / 8)) <= 32) GS *= 2;
      if (GS && p.hw > 64) {       // hw <= 64: the one-workgroup-per-group kernel below is as fast or faster (measured, tools/gn_ab.sh)
        const int cpp = GS * cg / 8;
        static const int tforce = getenv("NR_GN_T") ? atoi(getenv("NR_GN_T")) : 0;
        // 512 threads (2 waves per SIMD); NR_GN_T=1024 measured 3-6 % slower at every shape of config 2 (tools/gn_ab.sh)
        const int T = tforce == 1024 ? 1024 : 512;
        const int NPL = T / cpp;
        const int nv = (p.hw + NPL - 1) / NPL;
        const unsigned grid = (unsigned)(p.nimg * (p.groups / GS));
#define NR_GN_SLAB(NVV)                                                                                                  \
  do {                                                                                                                   \
    if (T == 1024) GNL((gn_slab_kernel<NVV, 1024>), dim3(grid), dim3(1024), 0, stream, p, GS);            \
    else GNL((gn_slab_kernel<NVV, 512>), dim3(grid), dim3(512), 0, stream, p, GS);                        \
    return 0;                                                                                                            \
  } while (0)
        if (nv <= 2) NR_GN_SLAB(2);
        else if (nv <= 4) NR_GN_SLAB(4);
        else if (nv <= 8) NR_GN_SLAB(8);
        else if (nv <= 12) NR_GN_SLAB(12);
        else if (nv <= 16) NR_GN_SLAB(16);
#undef NR_GN_SLAB
        // 24 / 32 chunks per thread only with 512 threads (at 1024 threads the 128-VGPR budget would spill)
        else if (nv <= 24 && T == 512) { GNL((gn_slab_kernel<24, 512>), dim3(grid), dim3(512), 0, stream, p, GS); return 0; }
        else if (nv <= 32 && T == 512) { GNL((gn_slab_kernel<32, 512>), dim3(grid), dim3(512), 0, stream, p, GS); return 0; }
      }
    }
  }
  {
    // small images: single fused launch (needs an even channels-per-group and an even split point of the concat)
    const int cg = C / p.groups;
    const long long pairs = (long long)p.hw * (cg / 2);
    static const bool small_on = !(getenv("NR_GN_SMALL") && getenv("NR_GN_SMALL")[0] == '0');   // bisect switch (profiles/r03_race_*)
    if (small_on && p.hw <= 64 && cg % 2 == 0 && p.c0 % 2 == 0 && pairs <= 256LL * 48) {
      dim3 grid(p.groups, p.nimg);
      if (pairs <= 256LL * 8) GNL((gn_fused_small_kernel<8>), grid, dim3(256), 0, stream, p);
      else if (pairs <= 256LL * 16) GNL((gn_fused_small_kernel<16>), grid, dim3(256), 0, stream, p);
      else GNL((gn_fused_small_kernel<48>), grid, dim3(256), 0, stream, p);
      return 0;
    }
  }
  nr_gn_workspace_floats(pn, p.hw, p.groups, &p.pix_per_blk, &p.nchunk);     // chunking as for pn images (the engine sizes `partial` for it)
  const int CP = C / 8;
  const int PL = CP <= 256 ? 256 / CP : 1;
  const size_t shm_stats = (size_t)2 * PL * C * sizeof(float);
  const size_t shm_apply = (size_t)(2 * C + 128) * sizeof(float);
  if (shm_stats > 60000 || shm_apply > 60000) return 3;
  dim3 grid(p.nchunk, p.nimg);
  p.finalized = p.nchunk > 16 ? 1 : 0;
  GNL(gn_stats_kernel, grid, dim3(256), shm_stats, stream, p);
  if (p.finalized) GNL(gn_finalize_kernel, dim3(p.groups, p.nimg), dim3(256), 0, stream, p);
  GNL(gn_apply_kernel, grid, dim3(256), shm_apply, stream, p);
  return 0;
}
#undef GNL

extern "C" int nr_launch_layernorm(const bf16* x, int ldx, bf16* out, int ldo, int M, int C, const float* gamma,
                                   const float* beta, float eps, const float* pe, int pe_hw, int pe_F,
                                   hipStream_t stream) {
  if (C % 8 != 0 || M <= 0) return 1;
  const int CP = C / 8;
#define NR_LN(T, V) launch_ln<T, V>(x, ldx, out, ldo, M, C, gamma, beta, eps, pe, pe_hw, pe_F, stream)
  if (CP <= 8) NR_LN(8, 1);
  else if (CP <= 16) NR_LN(16, 1);
  else if (CP <= 24) NR_LN(8, 3);
  else if (CP <= 48) NR_LN(16, 3);      // C = 320 (40 chunks)
  else if (CP <= 96) NR_LN(32, 3);      // C = 640
  else if (CP <= 192) NR_LN(64, 3);     // C = 1280
  else if (CP <= 512) NR_LN(64, 8);
  else return 2;
#undef NR_LN
  return 0;
}
