// Implicit-GEMM convolution / Linear kernel on bf16 MFMA (v_mfma_f32_16x16x32_bf16), gfx950.
//
// One kernel serves every GEMM-shaped op on the denoiser path:
//   * 3x3 convs, stride 1/2, optional nearest-2x upsample gather, optional channel-concat of two
//     sources (reference: InflatedConv3d animatediff/models/resnet.py:10-18, Upsample3D :32-80,
//     Downsample3D :83-106, skip concat unet_blocks.py:634,740)
//   * 1x1 convs / nn.Linear (proj_in/out, to_q/k/v/out, GEGLU FF, conv_shortcut, ControlNet zero-convs)
// with fused epilogues: +bias, +time-embedding row vector (resnet.py:193-194), *scale, +residual,
// GEGLU gate (motion_module_new.py:516-518).
//
// Tiling: BMxBN block tile, BK=64, 256 threads = 2x2 waves, each wave (BM/2)x(BN/2) as 16x16 MFMA
// tiles.  The MFMA is issued "transposed" (weights as the A operand, activations as B) so that a
// lane's 4 accumulator registers are 4 consecutive output channels of one pixel -> 8-byte stores.
// LDS tiles are [rows][64] bf16 (128-B rows) with the 16-B chunk index XOR-swizzled by (row&7)
// so ds_read_b128 fragment reads are bank-conflict free.  Global->LDS is register staged and
// software pipelined (loads for tile k+1 in flight while tile k is multiplied).
#include "common.h"

namespace {

template <int BM, int BN>
__global__ __launch_bounds__(256) void igemm_bf16_kernel(NrGemmParams p) {
  constexpr int BK = 64;
  constexpr int WM = BM / 2, WN = BN / 2;
  constexpr int MT = WM / 16, NT = WN / 16;
  constexpr int AI = BM / 32, BI = BN / 32;
  __shared__ __attribute__((aligned(16))) bf16 smem[(BM + BN) * BK];
  bf16* sA = smem;            // activations tile  [BM][64]
  bf16* sB = smem + BM * BK;  // weights tile      [BN][64]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntn = (p.N + BN - 1) / BN;
  const int bm = blockIdx.x / ntn, bn = blockIdx.x - bm * ntn;
  const int m0 = bm * BM, n0 = bn * BN;
  const int chunk = tid & 7;   // 16-byte chunk within the 128-byte k-tile row
  const int lrow = tid >> 3;   // 0..31

  const int Cin = p.c0 + p.c1;

  // ---- per-thread A-row bookkeeping (rows lrow + 32*i of the block tile) ----
  int a_pix[AI];  // ksize==1: linear pixel index; ksize==3: image index n
  int a_oy[AI], a_ox[AI];
  bool a_ok[AI];
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int m = m0 + lrow + 32 * i;
    a_ok[i] = m < p.M;
    const int mm = a_ok[i] ? m : 0;
    if (p.ksize == 1) {
      a_pix[i] = mm; a_oy[i] = 0; a_ox[i] = 0;
    } else {
      const int ohw = p.OH * p.OW;
      const int n = mm / ohw;
      const int r = mm - n * ohw;
      a_pix[i] = n; a_oy[i] = r / p.OW; a_ox[i] = r - a_oy[i] * p.OW;
    }
  }
  const bf16* wrow[BI];
  bool b_ok[BI];
#pragma unroll
  for (int i = 0; i < BI; ++i) {
    const int n = n0 + lrow + 32 * i;
    b_ok[i] = n < p.N;
    wrow[i] = p.w + (size_t)(b_ok[i] ? n : 0) * p.K + chunk * 8;
  }

  bf16x8 ra[AI], rb[BI];
  const bf16x8 zero8 = bf16x8_zero();

  auto load_tiles = [&](int kt) {
    const int kbase = kt * BK;
    int tap = 0, c = kbase;
    if (p.ksize == 3) { tap = kbase / Cin; c = kbase - tap * Cin; }
    const bf16* src; int ld;
    if (c < p.c0) { src = p.a0 + c; ld = p.lda0; } else { src = p.a1 + (c - p.c0); ld = p.lda1; }
    if (p.ksize == 1) {
#pragma unroll
      for (int i = 0; i < AI; ++i)
        ra[i] = a_ok[i] ? *(const bf16x8*)(src + (size_t)a_pix[i] * ld + chunk * 8) : zero8;
    } else {
      const int ky = tap / 3, kx = tap - ky * 3;
      // virtual (post-upsample) input extent
      const int VH = p.ups ? p.H * 2 : p.H, VW = p.ups ? p.W * 2 : p.W;
#pragma unroll
      for (int i = 0; i < AI; ++i) {
        int iy = a_oy[i] * p.stride + ky - 1;
        int ix = a_ox[i] * p.stride + kx - 1;
        const bool ok = a_ok[i] && iy >= 0 && iy < VH && ix >= 0 && ix < VW;
        if (p.ups) { iy >>= 1; ix >>= 1; }
        const size_t pix = ((size_t)a_pix[i] * p.H + iy) * p.W + ix;
        ra[i] = ok ? *(const bf16x8*)(src + pix * ld + chunk * 8) : zero8;
      }
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) rb[i] = b_ok[i] ? *(const bf16x8*)(wrow[i] + kbase) : zero8;
  };

  auto store_tiles = [&]() {
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int row = lrow + 32 * i;
      *(bf16x8*)(sA + row * BK + ((chunk ^ (row & 7)) << 3)) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int row = lrow + 32 * i;
      *(bf16x8*)(sB + row * BK + ((chunk ^ (row & 7)) << 3)) = rb[i];
    }
  };

  f32x4 acc[NT][MT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / BK;
  const int fr = lane & 15, fg = lane >> 4;

  load_tiles(0);
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();
    store_tiles();
    __syncthreads();
    if (kt + 1 < nk) load_tiles(kt + 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 wf[NT], xf[MT];
      const int lc = ks * 4 + fg;
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const int row = wn * WN + i * 16 + fr;
        wf[i] = *(const bf16x8*)(sB + row * BK + ((lc ^ (row & 7)) << 3));
      }
#pragma unroll
      for (int j = 0; j < MT; ++j) {
        const int row = wm * WM + j * 16 + fr;
        xf[j] = *(const bf16x8*)(sA + row * BK + ((lc ^ (row & 7)) << 3));
      }
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
    }
  }

  // ---- epilogue: lane holds out[m = ..+fr][n = ..+4*fg + r], r = 0..3 ----
#pragma unroll
  for (int j = 0; j < MT; ++j) {
    const int m = m0 + wm * WM + j * 16 + fr;
    if (m >= p.M) continue;
    const float* rv = p.rowvec ? p.rowvec + (size_t)(m / p.rowvec_div) * p.rowvec_ld : nullptr;
    if (!p.geglu) {
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const int n = n0 + wn * WN + i * 16 + 4 * fg;
        if (n >= p.N) continue;
        f32x4 v = acc[i][j];
        if (p.bias) { const f32x4 b = *(const f32x4*)(p.bias + n); v += b; }
        if (rv) { const f32x4 t = *(const f32x4*)(rv + n); v += t; }
        v *= p.out_scale;
        if (p.res) {
          const bf16x4 r = *(const bf16x4*)(p.res + (size_t)m * p.ldr + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += (float)r[e];
        }
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16)v[e];
        *(bf16x4*)(p.out + (size_t)m * p.ldo + n) = o;
      }
    } else {
#pragma unroll
      for (int i = 0; i < NT; i += 2) {
        const int nv = n0 + wn * WN + i * 16 + 4 * fg;  // value columns (permuted W row index)
        if (nv >= p.N) continue;
        const int ng = nv + 16;                          // matching gate columns
        f32x4 v = acc[i][j], g = acc[i + 1][j];
        if (p.bias) {
          v += *(const f32x4*)(p.bias + nv);
          g += *(const f32x4*)(p.bias + ng);
        }
        const int oc = ((n0 + wn * WN + i * 16) >> 1) + 4 * fg;
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16)(v[e] * gelu_erf_f(g[e]));
        *(bf16x4*)(p.out + (size_t)m * p.ldo + oc) = o;
      }
    }
  }
}

}  // namespace

// Host launcher.  Returns 0 on success, nonzero on unsupported shape.
extern "C" int nr_launch_igemm(const NrGemmParams* pp, hipStream_t stream) {
  const NrGemmParams& p = *pp;
  const int Cin = p.c0 + p.c1;
  if (p.K % 64 != 0 || Cin % 64 != 0 || p.N % 32 != 0) return 1;
  if (p.a1 && (p.c0 % 64 != 0)) return 2;
  if (p.K != p.ksize * p.ksize * Cin) return 3;
  if (p.ksize != 1 && p.ksize != 3) return 4;
  if (p.geglu && (p.N % 32 != 0)) return 5;
  auto nblk = [&](int bm, int bn) { return (long long)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn); };
  // pick the largest tile that still yields >= 2 blocks per CU; else the smallest tile
  if (nblk(128, 128) >= 512 && p.N % 128 == 0) {
    hipLaunchKernelGGL((igemm_bf16_kernel<128, 128>), dim3((unsigned)nblk(128, 128)), dim3(256), 0, stream, p);
  } else if (nblk(128, 64) >= 512) {
    hipLaunchKernelGGL((igemm_bf16_kernel<128, 64>), dim3((unsigned)nblk(128, 64)), dim3(256), 0, stream, p);
  } else {
    hipLaunchKernelGGL((igemm_bf16_kernel<64, 64>), dim3((unsigned)nblk(64, 64)), dim3(256), 0, stream, p);
  }
  return 0;
}
