// 256-row "ping-pong" implicit-GEMM kernel for the BIG GEMM-shaped launches of the denoiser path (round 4): the SparseCtrl groups
// (M = 40 960 / 163 840 rows), BASELINE configs 4 / 5 (8 clips per call; 32 frames x 64x64), the first-stage VAE.  Same arithmetic and
// the same fused epilogues as gemm.hip (C[M][N] = epilogue(A[M][K] W[N][K]^T), A an implicit im2col view: reference InflatedConv3d
// animatediff/models/resnet.py:10-18 and the nn.Linear / 1x1 convs of attention.py, motion_module.py), bit-compatible with it (same k order
// per accumulator), but another loop structure, because the tiled igemm tops out at ~0.34 of the dense MFMA peak whatever M is:
//
//   * tile 256 (M) x 64 NT (N), NT = 2..5 -> 128 / 192 / 256 / 320 columns, BK = 64, ONE 512-thread workgroup per CU: 8 waves as 2 (M) x 4 (N),
//     a wave owns 128 x 16 NT outputs = 8 x NT accumulator tiles of v_mfma_f32_16x16x32_bf16 (weights = A operand, so a lane holds 4
//     consecutive output channels of one pixel, as in gemm.hip).  Per FLOP this stages 0.6x the LDS-DMA pieces of the 128 x 160 tile.
//   * the two wave groups (waves 0-3 / 4-7 = the two M halves: one wave of each on every SIMD) run the SAME program staggered by one
//     barrier: while one group issues its MFMAs (a "phase" = 2 row tiles x NT x 2 k-steps = 4 NT MFMAs, setprio 1), its SIMD partners read
//     fragments from LDS and issue LDS-DMA, then they swap.  The matrix pipe of a SIMD always has one wave feeding it
//     (MI355X_MICROARCH.md "Two waves per SIMD" item 9; cdna_hip_programming.md 5.5 T3+T4+T5).
//   * k-tile t is computed in NPH = 2 (default for tiles of <= 256 columns) or 4 phases; the description below is the 4-phase form, the
//     2-phase form merges phases {0,1} and {2,3} (half the barriers; the weight region is then re-staged one interval after its last read,
//     so the weight-fragment reads are retired by an lgkmcnt(0) BEFORE the phase's first barrier).  Weight fragments (NT x 2) are read ONCE per k-tile in phase 0 and stay in registers, activation
//     fragments (2 x 2) per phase.  LDS-DMA (asm global_load_lds_dwordx4, 1 KiB pieces, XOR swizzle on the SOURCE address) is spread over
//     the phases, two or three pieces per wave and phase: phases 0-1 stage the activation half-tile of k-tile t+1 (each group stages and
//     reads ITS OWN 128 rows), phases 2-3 the weight tile of k-tile t+2 -- the weight region of the current buffer is free by then,
//     because weights are consumed in phase 0.  Nothing is ever drained: one counted s_waitcnt vmcnt(N) after the MFMAs of phase 2
//     (weights of t+1 landed) and one after phase 3 (activations of t+1 landed), each followed by the phase's closing barrier, one
//     interval before the first read (RAW: wait -> barrier -> read; WAR: a region is re-staged only after a barrier that follows the
//     lgkmcnt(0) of its last readers).  Two LDS buffers of (256 + 64 NT) x 128 B.
//   * epilogue through LDS in two (NT <= 4) or four rounds (fp32 tile, XOR-swizzled; both wave groups park their rows side by side), 16-byte
//     coalesced bias / row-vector / residual / output accesses, identical arithmetic to gemm.hip's staged epilogue (incl. GEGLU, even NT).
// Not here (the launcher falls back to gemm.hip): split-K, the LayerNorm-folded variant, raw fp32 output, strided / upsampling /
// tap-major 3x3 convs, grids that would leave the chip under-filled.
#include "common.h"
#include <cstdlib>

namespace {

__device__ __attribute__((aligned(16))) const unsigned int g8p_zero16[4] = {0u, 0u, 0u, 0u};
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void glds16(const void* src, unsigned lds_wave_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(lds_wave_base) : "memory", "m0");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ int fdiv_small(int a, int d) {      // exact for 0 <= a < 2^24, quotient < 2^22 (see gemm.hip)
  int q = (int)((float)a * __builtin_amdgcn_rcpf((float)d));
  const int r = a - q * d;
  q += (r >= d ? 1 : 0) - (r < 0 ? 1 : 0);
  return q;
}
__device__ __forceinline__ size_t rowvec_row(const NrGemmParams& p, int m) {
  int r = m / p.rowvec_div;
  if (p.rowvec_mod) r %= p.rowvec_mod;
  return (size_t)r * p.rowvec_ld;
}

#ifndef NR_G8P_ABLATE
#define NR_G8P_ABLATE 0     // timing ablations (experiments build only, results wrong): 1 no LDS-DMA in the loop, 2 no fragment reads in the loop, 4 no MFMAs
#endif

constexpr int G8_BM = 256, G8_BK = 64;

template <int NT, bool TAPI, int NPH = 4>
__global__ __launch_bounds__(512, 2) void g8p_kernel(NrGemmParams p_arg, int m_fast) {
  const NrGemmParams p = nr_pin_params(p_arg);      // one batch of scalar loads at entry instead of a round trip per first use (common.h)
  m_fast = nr_pin(m_fast);
  constexpr int BN = 64 * NT, WN = 16 * NT;
  constexpr int A_BYTES = G8_BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  static_assert(NPH == 4 || (NPH == 2 && NT <= 4), "phases per k-tile");
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 x STAGE

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wave >> 2, wn = wave & 3;             // M half (= stagger group), N quarter
  const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + G8_BM - 1) / G8_BM;
  int bid;
  {                                                    // XCD-contiguous tile ranges (bijective), as gemm.hip
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7, local = orig >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
  }
  int bm, bn;
  if (m_fast >= 2) {
    const int G = m_fast;
    const int band = fdiv_small(bid, G * ntn);
    const int first = band * G;
    const int gsz = min(G, ntm - first);
    const int r = bid - band * G * ntn;
    bn = fdiv_small(r, gsz);
    bm = first + r - bn * gsz;
  } else if (m_fast) { bn = fdiv_small(bid, ntm); bm = bid - bn * ntm; } else { bm = fdiv_small(bid, ntn); bn = bid - bm * ntn; }
  const int m0 = bm * G8_BM, n0 = bn * BN;

  const int lr = lane >> 3, lp = lane & 7;
  const unsigned lchunk_b = (unsigned)((lp ^ lr) << 4);          // byte offset of the logical 16-byte chunk this lane fetches
  const char* zsrc = reinterpret_cast<const char*>(g8p_zero16);

  // ---- per-lane staging state: 32-bit byte offsets from wave-uniform 64-bit bases (the launcher guarantees every source < 4 GiB) ----
  // activation pieces j = 0..3 of this wave: tile rows 8 * (16 g + 4 wn + j) + lr  (the group's own half)
  unsigned a_off[4];            // TAPI: centre pixel; 1x1: row start (current source); + lchunk_b
  unsigned a_msk = 0, a_msk3 = 0;   // TAPI: 9 in-image tap bits per piece (pieces 0-2 in a_msk, piece 3 in a_msk3); 1x1: bit j of a_msk = row valid
  int a_m[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int m = m0 + 8 * (16 * g + 4 * wn + j) + lr;
    const bool ok = m < p.M;
    a_m[j] = ok ? m : 0;
    if constexpr (TAPI) {
      const int ohw = p.OH * p.OW;
      const int n = fdiv_small(a_m[j], ohw);
      const int r = a_m[j] - n * ohw;
      const int oy = fdiv_small(r, p.OW), ox = r - oy * p.OW;
      a_off[j] = (unsigned)((((size_t)n * p.H + oy) * p.W + ox) * (size_t)p.lda0 * sizeof(bf16)) + lchunk_b;
      unsigned msk = 0;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int iy = oy + t / 3 - 1, ix = ox + t % 3 - 1;
        if (ok && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) msk |= 1u << t;
      }
      if (j < 3) a_msk |= msk << (9 * j);
      else a_msk3 = msk;
    } else {
      a_off[j] = (unsigned)((size_t)a_m[j] * (size_t)p.lda0 * sizeof(bf16)) + lchunk_b;
      if (ok) a_msk |= 1u << j;
    }
  }
  // weight pieces j = 0..NT-1: tile rows 8 * (NT wave + j) + lr
  unsigned w_off[NT];
  unsigned w_msk = 0;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = n0 + 8 * (NT * wave + j) + lr;
    const bool ok = n < p.N;
    w_off[j] = (unsigned)((size_t)(ok ? n : 0) * (size_t)p.K * sizeof(bf16)) + lchunk_b;
    if (ok) w_msk |= 1u << j;
  }

  const int nk = p.K / G8_BK;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(lptr_t)smem);
  const unsigned lds_a = lds0 + (unsigned)((16 * g + 4 * wn) * 1024);                // + stage * STAGE + j * 1024
  const unsigned lds_w = lds0 + (unsigned)A_BYTES + (unsigned)(NT * wave * 1024);

  // wave-uniform source base of the activation k-tile `kt` (+ which tap it is, TAPI) -- recomputed from kt, a handful of SALU ops
  auto a_base = [&](int kt, int& tap) -> const char* {
    if constexpr (TAPI) {
      const int c64 = kt / 9;
      tap = kt - 9 * c64;
      const int ky = tap / 3, kx = tap - 3 * ky;
      const long long delta = ((long long)(ky - 1) * p.W + (kx - 1)) * p.lda0 * (long long)sizeof(bf16);
      return reinterpret_cast<const char*>(p.a0 + c64 * G8_BK) + delta;
    } else {
      tap = 0;
      const int c = kt * G8_BK;
      return c < p.c0 ? reinterpret_cast<const char*>(p.a0 + c) : reinterpret_cast<const char*>(p.a1 + (c - p.c0));
    }
  };
  // two-source operand (1x1 only): the row offsets depend on the source's pixel stride; switch once, when the k walk crosses c0
  int a_src = 0;
  auto a_select_source = [&](int kt) {
    if constexpr (!TAPI) {
      const int want = (p.c1 > 0 && kt * G8_BK >= p.c0) ? 1 : 0;
      if (want != a_src) {
        a_src = want;
        const int ld = want ? p.lda1 : p.lda0;
#pragma unroll
        for (int j = 0; j < 4; ++j) a_off[j] = (unsigned)((size_t)a_m[j] * (size_t)ld * sizeof(bf16)) + lchunk_b;
      }
    }
  };
  auto issue_a = [&](const char* base, int tap, int stage, int j) {
    bool ok;
    if constexpr (TAPI) ok = ((j < 3 ? (a_msk >> (9 * j)) : a_msk3) >> tap) & 1u;
    else ok = (a_msk >> j) & 1u;
    const char* src = ok ? base + a_off[j] : zsrc;
    glds16(src, lds_a + (unsigned)(stage * STAGE + j * 1024));
  };
  auto issue_w = [&](int kt, int stage, int j) {
    const char* base = reinterpret_cast<const char*>(p.w) + (size_t)kt * (G8_BK * sizeof(bf16));
    const char* src = ((w_msk >> j) & 1u) ? base + w_off[j] : zsrc;
    glds16(src, lds_w + (unsigned)(stage * STAGE + j * 1024));
  };

  f32x4 acc[NT][8];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fg = lane >> 4;
  const unsigned frag0 = (unsigned)(fr * 128 + ((fg ^ (fr & 7)) << 4));      // k-step 0; k-step 1 = frag0 ^ 64
  const unsigned xrow = (unsigned)(g * 128 * 128), wrow = (unsigned)(A_BYTES + wn * WN * 128);

  // ---- prologue: A(0), W(0), W(1) ----
  {
    int tap;
    a_select_source(0);
    const char* ab = a_base(0, tap);
#pragma unroll
    for (int j = 0; j < 4; ++j) issue_a(ab, tap, 0, j);
#pragma unroll
    for (int j = 0; j < NT; ++j) issue_w(0, 0, j);
    if (nk > 1) {
#pragma unroll
      for (int j = 0; j < NT; ++j) issue_w(1, 1, j);
    }
    wait_vm<0>();
    __builtin_amdgcn_s_barrier();
  }
  if (g == 1) __builtin_amdgcn_s_barrier();            // stagger: group 1 runs one interval behind group 0

  // NPH phases per k-tile: 4 (two row tiles per phase: the verified default) or 2 (four row tiles per phase: half the barriers; NT <= 4 only,
  // the activation fragments of a phase then take 32 VGPRs).  Activation pieces go out in the first NPH / 2 phases, weight pieces in the rest.
  constexpr int MPP = 8 / NPH;                         // row tiles per phase
  constexpr int APP = 4 / (NPH / 2);                   // activation pieces per (early) phase
  constexpr int WPP = (NT + NPH / 2 - 1) / (NPH / 2);  // weight pieces per (late) phase (the last one takes the remainder)
  constexpr int W_BEFORE_LAST = WPP * (NPH / 2 - 1);   // weight pieces of t+2 already issued when the W(t+1) wait executes
  bf16x8 wf[2][NT], xf[2][MPP];
  for (int t = 0; t < nk; ++t) {
    const int st = t & 1;
    const char* sbase = smem + st * STAGE;
    const bool has1 = t + 1 < nk, has2 = t + 2 < nk;
    int tap1 = 0;
    const char* ab1 = nullptr;
    if (has1) { a_select_source(t + 1); ab1 = a_base(t + 1, tap1); }
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph) {
      // ---------------- load section ----------------
      if (!(NR_G8P_ABLATE & 2) || t == 0) {
        if (ph == 0) {
#pragma unroll
          for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < NT; ++i) wf[ks][i] = *(const bf16x8*)(sbase + wrow + i * 16 * 128 + (frag0 ^ (unsigned)(ks * 64)));
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int jj = 0; jj < MPP; ++jj) xf[ks][jj] = *(const bf16x8*)(sbase + xrow + (MPP * ph + jj) * 16 * 128 + (frag0 ^ (unsigned)(ks * 64)));
      }
      __builtin_amdgcn_sched_barrier(0);
      if (!(NR_G8P_ABLATE & 1)) {
        if (ph < NPH / 2) {
          if (has1) {
#pragma unroll
            for (int j = 0; j < APP; ++j) issue_a(ab1, tap1, st ^ 1, APP * ph + j);
          }
        } else {
          if (has2) {
#pragma unroll
            for (int j = WPP * (ph - NPH / 2); j < (ph == NPH - 1 ? NT : WPP * (ph - NPH / 2 + 1)); ++j) issue_w(t + 2, st, j);
          }
        }
      }
      if constexpr (NPH == 2) {
        // the weight region of this buffer is re-staged by the OTHER group one interval from now: retire this wave's weight-fragment reads
        // before the barrier (they were issued ahead of the LDS-DMA code above, so this wait is normally free), so that the barrier orders them
        if (ph == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // ---------------- MFMA section ----------------
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int jj = 0; jj < MPP; ++jj)
#pragma unroll
          for (int i = 0; i < NT; ++i) {
#if NR_G8P_ABLATE & 4
            asm volatile("" : : "v"(wf[ks][i]), "v"(xf[ks][jj]));
#else
            acc[i][MPP * ph + jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][i], xf[ks][jj], acc[i][MPP * ph + jj], 0, 0, 0);
#endif
          }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      if (!(NR_G8P_ABLATE & 1)) {
        // counted waits: everything but the N youngest pieces of this wave has landed (vmcnt counts in issue order)
        if (ph == NPH - 2 && has1) { if (has2) wait_vm<4 + W_BEFORE_LAST>(); else wait_vm<4>(); }   // W(t+1) landed; A(t+1) [+ part of W(t+2)] may fly
        if (ph == NPH - 1 && has1) { if (has2) wait_vm<NT>(); else wait_vm<0>(); }                  // A(t+1) landed; W(t+2) may fly
      }
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (g == 0) __builtin_amdgcn_s_barrier();            // re-align the two groups

  // ---- epilogue through LDS (fp32, 16-byte chunks XOR-swizzled with row & 7): per round BOTH groups park RT of their 8 row tiles (so the
  // GEGLU gate / the stores of all eight waves run side by side), then all 512 threads walk the 2 x RT x 16 rows in 16-byte pieces ----
  constexpr int RT = NT <= 4 ? 4 : 2;                  // row tiles per group and round: 2 x RT x 16 x BN fp32 must fit the operand ring
  static_assert((size_t)2 * RT * 16 * BN * sizeof(float) <= (size_t)2 * STAGE, "epilogue tile exceeds the LDS ring");
  float* sC = reinterpret_cast<float*>(smem);
  const int bno = p.geglu ? BN / 2 : BN;
  const int c8n = bno >> 3;
  const int nout = p.geglu ? p.N / 2 : p.N;
  const int nb0 = p.geglu ? n0 / 2 : n0;
#pragma unroll
  for (int r = 0; r < 8 / RT; ++r) {
    __syncthreads();                                   // r = 0: every wave has left the operand ring; r > 0: the previous round was read
#pragma unroll
    for (int jj = 0; jj < RT; ++jj) {
      const int j = r * RT + jj;
      const int row = (g * RT + jj) * 16 + fr;
      if (!p.geglu) {
#pragma unroll
        for (int i = 0; i < NT; ++i) {
          const int c4 = ((wn * WN + i * 16) >> 2) + fg;
          *(f32x4*)(sC + row * BN + ((c4 ^ (row & 7)) << 2)) = acc[i][j];
        }
      } else {
        if constexpr (NT % 2 == 0) {
#pragma unroll
          for (int i = 0; i < NT; i += 2) {
            const int nv = n0 + wn * WN + i * 16 + 4 * fg;
            f32x4 v = acc[i][j], gt = acc[i + 1][j];
            if (p.bias && nv < p.N) { v += *(const f32x4*)(p.bias + nv); gt += *(const f32x4*)(p.bias + nv + 16); }
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = v[e] * gelu_erf_fast(gt[e]);
            const int c4 = (((wn * WN + i * 16) >> 1) >> 2) + fg;
            *(f32x4*)(sC + row * BN + ((c4 ^ (row & 7)) << 2)) = o;
          }
        }
      }
    }
    __syncthreads();
    for (int idx = tid; idx < 2 * RT * 16 * c8n; idx += 512) {
      const int row = idx / c8n, c8 = idx - row * c8n;
      const int gg = row / (RT * 16), lrow = row - gg * (RT * 16);
      const int m = m0 + gg * 128 + r * (RT * 16) + lrow, n = nb0 + c8 * 8;
      if (m >= p.M || n >= nout) continue;
      f32x4 va = *(const f32x4*)(sC + row * BN + (((2 * c8) ^ (row & 7)) << 2));
      f32x4 vb = *(const f32x4*)(sC + row * BN + (((2 * c8 + 1) ^ (row & 7)) << 2));
      if (!p.geglu) {
        if (p.bias) { va += *(const f32x4*)(p.bias + n); vb += *(const f32x4*)(p.bias + n + 4); }
        if (p.rowvec) {
          const float* rv = p.rowvec + rowvec_row(p, m) + n;
          va += *(const f32x4*)rv; vb += *(const f32x4*)(rv + 4);
        }
        va *= p.out_scale; vb *= p.out_scale;
        if (p.act == 1) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { va[e] = quick_gelu_f(va[e]); vb[e] = quick_gelu_f(vb[e]); }
        }
        if (p.res) {
          const bf16x8 rr = *(const bf16x8*)(p.res + (size_t)m * p.ldr + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) { va[e] += (float)rr[e]; vb[e] += (float)rr[4 + e]; }
        }
      }
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) { o[e] = (bf16)va[e]; o[4 + e] = (bf16)vb[e]; }
      nr_store16(p.out + (size_t)m * p.ldo + n, o);
    }
  }
}

inline bool attr_needed(unsigned long long& mask) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (mask & bit) return false;
  mask |= bit;
  return true;
}

template <int NT, bool TAPI, int NPH>
int launch_g8p_n(const NrGemmParams& p, int m_fast, hipStream_t stream) {
  constexpr int BN = 64 * NT;
  constexpr size_t shm = (size_t)2 * (G8_BM + BN) * 128;
  static unsigned long long attr = 0;
  if (attr_needed(attr) &&
      hipFuncSetAttribute((const void*)g8p_kernel<NT, TAPI, NPH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess) return 20;
  const unsigned grid = (unsigned)(((p.M + G8_BM - 1) / G8_BM) * ((p.N + BN - 1) / BN));
  hipLaunchKernelGGL((g8p_kernel<NT, TAPI, NPH>), dim3(grid), dim3(512), shm, stream, p, m_fast);
  return 0;
}
static int g8p_phases = -1;     // NR_G8P_PHASES: 2 (default where the instantiation exists, NT <= 4: never slower in tools/g8p_ab.py, GEGLU 1.08-1.16x
                                // instead of 1.00-1.08x) or 4 phases per k-tile; nr_g8p_set_phases overrides
template <int NT, bool TAPI>
int launch_g8p(const NrGemmParams& p, int m_fast, hipStream_t stream) {
  if (g8p_phases < 0) g8p_phases = getenv("NR_G8P_PHASES") ? atoi(getenv("NR_G8P_PHASES")) : 2;
  if constexpr (NT <= 4) { if (g8p_phases == 2) return launch_g8p_n<NT, TAPI, 2>(p, m_fast, stream); }
  return launch_g8p_n<NT, TAPI, 4>(p, m_fast, stream);
}

}  // namespace

// 0 = not for this kernel; else NT (columns per tile / 64).  Pure function of the launch parameters (and of NR_G8P / NR_G8P_MIN_TILES).
static int g8p_mode = -1;       // NR_G8P: 0 off (A/B), 1 heuristic (default), 2 whenever the shape is supported; nr_g8p_set_mode overrides (tests, A/B tools)
extern "C" void nr_g8p_set_mode(int mode) { g8p_mode = mode; }
extern "C" void nr_g8p_set_phases(int phases) { g8p_phases = phases == 2 ? 2 : 4; }
extern "C" int nr_g8p_plan(const NrGemmParams* pp) {
  const NrGemmParams& p = *pp;
  if (g8p_mode < 0) g8p_mode = getenv("NR_G8P") ? atoi(getenv("NR_G8P")) : 1;
  const int mode = g8p_mode;
  if (!mode) return 0;
  if (p.ln_c || p.out_f32) return 0;
  const int Cin = p.c0 + p.c1;
  if (p.K % 64 != 0 || Cin % 64 != 0 || p.N % 32 != 0 || p.K != p.ksize * p.ksize * Cin) return 0;
  if (p.ksize == 3) { if (!p.tap_inner || p.stride != 1 || p.ups || p.pad_tl0 || p.a1) return 0; }
  else if (p.ksize != 1) return 0;
  if (p.a1 && p.c0 % 64 != 0) return 0;
  if (p.K / 64 < 4) return 0;
  // 32-bit byte offsets inside every source / the weight matrix
  const size_t rows_src = p.ksize == 3 ? (size_t)(p.M / (p.OH * p.OW)) * p.H * p.W : (size_t)p.M;
  if (rows_src * (size_t)p.lda0 * 2 + 4096 >= 0xffffffffull) return 0;
  if (p.a1 && rows_src * (size_t)p.lda1 * 2 + 4096 >= 0xffffffffull) return 0;
  if ((size_t)p.N * p.K * 2 + 4096 >= 0xffffffffull) return 0;
  if (p.lda0 % 8 != 0 || (p.a1 && p.lda1 % 8 != 0) || p.ldo % 8 != 0 || (p.res && p.ldr % 8 != 0)) return 0;
  if (p.rowvec && p.rowvec_ld % 4 != 0) return 0;
  const long long Mp = p.plan_m > 0 && p.plan_m < p.M ? p.plan_m : p.M;     // batch-independent choice (common.h): as for one clip
  const long long ntm = (Mp + G8_BM - 1) / G8_BM;
  const int nk = p.K / 64;
  // Where it pays (tools/g8p_ab.py on MI355X, profiles/r04_g8p_ab_*.txt): long-K launches whose grid fills the chip in whole rounds of 256
  // workgroups -- 1.2-1.3x the tiled igemm on the 3x3 convs of the SparseCtrl groups / configs 4 / 5 / the VAE (1.22-1.34 PFLOP/s).  With
  // one workgroup per CU the prologue (first tiles from HBM) and the epilogue are exposed once per tile, so short-K Linears (K = 640: 10
  // k-tiles) only win on very large grids, narrow tiles (N = 128: 8 MFMAs per phase) lose to the load section, and under-filled grids lose
  // to the tiled kernel's 2 x 256 slots.
  static const int min_tiles = getenv("NR_G8P_MIN_TILES") ? atoi(getenv("NR_G8P_MIN_TILES")) : 200;
  int best = 0;
  double best_score = 0.0;
  for (int nt = 5; nt >= 2; --nt) {
    if (p.geglu && nt % 2) continue;
    if (mode != 2 && nt < 4) continue;
    const int bn = 64 * nt;
    const long long ntn = (p.N + bn - 1) / bn;
    const long long tiles = ntm * ntn;
    if (mode != 2 && tiles < (nk >= 16 ? min_tiles : 512)) continue;
    const double fill = (double)tiles / (double)(((tiles + 255) / 256) * 256);            // wave quantisation
    const double used = (double)p.N / (double)(ntn * bn) * (double)Mp / (double)(ntm * G8_BM);   // padded outputs
    const double width = nt >= 4 ? 1.0 : (nt == 3 ? 0.9 : 0.8);                         // narrower tiles stage more bytes per FLOP
    const double score = fill * used * width;
    if (score > best_score) { best_score = score; best = nt; }
  }
  if (mode != 2 && best_score < 0.70) return 0;
  // GEGLU projections (gate evaluated in the epilogue, all eight waves side by side): 1.07-1.16x on the two-phase schedule from 512 tiles
  // with K >= 1024 or M >= 32768; the B = 1 U-Net's M = 8192 case loses (0.96x)
  static const int geglu_ok = getenv("NR_G8P_GEGLU") ? atoi(getenv("NR_G8P_GEGLU")) : 1;
  if (mode != 2 && p.geglu) {
    const long long tiles = best ? ntm * ((p.N + 64 * best - 1) / (64 * best)) : 0;
    if (!geglu_ok || tiles < 512 || !(nk >= 16 || Mp >= 32768)) return 0;
  }
  return best;
}

extern "C" int nr_launch_g8p(const NrGemmParams* pp, int m_fast, hipStream_t stream) {
  const NrGemmParams& p = *pp;
  const int nt = nr_g8p_plan(pp);
  if (!nt) return 21;
  const bool tapi = p.ksize == 3;
  switch (nt) {
    case 2: return tapi ? launch_g8p<2, true>(p, m_fast, stream) : launch_g8p<2, false>(p, m_fast, stream);
    case 3: return tapi ? launch_g8p<3, true>(p, m_fast, stream) : launch_g8p<3, false>(p, m_fast, stream);
    case 4: return tapi ? launch_g8p<4, true>(p, m_fast, stream) : launch_g8p<4, false>(p, m_fast, stream);
    case 5: return tapi ? launch_g8p<5, true>(p, m_fast, stream) : launch_g8p<5, false>(p, m_fast, stream);
  }
  return 22;
}
