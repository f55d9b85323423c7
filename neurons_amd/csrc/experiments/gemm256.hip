// 256-row implicit-GEMM tile for the long-K convolutions and Linears of the denoisers (same operator as gemm.hip: 3x3 / 1x1 convs as an
// implicit im2col GEMM on v_mfma_f32_16x16x32_bf16, InflatedConv3d animatediff/models/resnet.py:10-18, Upsample3D :32-80, Downsample3D
// :83-106, skip concat unet_blocks.py:634,740; nn.Linear of the transformers).
//
// Why a second kernel.  The 128-wide tiles of gemm.hip are bound by the L2 -> LDS fill rate ((BM + BN) * 128 B per BM * BN * 128 FLOP:
// 1.15 / 1.28 PFLOP/s caps for 128 x 128 / 128 x 160, which the long-K convs reach); a 256 x 128 / 256 x 160 tile halves the fill per
// FLOP, but only ONE such workgroup fits a CU (3 x 48-52 KiB of LDS ring), so the overlap of loads and MFMAs that two co-resident
// 128-wide workgroups give each other for free has to come from INSIDE the workgroup.  Structure:
//   * 512 threads = 8 waves; waves w and w + 4 share a SIMD.  Wave w owns the 64 x (BN / 2) output sub-tile (rows 64 (w & 3), column half
//     w >> 2): 4 x (BN / 32) accumulator tiles.
//   * the two wave GROUPS (w < 4, w >= 4) alternate roles phase by phase, one raw s_barrier per phase: in an MFMA phase a wave issues the
//     32 / 40 MFMAs of one k-tile from fragments that are ALREADY in its registers (no LDS wait inside) with its share of the LDS-DMA of
//     the tile two ahead interleaved; in the other phase it reads the next k-tile's 16 / 18 fragments from LDS.  Every SIMD therefore
//     always has one wave in the matrix pipe and one on the LDS / DMA side (MI355X_MICROARCH.md, "Two waves per SIMD").
//   * LDS-DMA from inline asm with counted vmcnt (tiles really stay in flight across the barriers), 3-stage ring, XOR-swizzled [rows][64]
//     images, im2col / zero-padding / tails in the per-lane source address, running source pointers: all as in gemm.hip.
//   Group 0 computes k-tile t in phase 2t and reads the fragments of t + 1 in phase 2t + 1; group 1 is one phase behind.  Tile t + 2 is
//   issued in phase 2t by every wave (group 0 between its MFMAs, group 1 behind its fragment reads) and must have landed by the barrier
//   in front of phase 2t + 3 (group 0's fragment read): every wave waits for ITS pieces at the end of phase 2t + 2
//   (vmcnt(pieces issued in that phase)); the slot it overwrites held tile t - 1, last read in phase 2t - 2.
// STATUS (round 3, profiles/r03_gemm256_ab.txt, r03_gemm256_ablation.txt): REJECTED, not part of the product library (built only by
// `make -C neurons_amd/csrc experiments`; NR_IGEMM256=2 routes every eligible launch here for the A/B).  Bit-compatible with gemm.hip
// (same sums up to split-K order) but SLOWER.  On the long-K 3x3 convs the 128-row
// kernel runs 950-1,050 TFLOP/s; this one 600-770 (per occupied CU about equal, 1.0-1.17 PFLOP/s-equivalent; with 128-wide tiles the
// U-Net's grids are 80-384 tiles = 0.6-0.75 of the CUs, and the 160-wide instantiation that would make them 64-256 spills 18-58
// VGPRs).  The ablation build said why the per-CU rate does not beat two co-resident 128-wide workgroups: with no DMA and no fragment
// reads at all the alternating phases still take 1.8-2x the MFMA time (one wave per SIMD issuing, one barrier per 32-MFMA phase), and the
// six LDS-DMA pieces a wave issues per phase cost about as much as its 32 MFMAs (MI355X_MICROARCH.md: 60-185 cycles per piece).
// Epilogue options: bias, fp32 row vector (time embedding), scale, quick_gelu, residual, deterministic split-K slabs (reduce kernel of
// gemm.hip).  No LayerNorm fold and no GEGLU here: those launches stay on gemm.hip / rowpanel.hip / ffpanel.hip.
#include "common.h"
#include <cstdlib>

namespace {

__device__ __attribute__((aligned(16))) const unsigned int nr256_zero16[4] = {0u, 0u, 0u, 0u};
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void glds16_asm(const void* src, unsigned lds_wave_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(lds_wave_base) : "memory", "m0");
}
__device__ __forceinline__ void wait_vmcnt_dyn(int n) {       // wave-uniform n in {0, 6, 7}
  if (n >= 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  else if (n == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ size_t rowvec_row(const NrGemmParams& p, int m) {
  int r = m / p.rowvec_div;
  if (p.rowvec_mod) r %= p.rowvec_mod;
  return (size_t)r * p.rowvec_ld;
}

template <int BN, bool TI>     // TI: tap-inner 3x3 (NrGemmParams::tap_inner), else the generic im2col / 1x1 path
__global__ __launch_bounds__(512) void igemm256_kernel(NrGemmParams p, int splitk, float* partial, int m_fast) {
  constexpr int BM = 256, BK = 64, NS = 3;
  constexpr int MT = 4, NT = BN / 32;                 // accumulator tiles of a wave: 64 rows x BN / 2 columns
  constexpr int GA = 4;                               // 8-row groups of the A tile a wave stages
  constexpr int GBMAX = BN == 160 ? 3 : 2;            // ... of the B tile (BN = 160: waves 0-3 stage three groups, waves 4-7 two)
  constexpr int TILE = (BM + BN) * BK;                // elements per LDS stage
  extern __shared__ __attribute__((aligned(16))) bf16 smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;                          // role group; also the column half of the wave's output sub-tile
  const int wm = wave & 3;
  const int ntn = (p.N + BN - 1) / BN;
  const int ntm = (p.M + BM - 1) / BM;
  int bid;
  {   // XCD-aware remap (bijective), as gemm.hip
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7, local = orig >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
  }
  const int slice = bid / (ntn * ntm);
  bid -= slice * ntn * ntm;
  int bm, bn;
  if (m_fast >= 2) {
    const int G = m_fast;
    const int band = bid / (G * ntn);
    const int first = band * G;
    const int gsz = min(G, ntm - first);
    const int r = bid - band * G * ntn;
    bm = first + r % gsz;
    bn = r / gsz;
  } else if (m_fast) { bn = bid / ntm; bm = bid - bn * ntm; } else { bm = bid / ntn; bn = bid - bm * ntn; }
  const int m0 = bm * BM, n0 = bn * BN;
  const int lr = lane >> 3, lp = lane & 7;
  const int lchunk = (lp ^ lr) << 3;
  const int Cin = p.c0 + p.c1;

  // ---- staging roles: A groups 4 wave .. 4 wave + 3; B groups (BN = 128) 2 wave, 2 wave + 1, (BN = 160) 3 per wave of group 0, 2 of group 1 ----
  const int gb_n = (BN == 160 && grp == 0) ? 3 : 2;
  const int gb_base = BN == 160 ? (grp == 0 ? 3 * wave : 12 + 2 * (wave - 4)) : 2 * wave;
  const int pieces = GA + gb_n;                       // LDS-DMA instructions of this wave per k-tile (6 or 7)

  int a_pix[GA], a_oy[GA], a_ox[GA];
  bool a_ok[GA];
#pragma unroll
  for (int j = 0; j < GA; ++j) {
    const int m = m0 + 8 * (wave * GA + j) + lr;
    a_ok[j] = m < p.M;
    const int mm = a_ok[j] ? m : 0;
    if (p.ksize == 1) {
      a_pix[j] = mm; a_oy[j] = 0; a_ox[j] = 0;
    } else {
      const int ohw = p.OH * p.OW;
      const int n = mm / ohw;
      const int r = mm - n * ohw;
      a_pix[j] = n; a_oy[j] = r / p.OW; a_ox[j] = r - a_oy[j] * p.OW;
    }
  }
  const bf16* zsrc = (const bf16*)nr256_zero16;
  constexpr bool tap_inner = TI;
  long long a_cbyte[GA];
  unsigned a_tapmask[GA];
  if constexpr (TI) {
#pragma unroll
    for (int j = 0; j < GA; ++j) {
      a_cbyte[j] = (((long long)a_pix[j] * p.H + a_oy[j]) * p.W + a_ox[j]) * p.lda0 * (long long)sizeof(bf16);
      unsigned msk = 0;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int iy = a_oy[j] + t / 3 - 1, ix = a_ox[j] + t % 3 - 1;
        if (a_ok[j] && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) msk |= 1u << t;
      }
      a_tapmask[j] = msk;
    }
  }
  const int nk_total = p.K / BK;
  const int kt_begin = (int)(((long long)nk_total * slice) / splitk);
  const int kt_end = (int)(((long long)nk_total * (slice + 1)) / splitk);
  const int nk = kt_end - kt_begin;

  // ---- running source pointers of the next k-tile to stage (gemm.hip) ----
  const bf16* ap[GA];
  int ainc[GA];
  const bf16* wp[GBMAX];
  int winc[GBMAX];
  int st_tap, st_c;
  {
    const int kbase = kt_begin * BK;
    if constexpr (TI) { st_tap = kt_begin % 9; st_c = (kt_begin / 9) * BK; }
    else { st_tap = p.ksize == 3 ? kbase / Cin : 0; st_c = kbase - st_tap * Cin; }
#pragma unroll
    for (int j = 0; j < GBMAX; ++j) {
      const int n = n0 + 8 * (gb_base + j) + lr;
      const bool ok = j < gb_n && n < p.N;
      wp[j] = ok ? p.w + (size_t)n * p.K + kbase + lchunk : zsrc;
      winc[j] = ok ? BK : 0;
    }
  }
  auto setup_rows_tap_inner = [&]() {
    const int ky = st_tap / 3, kx = st_tap - ky * 3;
    const long long delta = ((long long)(ky - 1) * p.W + (kx - 1)) * p.lda0 * (long long)sizeof(bf16);
    const char* base = reinterpret_cast<const char*>(p.a0 + st_c + lchunk) + delta;
#pragma unroll
    for (int j = 0; j < GA; ++j) {
      const bool ok = (a_tapmask[j] >> st_tap) & 1u;
      ap[j] = ok ? reinterpret_cast<const bf16*>(base + a_cbyte[j]) : zsrc;
    }
  };
  auto setup_rows = [&]() {
    if constexpr (TI) { setup_rows_tap_inner(); return; }
    else {
    const bf16* src; int ld;
    if (st_c < p.c0) { src = p.a0 + st_c; ld = p.lda0; } else { src = p.a1 + (st_c - p.c0); ld = p.lda1; }
    src += lchunk;
    if (p.ksize == 1) {
#pragma unroll
      for (int j = 0; j < GA; ++j) {
        ap[j] = a_ok[j] ? src + (size_t)a_pix[j] * ld : zsrc;
        ainc[j] = a_ok[j] ? BK : 0;
      }
    } else {
      const int ky = st_tap / 3, kx = st_tap - ky * 3;
      const int VH = p.ups ? p.H * 2 : p.H, VW = p.ups ? p.W * 2 : p.W;
#pragma unroll
      for (int j = 0; j < GA; ++j) {
        int iy = a_oy[j] * p.stride + ky - (p.pad_tl0 ? 0 : 1);
        int ix = a_ox[j] * p.stride + kx - (p.pad_tl0 ? 0 : 1);
        const bool ok = a_ok[j] && iy >= 0 && iy < VH && ix >= 0 && ix < VW;
        if (p.ups) { iy >>= 1; ix >>= 1; }
        const size_t pix = ((size_t)a_pix[j] * p.H + iy) * p.W + ix;
        ap[j] = ok ? src + pix * ld : zsrc;
        ainc[j] = ok ? BK : 0;
      }
    }
    }
  };
  setup_rows();
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(lptr_t)smem);
  const unsigned la_off = (unsigned)(wave * GA * 8 * BK * (int)sizeof(bf16));
  const unsigned lb_off = (unsigned)((BM * BK + gb_base * 8 * BK) * (int)sizeof(bf16));
  // piece i (0 .. pieces-1) of the tile the running pointers stand on, into ring slot `buf`
  auto issue_piece = [&](int buf, int i) {
    const unsigned sbase = lds_base + (unsigned)(buf * TILE * (int)sizeof(bf16));
    if (i < GA) glds16_asm(ap[i], sbase + la_off + (unsigned)(i * 8 * BK * (int)sizeof(bf16)));
    else glds16_asm(wp[i - GA], sbase + lb_off + (unsigned)((i - GA) * 8 * BK * (int)sizeof(bf16)));
  };
  auto advance = [&]() {                                // running pointers -> next k-tile
#pragma unroll
    for (int j = 0; j < GBMAX; ++j) wp[j] += winc[j];
    if constexpr (TI) {
      st_tap += 1;
      if (st_tap == 9) { st_tap = 0; st_c += BK; }
      setup_rows_tap_inner();
      return;
    } else {
    st_c += BK;
    bool resetup = false;
    if (st_c == Cin) { st_c = 0; st_tap += 1; resetup = true; }
    else if (p.c1 > 0 && st_c == p.c0) resetup = true;
    if (resetup) {
      if (st_tap < p.ksize * p.ksize) setup_rows();
    } else {
#pragma unroll
      for (int j = 0; j < GA; ++j) ap[j] += ainc[j];
    }
    }
  };
  auto issue_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < GA + GBMAX; ++i) if (i < pieces) issue_piece(buf, i);
    advance();
  };

  f32x4 acc[NT][MT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fg = lane >> 4;
  bf16x8 wf[2][NT], xf[2][MT];
  auto read_frags = [&](int buf) {
    const bf16* sA = smem + buf * TILE;
    const bf16* sB = sA + BM * BK;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const int row = grp * (BN / 2) + i * 16 + fr;
        wf[ks][i] = *(const bf16x8*)(sB + row * BK + (((4 * ks + fg) ^ (row & 7)) << 3));
      }
#pragma unroll
      for (int j = 0; j < MT; ++j) {
        const int row = wm * 64 + j * 16 + fr;
        xf[ks][j] = *(const bf16x8*)(sA + row * BK + (((4 * ks + fg) ^ (row & 7)) << 3));
      }
    }
  };

  // ---- prologue: tiles 0 and 1 in flight; tile 0 landed; group 0 holds its fragments ----
  if (nk > 0) issue_tile(0);
  if (nk > 1) issue_tile(1);
  wait_vmcnt_dyn(nk > 1 ? pieces : 0);
  __builtin_amdgcn_s_barrier();
  if (grp == 0 && nk > 0) {
    read_frags(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_sched_barrier(0);

  // One straight loop per role group (wave-uniform branch): the fragments are defined and consumed without passing through a merge of the
  // two roles, so no register copies.  Both groups issue the DMA of tile kt + 2 in phase 2 kt (group 0 between its MFMAs, group 1 behind
  // its fragment reads) and retire their pieces of tile kt + 1 at the end of that phase: 2.5 phases of flight time for every piece.
  auto mfma_phase = [&](bool pf, int pbuf) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < NT; ++i) {
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][i], xf[ks][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        const int slot = ks * NT + i;                   // one piece behind each of the first `pieces` 4-MFMA bursts
        if (pf && slot < GA + GBMAX && slot < pieces) issue_piece(pbuf, slot);
        __builtin_amdgcn_sched_barrier(0);
      }
    __builtin_amdgcn_s_setprio(0);
  };
  if (grp == 0) {
    for (int kt = 0; kt < nk; ++kt) {
      const bool pf = kt + 2 < nk;
      mfma_phase(pf, (kt + 2) % NS);                    // phase 2 kt
      if (pf) advance();
      __builtin_amdgcn_sched_barrier(0);
      wait_vmcnt_dyn(pf ? pieces : 0);
      __builtin_amdgcn_s_barrier();
      read_frags((kt + 1) % NS);                        // phase 2 kt + 1 (last iteration: a read nobody uses)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
    }
  } else {
    for (int kt = 0; kt < nk; ++kt) {
      const bool pf = kt + 2 < nk;
      read_frags(kt % NS);                              // phase 2 kt
      if (pf) {
#pragma unroll
        for (int i = 0; i < GA + GBMAX; ++i) if (i < pieces) issue_piece((kt + 2) % NS, i);
        advance();
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      wait_vmcnt_dyn(pf ? pieces : 0);
      __builtin_amdgcn_s_barrier();
      mfma_phase(false, 0);                             // phase 2 kt + 1
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
    }
  }

  // ---- epilogue: lane holds out[m = .. + fr][n = .. + 4 fg + r] ----
  if (partial) {
    float* slab = partial + (size_t)slice * p.M * p.N;
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      const int m = m0 + wm * 64 + j * 16 + fr;
      if (m >= p.M) continue;
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const int n = n0 + grp * (BN / 2) + i * 16 + 4 * fg;
        if (n >= p.N) continue;
        *(f32x4*)(slab + (size_t)m * p.N + n) = acc[i][j];
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < MT; ++j) {
    const int m = m0 + wm * 64 + j * 16 + fr;
    if (m >= p.M) continue;
    const float* rv = p.rowvec ? p.rowvec + rowvec_row(p, m) : nullptr;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int n = n0 + grp * (BN / 2) + i * 16 + 4 * fg;
      if (n >= p.N) continue;
      f32x4 v = acc[i][j];
      if (p.bias) { const f32x4 b = *(const f32x4*)(p.bias + n); v += b; }
      if (rv) { const f32x4 t = *(const f32x4*)(rv + n); v += t; }
      v *= p.out_scale;
      if (p.act == 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = quick_gelu_f(v[e]);
      }
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
  }
}

unsigned long long g_attr256 = 0;

}  // namespace

// shapes this kernel serves; bn_out / splitk_out: its tile width and K split
extern "C" int nr_igemm256_plan(const NrGemmParams* pp, int* bn_out, int* splitk_out) {
  const NrGemmParams& p = *pp;
  const int mode = getenv("NR_IGEMM256") ? atoi(getenv("NR_IGEMM256")) : 0;     // 0 off (default: measured slower, see header), 1 heuristic, 2 every eligible launch
  if (!mode) return 0;
  const int Cin = p.c0 + p.c1;
  if (p.ln_c || p.geglu || p.out_f32) return 0;
  if (p.K % 64 != 0 || Cin % 64 != 0 || (p.a1 && p.c0 % 64 != 0) || p.K != p.ksize * p.ksize * Cin) return 0;
  if (p.tap_inner && (p.ksize != 3 || p.stride != 1 || p.ups || p.pad_tl0 || p.a1)) return 0;
  // 128-wide tiles (no spills); N = 320 takes three of them (the third half empty).  The 160-wide instantiation (two full tiles for N = 320)
  // spills 18-27 VGPRs and is only reachable through NR_IGEMM256_BN=160 for comparison.
  const int bn_force = getenv("NR_IGEMM256_BN") ? atoi(getenv("NR_IGEMM256_BN")) : 0;
  int bn = 128;
  if (bn_force == 160 && p.N % 160 == 0) bn = 160;
  if (p.N % 32 != 0) return 0;
  const int nk = p.K / 64;
  if (mode == 1 && (p.M < 2048 || nk < 16)) return 0;          // short-K / small-M launches stay on the 128-wide tiles
  const long long tiles = (long long)((p.M + 255) / 256) * ((p.N + bn - 1) / bn);
  int sk = 1;
  if (tiles < 200) {                                           // fill the 256 CUs with K slices (deterministic slab reduce), >= 12 k-tiles each
    sk = (int)((256 + tiles - 1) / tiles);
    if (sk > nk / 12) sk = nk / 12;
    if (sk < 1) sk = 1;
    if (sk > 16) sk = 16;
  }
  if (const char* f = getenv("NR_IGEMM256_SPLITK")) sk = atoi(f) > 0 ? atoi(f) : sk;
  if (bn_out) *bn_out = bn;
  if (splitk_out) *splitk_out = sk;
  return 1;
}

extern "C" size_t nr_igemm256_workspace_bytes(const NrGemmParams* pp) {
  int bn, sk;
  if (!nr_igemm256_plan(pp, &bn, &sk)) return 0;
  return sk > 1 ? (size_t)sk * pp->M * pp->N * sizeof(float) : 0;
}

// returns 0 on success; the caller runs the split-K reduce kernel of gemm.hip when *splitk_used > 1
extern "C" int nr_launch_igemm256(const NrGemmParams* pp, float* workspace, int m_fast, int* splitk_used, hipStream_t stream) {
  int bn, sk;
  if (!nr_igemm256_plan(pp, &bn, &sk)) return 1;
  if (sk > 1 && !workspace) return 6;
  const NrGemmParams& p = *pp;
  const unsigned grid = (unsigned)(((p.M + 255) / 256) * ((p.N + bn - 1) / bn) * sk);
  float* partial = sk > 1 ? workspace : nullptr;
  int dev = 0;
  (void)hipGetDevice(&dev);
  typedef void (*kern_t)(NrGemmParams, int, float*, int);
  static const kern_t ks[4] = {igemm256_kernel<128, false>, igemm256_kernel<128, true>, igemm256_kernel<160, false>, igemm256_kernel<160, true>};
  if (!(g_attr256 >> (dev & 63) & 1ull)) {
    for (int i = 0; i < 4; ++i)
      (void)hipFuncSetAttribute((const void*)ks[i], hipFuncAttributeMaxDynamicSharedMemorySize, 3 * (256 + (i < 2 ? 128 : 160)) * 64 * 2);
    g_attr256 |= 1ull << (dev & 63);
  }
  const size_t shm = (size_t)3 * (256 + bn) * 64 * 2;
  hipLaunchKernelGGL(ks[(bn == 160 ? 2 : 0) + (p.tap_inner ? 1 : 0)], dim3(grid), dim3(512), shm, stream, p, sk, partial, m_fast);
  if (splitk_used) *splitk_used = sk;
  return 0;
}
