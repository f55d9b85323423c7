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
// Structure (per 256-thread workgroup = 2x2 waves, BMxBN output tile, BK = 64):
//   * global -> LDS by LDS-DMA (global_load_lds_dwordx4: 1 KiB = 8 tile rows per wave-instruction, no
//     VGPR staging).  The im2col gather, the zero padding and the M/N tails are all expressed through the
//     per-lane SOURCE address (invalid lanes read a 16-byte zero word), the LDS image stays lane-linear.
//   * LDS image [rows][64] bf16 (128-B rows); the 16-B chunk index is XOR-swizzled with (row & 7) on the
//     source side and on the ds_read_b128 side (same involution) -> conflict-free fragment reads.
//   * two LDS buffers, ONE barrier per k-tile: tile k+1 streams in while tile k is multiplied.
//   * the MFMA is issued "transposed" (weights = A operand, activations = B operand) so a lane's 4
//     accumulator registers are 4 consecutive output channels of one pixel -> 8/16-byte epilogue accesses.
//   * split-K for small-M / huge-K layers (the 4x4 and 8x8 levels: M = 512..2048, K up to 23 040):
//     each slice writes an fp32 slab, a second kernel sums the slabs in fixed order (deterministic, no
//     atomics) and applies the epilogue.
#include "common.h"
#include <cstdio>
#include <cstdlib>

namespace {

__device__ __attribute__((aligned(16))) const unsigned int nr_zero16[4] = {0u, 0u, 0u, 0u};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// LDS-DMA as inline asm (glds16_asm), hidden from the compiler: a BUILTIN global_load_lds is a pending LDS write to hipcc, which then places
// s_waitcnt vmcnt(0) in front of the next ds_read that may alias it, i.e. behind every barrier of the main loop, so a ring deeper
// than two stages never actually has more than one tile in flight.  With the asm form only the counted wait + barrier of the main
// loop order the DMA against the fragment reads (cdna_hip_programming.md 5.7); M0 is written and restored in the same statement.
// M0 (the LDS destination) is written in the same statement that reads it and declared clobbered, so nothing is saved or
// restored per transfer (hipcc only warns that m0 is a reserved register; it keeps no value in it across the statement).
__device__ __forceinline__ void glds16_asm(const void* src, unsigned lds_wave_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(lds_wave_base) : "memory", "m0");
}
__device__ __forceinline__ void glds16_builtin(const void* src, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(lptr_t)p);
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N <= 63) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__device__ __forceinline__ size_t rowvec_row(const NrGemmParams& p, int m) {
  int r = m / p.rowvec_div;
  if (p.rowvec_mod) r %= p.rowvec_mod;
  return (size_t)r * p.rowvec_ld;
}

// WGM x WGN = wave grid over the (M, N) tile; 64*WGM*WGN threads
// ADMA: LDS-DMA issued from inline asm (tiles really stay in flight across the barrier; pays for long K) instead of the builtin
// (the compiler then drains the DMA in front of the next fragment read: DMA and MFMA of a k-tile do not overlap, but its M0
// handling is cheaper: measured faster for the short-K Linears of this workload).
// a / d for 0 <= a < 2^22, d > 0 through the float reciprocal with one correction step (exact in that range).  The tile-index and
// im2col-row arithmetic of the prologue used ~10 integer (two of them 64-bit) divisions = 2,300-3,100 of the 4,400-4,900 cycles between
// kernel entry and the first LDS-DMA (in-kernel stamps, profiles/r03_igemm_timeline_smallm.txt): per WORKGROUP, i.e. once per tile.
__device__ __forceinline__ int fdiv_small(int a, int d) {
  int q = (int)((float)a * __builtin_amdgcn_rcpf((float)d));
  const int r = a - q * d;
  q += (r >= d ? 1 : 0) - (r < 0 ? 1 : 0);
  return q;
}

#ifndef NR_ABLATE
#define NR_ABLATE 0      // timing ablations of the k-loop (experiments build only, results wrong): 2 no LDS-DMA in the loop, 4 fragment reads in the
#endif                   // first iteration only, 8 no MFMAs (tools/conv_halo_potential.py)
#ifdef NR_STAMP
// Diagnostic build only (make stamp -> libneurons_amd_stamp.so, tools/igemm_timeline.py): shader-clock stamps of wave 0 of the first
// 512 workgroups.  The stamps go to a buffer of their own; no output value depends on them.
#define NR_STAMP_SLOTS 48
__device__ unsigned long long nr_stamp_buf[512][NR_STAMP_SLOTS];
#define NR_STAMP_AT(slot) do { if (threadIdx.x == 0 && blockIdx.x < 512 && (slot) < NR_STAMP_SLOTS) nr_stamp_buf[blockIdx.x][(slot)] = __builtin_amdgcn_s_memtime(); } while (0)
// the chip-wide 100 MHz counter (s_memtime counters are not synchronised across the chip): entry spread / kernel span over all workgroups
#define NR_STAMP_RT(slot) do { if (threadIdx.x == 0 && blockIdx.x < 512) nr_stamp_buf[blockIdx.x][(slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define NR_STAMP_AT(slot) do { } while (0)
#define NR_STAMP_RT(slot) do { } while (0)
#endif

// LIN: the launch is a plain Linear (1x1, one source): the im2col / tap / two-source paths are compiled out.  Same arithmetic; what it buys is
// code size: a launch starts with a cold instruction cache (tools/icache_probe.py: +0.4-1.4 us per launch when instantiations alternate, as
// they do in the engine's graphs), and two thirds of the launches of a denoising step are Linears.  LayerNorm-folded launches are always Linears.
template <int BM, int BN, int NS, int WGM, int WGN, bool LNF = false, bool ADMA = false, bool LIN = LNF>
__global__ __launch_bounds__(64 * WGM * WGN, (LNF && WGM * WGN == 8) ? 4 : 1) void igemm_bf16_kernel(NrGemmParams p_arg, int splitk, float* partial, int m_fast) {
  NrGemmParams p = nr_pin_params(p_arg);
  if constexpr (LIN) { p.ksize = 1; p.stride = 1; p.ups = 0; p.a1 = nullptr; p.c1 = 0; p.tap_inner = 0; p.pad_tl0 = 0; }
  splitk = nr_pin(splitk); partial = nr_pin(partial); m_fast = nr_pin(m_fast);
  constexpr int BK = 64;
  constexpr int NW = WGM * WGN;
  constexpr int WM = BM / WGM, WN = BN / WGN;
  constexpr int MT = WM / 16, NT = WN / 16;
  constexpr int GA = BM / (8 * NW), GB = BN / (8 * NW);   // 8-row groups per wave for the A / B tile
  static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile rows must split evenly over the waves");
  constexpr int TILE = (BM + BN) * BK;               // elements per LDS buffer
  extern __shared__ __attribute__((aligned(16))) bf16 smem[];   // NS * TILE elements (ring of NS k-tiles)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  NR_STAMP_AT(0);
  NR_STAMP_RT(44);
  const int wm = wave / WGN, wn = wave % WGN;
  const int ntn = (p.N + BN - 1) / BN;
  const int ntm = (p.M + BM - 1) / BM;
  // XCD-aware remap (bijective): workgroups are dealt round-robin over the 8 XCDs, so give each XCD a
  // CONTIGUOUS range of logical tiles: the n-tiles that share one A row-panel then hit the same L2
  // instead of re-fetching the panel from HBM once per XCD.  Placement only affects speed.
  int bid;
  {
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7, local = orig >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
  }
  // splitk < 0: |splitk| K-slices with the IN-LAUNCH reduction (round 6, experiment NR_SPLITK_L2=1): the slices of one output tile are
  // consecutive logical ids, i.e. (the XCD ranges above being contiguous and a multiple of |splitk| long: the launcher checks) they run on ONE
  // XCD, their fp32 slabs meet in that XCD's L2 and the last arriver sums them and runs the epilogue -- no second launch, no cache-wide fence
  // MEASURED SLOWER (0.43-1.00 x, profiles/r06_splitk_xcd_ab.txt: the last arriver reads its tile's slabs alone): experiments library only.
#ifdef NR_EXPERIMENTS
  const bool l2red = splitk < 0;
  if (l2red) splitk = -splitk;
  int slice;
  if (l2red) { const int t = fdiv_small(bid, splitk); slice = bid - t * splitk; bid = t; }
  else { slice = splitk > 1 ? fdiv_small(bid, ntn * ntm) : 0; bid -= slice * ntn * ntm; }
  const int tile_id = bid;
#else
  const int slice = splitk > 1 ? fdiv_small(bid, ntn * ntm) : 0;
  bid -= slice * ntn * ntm;
#endif
  // tile order inside an XCD's range: the operand that is re-used by neighbouring tiles should be the BIG
  // one.  m_fast: neighbours share a weight panel (weight-heavy 4x4 / 8x8 levels); else an activation panel.
  int bm, bn;
  if (m_fast >= 2) {
    // grouped order (m_fast = G >= 2): bands of G m-tiles, inside a band the m index runs fastest, then n.  The tiles that are in
    // flight together on an XCD (2 per CU) then form a G x (64 / G) block: both operand panels of the block fit its 4 MiB L2, where
    // a plain n-fastest walk streams the whole weight matrix past the L2 once per m-tile row (W > L2: GEGLU / q|k|v at 16x16, 8x8)
    const int G = m_fast;
    const int band = fdiv_small(bid, G * ntn);
    const int first = band * G;
    const int gsz = min(G, ntm - first);
    const int r = bid - band * G * ntn;
    bn = fdiv_small(r, gsz);
    bm = first + r - bn * gsz;
  } else if (m_fast) { bn = fdiv_small(bid, ntm); bm = bid - bn * ntm; } else { bm = fdiv_small(bid, ntn); bn = bid - bm * ntn; }
  const int m0 = bm * BM, n0 = bn * BN;
  NR_STAMP_AT(40);
  const int lr = lane >> 3;                 // row within the 8-row group
  const int lp = lane & 7;                  // physical 16-B chunk
  const int lchunk = (lp ^ lr) << 3;        // logical chunk (elements) this lane must fetch

  const int Cin = p.c0 + p.c1;

  // ---- per-lane A rows: group g = wave*GA + j, tile row = 8*g + lr ----
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
      const int n = fdiv_small(mm, ohw);
      const int r = mm - n * ohw;
      a_pix[j] = n; a_oy[j] = fdiv_small(r, p.OW); a_ox[j] = r - a_oy[j] * p.OW;
    }
  }
  const bf16* zsrc = (const bf16*)nr_zero16;
  // tap-inner K order (p.tap_inner): per row the byte offset of the centre pixel and a 9-bit mask of the taps that stay inside the image
  // (compiled out of the LayerNorm-fused instantiations, which only ever run 1x1: their 8-wave tiles must stay within 128 VGPRs to keep
  // two workgroups per CU)
  const bool tap_inner = !LNF && p.tap_inner != 0;
  long long a_cbyte[GA];
  unsigned a_tapmask[GA];
  if (tap_inner) {
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

  NR_STAMP_AT(41);
  const int nk_total = p.K / BK;
  int kt_begin = 0, kt_end = nk_total;
  if (splitk > 1) { kt_begin = fdiv_small(nk_total * slice, splitk); kt_end = fdiv_small(nk_total * (slice + 1), splitk); }

  // ---- running source pointers of the NEXT k-tile to stage.  The k index walks (tap, channel) with the
  // channel fastest, so between two k-tiles every pointer simply advances by 64 elements; the im2col
  // address arithmetic (border test, pixel index, 64-bit multiply) is redone only when the tap or the concat
  // source changes (once per Cin/64 tiles).  Rows that are padding / out of range sit on a 16-byte zero word
  // with increment 0.  This keeps the main loop at ~2 VALU per LDS-DMA instead of ~15. ----
  const bf16* ap[GA];
  int ainc[GA];
  const bf16* wp[GB];
  int winc[GB];
  int st_tap, st_c;
  {
    const int kbase = kt_begin * BK;
    if (tap_inner) { const int q9 = fdiv_small(kt_begin, 9); st_tap = kt_begin - 9 * q9; st_c = q9 * BK; }
    else { st_tap = p.ksize == 3 ? fdiv_small(kbase, Cin) : 0; st_c = kbase - st_tap * Cin; }
#pragma unroll
    for (int j = 0; j < GB; ++j) {
      const int n = n0 + 8 * (wave * GB + j) + lr;
      const bool ok = n < p.N;
      wp[j] = ok ? p.w + (size_t)n * p.K + kbase + lchunk : zsrc;
      winc[j] = ok ? BK : 0;
    }
  }
  // tap-inner: every k-tile is another tap of the same 64 channels: pointer = centre + (wave-uniform) tap offset, or the zero word
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
    if (tap_inner) { setup_rows_tap_inner(); return; }
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
  };
  NR_STAMP_AT(42);
  setup_rows();
  NR_STAMP_AT(43);

  auto stage = [&](int buf) {
    bf16* sA = smem + buf * TILE;
    bf16* sB = sA + BM * BK;
    if constexpr (ADMA) {
      const unsigned la = lds_addr(sA + wave * GA * 8 * BK), lb = lds_addr(sB + wave * GB * 8 * BK);
#ifdef NR_ABLATE_A
      // timing ablation (wrong results): the activation tile is fetched for one tap in nine only -- what a halo tile held in LDS would
      // leave of the A traffic of a tap-inner 3x3 conv (tools/conv_halo_potential.py; 2-stage instantiations wait with vmcnt(0), so the
      // barrier protocol stays valid)
      if (!tap_inner || st_tap == 0)
#endif
#pragma unroll
      for (int j = 0; j < GA; ++j) glds16_asm(ap[j], la + (unsigned)(j * 8 * BK * (int)sizeof(bf16)));
#pragma unroll
      for (int j = 0; j < GB; ++j) glds16_asm(wp[j], lb + (unsigned)(j * 8 * BK * (int)sizeof(bf16)));
    } else {
#pragma unroll
      for (int j = 0; j < GA; ++j) glds16_builtin(ap[j], sA + (wave * GA + j) * 8 * BK);
#pragma unroll
      for (int j = 0; j < GB; ++j) glds16_builtin(wp[j], sB + (wave * GB + j) * 8 * BK);
    }
    // advance to the next k-tile
#pragma unroll
    for (int j = 0; j < GB; ++j) wp[j] += winc[j];
    if (tap_inner) {
      st_tap += 1;
      if (st_tap == 9) { st_tap = 0; st_c += BK; }
      setup_rows_tap_inner();
      return;
    }
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
  };

  f32x4 acc[NT][MT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fg = lane >> 4;
  // LayerNorm fusion: row sums / sums of squares of the raw activation fragments.  The WGN waves that share an A slab
  // split the k-steps between them (v_dot2c_f32_bf16: 8 VALU ops per fragment), combined through LDS after the loop.
  float ln_s1[MT], ln_s2[MT];
#pragma unroll
  for (int j = 0; j < MT; ++j) { ln_s1[j] = 0.f; ln_s2[j] = 0.f; }
  float* ln_stat = reinterpret_cast<float*>(smem + NS * TILE);   // LNF only: 2 * WGN * BM floats behind the operand ring

  // ---- main loop: ring of NS LDS buffers, tiles kt+1 .. kt+NS-2 stay in flight across the barrier ----
  constexpr int G = GA + GB;                 // LDS-DMA instructions per wave per k-tile (wave-uniform)
#pragma unroll
  for (int s0 = 0; s0 < NS - 1; ++s0)
    if (kt_begin + s0 < kt_end) stage(s0);
  int cur = 0;
  NR_STAMP_AT(1);
#if NR_ABLATE & 4
  bf16x8 wf[2][NT], xf[2][MT];
#endif
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    // tile kt must have landed; the younger (NS-2) tiles may stay outstanding (vmcnt counts in issue order)
    if (kt + (NS - 2) < kt_end) wait_vmcnt<(NS - 2) * G>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();            // everyone's pieces of tile kt landed; everyone left tile kt-1
    NR_STAMP_AT(4 + (kt - kt_begin));
    const bf16* sA = smem + cur * TILE;
    const bf16* sB = sA + BM * BK;
    // fragment reads of k-step 0 go out FIRST, so their LDS latency is covered by the staging code below
    // (pointer bumps + LDS-DMA issue for tile kt+NS-1) instead of sitting exposed in front of the MFMAs
#if !(NR_ABLATE & 4)
    bf16x8 wf[2][NT], xf[2][MT];
#else
    if (kt == kt_begin) {
#endif
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int row = wn * WN + i * 16 + fr;
      wf[0][i] = *(const bf16x8*)(sB + row * BK + ((fg ^ (row & 7)) << 3));
    }
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      const int row = wm * WM + j * 16 + fr;
      xf[0][j] = *(const bf16x8*)(sA + row * BK + ((fg ^ (row & 7)) << 3));
    }
#if NR_ABLATE & 4
    }
#endif
    __builtin_amdgcn_sched_barrier(0);
    {
      const int nxt = kt + NS - 1;           // refill the buffer tile kt-1 occupied
      int nb = cur + NS - 1; if (nb >= NS) nb -= NS;
      if (nxt < kt_end && !(NR_ABLATE & 2)) stage(nb);
    }
    __builtin_amdgcn_sched_barrier(0);
#if NR_ABLATE & 4
    if (kt == kt_begin) {
#endif
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int row = wn * WN + i * 16 + fr;
      wf[1][i] = *(const bf16x8*)(sB + row * BK + (((4 + fg) ^ (row & 7)) << 3));
    }
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      const int row = wm * WM + j * 16 + fr;
      xf[1][j] = *(const bf16x8*)(sA + row * BK + (((4 + fg) ^ (row & 7)) << 3));
    }
#if NR_ABLATE & 4
    }
#endif
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j)
#if NR_ABLATE & 8
          asm volatile("" : : "v"(wf[ks][i]), "v"(xf[ks][j]));
#else
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][i], xf[ks][j], acc[i][j], 0, 0, 0);
#endif
    if constexpr (LNF) {
      const bf16x2 one2 = {(bf16)1.0f, (bf16)1.0f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if (((2 * kt + ks) & (WGN - 1)) != wn) continue;      // this k-step belongs to a sibling wave
#pragma unroll
        for (int j = 0; j < MT; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const bf16x2 pr = {xf[ks][j][2 * e], xf[ks][j][2 * e + 1]};
            ln_s2[j] = __builtin_amdgcn_fdot2_f32_bf16(pr, pr, ln_s2[j], false);
            ln_s1[j] = __builtin_amdgcn_fdot2_f32_bf16(pr, one2, ln_s1[j], false);
          }
      }
    }
    cur = cur + 1 == NS ? 0 : cur + 1;
  }
  NR_STAMP_AT(2);
  // per-row mean / rstd of this wave's MT row tiles (lane: row fr of each tile)
  float ln_mu[MT], ln_rs[MT];
  if constexpr (LNF) {
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      float a = ln_s1[j], b = ln_s2[j];
      a += __shfl_xor(a, 16, 64); b += __shfl_xor(b, 16, 64);
      a += __shfl_xor(a, 32, 64); b += __shfl_xor(b, 32, 64);
      if (fg == 0) {
        const int row = wm * WM + j * 16 + fr;
        ln_stat[(0 * WGN + wn) * BM + row] = a;
        ln_stat[(1 * WGN + wn) * BM + row] = b;
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      const int row = wm * WM + j * 16 + fr;
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int w = 0; w < WGN; ++w) { a += ln_stat[(0 * WGN + w) * BM + row]; b += ln_stat[(1 * WGN + w) * BM + row]; }
      const float inv = 1.0f / (float)p.K;
      const float mu = a * inv;
      ln_mu[j] = mu;
      ln_rs[j] = rsqrtf(fmaxf(b * inv - mu * mu, 0.f) + p.ln_eps);
    }
    // acc <- rstd_m * (acc - mean_m * sum_k W'[n][k]); the (beta . W + bias) term is p.bias, added by the epilogues below
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int n = n0 + wn * WN + i * 16 + 4 * fg;
      const f32x4 cn = n < p.N ? *(const f32x4*)(p.ln_c + n) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < MT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] = ln_rs[j] * (acc[i][j][r] - ln_mu[j] * cn[r]);
    }
  }

  // ---- epilogue: lane holds out[m = ..+fr][n = ..+4*fg + r], r = 0..3 ----
  if (partial) {
    float* slab = partial + (size_t)slice * p.M * p.N;
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      const int m = m0 + wm * WM + j * 16 + fr;
      if (m >= p.M) continue;
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const int n = n0 + wn * WN + i * 16 + 4 * fg;
        if (n >= p.N) continue;
        nr_store16f(slab + (size_t)m * p.N + n, acc[i][j]);
      }
    }
#ifndef NR_EXPERIMENTS
    return;
#else
    if (!l2red) return;
    // ---- in-launch reduction.  Every thread's slab stores are complete (acknowledged by the L2: the vector L1 is write-through) before the
    // workgroup's arrival is counted; the counter lives in the XCD's L2 (workgroup-scope atomic: executed there, not at the memory side), and so
    // do the sibling slabs the last arriver then reads (first touch by this CU in this launch: its L1, invalidated at kernel start, cannot hold
    // them).  Slices are summed in slice order whoever arrives last: bit-identical to splitk_reduce_kernel. ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* ctr = p.sk_ctr + tile_id;
    int* flag = reinterpret_cast<int*>(smem);
    if (tid == 0) {
      const int old = __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      const int last = old == splitk - 1;
      if (last) __hip_atomic_store(ctr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // self-cleaning: the next launch finds zeros
      *flag = last;
    }
    __syncthreads();
    const int last = *flag;
    __syncthreads();                       // (the staged epilogue below re-uses the LDS)
    if (!last) return;
    asm volatile("" ::: "memory");
    const size_t slab_elems = (size_t)p.M * p.N;
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      const int m = m0 + wm * WM + j * 16 + fr;
      if (m >= p.M) continue;
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const int n = n0 + wn * WN + i * 16 + 4 * fg;
        if (n >= p.N) continue;
        const float* src = partial + (size_t)m * p.N + n;
        f32x4 v = *(const f32x4*)src;
        for (int sl = 1; sl < splitk; ++sl) v += *(const f32x4*)(src + sl * slab_elems);
        acc[i][j] = v;
      }
    }
#endif
  }
  // Staged epilogue (whenever the fp32 C tile fits in the LDS ring): accumulators -> LDS (16-byte chunks
  // XOR-swizzled with row&7: conflict-free both ways) -> each thread handles 8 consecutive output channels of one
  // row, so bias / time-embedding / residual loads and the bf16 store are 16/32-byte accesses that cover whole
  // 128..256-byte row segments (the direct path below writes 8 bytes per lane in 32-byte segments).
  constexpr bool STAGED = (size_t)BM * BN * sizeof(float) <= (size_t)NS * TILE * sizeof(bf16);
  if constexpr (STAGED) {
    constexpr int NTH = 64 * NW;
    float* sC = reinterpret_cast<float*>(smem);
    __syncthreads();                       // every wave is done with the operand tiles
    const int bno = p.geglu ? BN / 2 : BN; // output columns of this tile
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      const int row = wm * WM + j * 16 + fr;
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
            f32x4 v = acc[i][j], g = acc[i + 1][j];
            if (p.bias && nv < p.N) { v += *(const f32x4*)(p.bias + nv); g += *(const f32x4*)(p.bias + nv + 16); }
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = v[e] * gelu_erf_fast(g[e]);
            const int c4 = (((wn * WN + i * 16) >> 1) >> 2) + fg;
            *(f32x4*)(sC + row * BN + ((c4 ^ (row & 7)) << 2)) = o;
          }
        }
      }
    }
    __syncthreads();
    const int c8n = bno >> 3;
    const int nout = p.geglu ? p.N / 2 : p.N;
    const int nb0 = p.geglu ? n0 / 2 : n0;
    for (int idx = tid; idx < BM * c8n; idx += NTH) {
      const int row = idx / c8n, c8 = idx - row * c8n;
      const int m = m0 + row, n = nb0 + c8 * 8;
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
          const bf16x8 r = *(const bf16x8*)(p.res + (size_t)m * p.ldr + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) { va[e] += (float)r[e]; vb[e] += (float)r[4 + e]; }
        }
      }
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) { o[e] = (bf16)va[e]; o[4 + e] = (bf16)vb[e]; }
      nr_store16(p.out + (size_t)m * p.ldo + n, o);
    }
#ifdef NR_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    NR_STAMP_AT(3);
    NR_STAMP_RT(45);
    return;
  }
#pragma unroll
  for (int j = 0; j < MT; ++j) {
    const int m = m0 + wm * WM + j * 16 + fr;
    if (m >= p.M) continue;
    const float* rv = p.rowvec ? p.rowvec + rowvec_row(p, m) : nullptr;
    if (!p.geglu) {
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const int n = n0 + wn * WN + i * 16 + 4 * fg;
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
        nr_store8(p.out + (size_t)m * p.ldo + n, o);
      }
    } else {
      if constexpr (NT % 2 == 0) {
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
          for (int e = 0; e < 4; ++e) o[e] = (bf16)(v[e] * gelu_erf_fast(g[e]));
          nr_store8(p.out + (size_t)m * p.ldo + oc, o);
        }
      }
    }
  }
}

// out[m][n] = epilogue( sum_s partial[s][m][n] ), 4 consecutive n per thread
__global__ __launch_bounds__(256) void splitk_reduce_kernel(NrGemmParams p_arg, int splitk, const float* __restrict__ partial) {
  const NrGemmParams p = nr_pin_params(p_arg);
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const int n4 = p.N >> 2;
  if (idx >= (long long)p.M * n4) return;
  const int m = (int)(idx / n4);
  const int n = (int)(idx - (long long)m * n4) << 2;
  const size_t slab = (size_t)p.M * p.N;
  const float* src = partial + (size_t)m * p.N + n;
  f32x4 v = *(const f32x4*)src;
  for (int s = 1; s < splitk; ++s) v += *(const f32x4*)(src + s * slab);
  if (p.bias) v += *(const f32x4*)(p.bias + n);
  if (p.rowvec) v += *(const f32x4*)(p.rowvec + rowvec_row(p, m) + n);
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
  nr_store8(p.out + (size_t)m * p.ldo + n, o);
}

struct Plan { int bm, bn, splitk, stages, waves; };

// Tile / wave-grid / split-K choice.  Rules distilled from tools/gemm_sweep.py on MI355X (profiles/): two LDS
// stages (2 workgroups per CU) beat a deeper ring; 8 waves help the 128x128 tile; long-K layers with few
// tiles gain 1.3-1.7x from split-K; 128x160 only pays for N = 320.
Plan choose_plan(const NrGemmParams& p) {
  auto nblk = [&](int bm, int bn) { return (long long)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn); };
  const int nk = p.K / 64;
  Plan pl;
  pl.splitk = 1; pl.stages = 2; pl.waves = 4;
  // 4x4-level convs (M <= 1024, K >= 8192): L2-bandwidth-bound with small tiles (every n-tile re-reads A, every m-tile
  // re-reads W), so take the biggest tile and get the parallelism from a deep deterministic split-K instead
  if (!p.geglu && p.M <= 1024 && nk >= 128 && p.N % 160 == 0) {
    pl.bm = 128; pl.bn = 160; pl.splitk = nk >= 300 ? 16 : 8;
    return pl;
  }
  // same reasoning one level up (8x8 convs, M = 2048, K >= 8192): +1.5 % on the DDIM step over 128x64 tiles with split-K 4
  if (!p.geglu && p.ksize == 3 && p.M <= 2048 && nk >= 128 && p.N % 160 == 0) {
    pl.bm = 128; pl.bn = 160; pl.splitk = 4;
    return pl;
  }
  // 16x16-level convs (M = 8192, N = 640, K >= 5760): 128x160 with two K slices instead of 128x64 (per-shape in-situ A/B,
  // tools/igemm_ab_shapes.py: -0.06 / -0.09 ms per DDIM step)
  if (!p.geglu && p.ksize == 3 && p.M <= 8192 && p.N % 160 == 0 && p.N < 960 && nk >= 64 && nblk(128, 160) >= 128 && nblk(128, 160) < 512) {
    pl.bm = 128; pl.bn = 160; pl.splitk = 2;
    return pl;
  }
  // folded FeedForward.net.2 + proj_out GEMMs (K = 5C = 6400) at the 8x8 level: same as the long-K convs above (tools/sweep_ff.sh:
  // 46.6 vs 55.2 us)
  if (!p.geglu && p.ksize == 1 && p.M <= 2048 && p.M > 512 && nk >= 96 && p.N % 160 == 0) {
    pl.bm = 128; pl.bn = 160; pl.splitk = 4;
    return pl;
  }
  // Linears whose 64 x 160 tiling is ONE round of the chip (128-256 tiles: N = 1280 at the 8x8 level, M = 1024 / 2048) with K = 1024..3072: one
  // workgroup per CU with a 3-deep ring instead of 320-640 smaller tiles two per CU: 16.2 vs 19.2 us at M = 2048, N = K = 1280, 26.5 vs 36.3 at
  // K = 2560, 17.1 vs 20.1 LayerNorm-folded (2-deep ring there) (profiles/r05_sweep_64x160.txt, HBM-cold weights)
  static const bool t64x160_rule = !(getenv("NR_IGEMM_T64X160") && getenv("NR_IGEMM_T64X160")[0] == '0');   // A/B switch
  if (t64x160_rule && p.ksize == 1 && !p.geglu && p.N % 160 == 0 && p.M > 512 && nk >= 16 && nk <= 48 && nblk(64, 160) >= 128 && nblk(64, 160) <= 256) {
    pl.bm = 64; pl.bn = 160; pl.stages = 3;
    return pl;
  }
  const bool n128 = p.N % 128 == 0 || p.N >= 960;          // <= 6 % padded columns otherwise
  if (!p.geglu && p.N % 160 == 0 && p.N < 960 && p.N % 128 != 0 && nk >= 20 && nblk(128, 160) >= 256) { pl.bm = 128; pl.bn = 160; }
  else if (n128 && nblk(128, 128) >= 400) { pl.bm = 128; pl.bn = 128; pl.waves = 8; }
  else if (nblk(128, 64) >= 512 || (nblk(128, 64) >= 256 && nk > 32)) { pl.bm = 128; pl.bn = 64; pl.waves = nk <= 32 ? 8 : 4; }
  else if (nblk(64, 64) >= 256 && nk <= 96) { pl.bm = 64; pl.bn = 64; }
  else if (p.M >= 1024 && nblk(128, 64) >= 64) { pl.bm = 128; pl.bn = 64; }
  else { pl.bm = 64; pl.bn = 64; }
  // 4x4-level / sgm 16x16-level Linears (M <= 512): fewer blocks than CUs, every k-step waits for cold weights from HBM.
  // A 4-deep ring (3 tiles in flight) and, where the epilogue allows, 64x32 tiles (twice the blocks) measured -10 % on the
  // sgm keyframe step (tools/igemm_ab_sgm.sh); deeper rings (6, 8) and split-K + reduce were slower.
  static const bool smallm_rule = !(getenv("NR_IGEMM_SMALLM") && getenv("NR_IGEMM_SMALLM")[0] == '0');   // A/B switch
  static const bool wide_rule = !(getenv("NR_IGEMM_WIDE") && getenv("NR_IGEMM_WIDE")[0] == '0');         // A/B switch
  static const bool rowwave_rule = !(getenv("NR_IGEMM_ROWWAVE") && getenv("NR_IGEMM_ROWWAVE")[0] == '0');   // A/B switch
  if (smallm_rule && p.ksize == 1 && p.M <= 512 && nk >= 8) {
    pl.bm = 64; pl.bn = p.geglu ? 64 : 32; pl.waves = 4; pl.stages = 4;
    // wide GEGLU projections (N = 10240): enough 128x128 tiles for the chip, 22.4 vs 29.9 us; K = 6400 (folded net.2 + proj_out):
    // 64x64 tiles with four K slices, 23.1 vs 29.6 us (tools/sweep_ff.sh)
    if (p.geglu && p.N >= 8192 && p.M >= 256) { pl.bm = 128; pl.bn = 128; pl.waves = 8; pl.stages = 2; }
    // ... and exactly 256 tiles of 128 x 160 at M = 512 (one round, one workgroup per CU, ring 3-4 deep: the 320 tiles of 128 x 128 wait for
    // HBM-cold k-tiles with one in flight): 4 x 1 waves of 32 x 160 so that a wave holds whole (value, gate) pairs; 23.1 vs 28.9 us plain,
    // 27.0 vs 31.7 us LayerNorm-folded (profiles/r05_sweep_rowwave.txt)
    if (rowwave_rule && p.geglu && p.N >= 8192 && p.N % 160 == 0 && p.M > 256 && nblk(128, 160) <= 256) {
      pl.bm = 128; pl.bn = 160; pl.waves = 41; pl.stages = p.ln_c ? 3 : 4;
    }
    else if (!p.geglu && nk >= 96 && !p.ln_c) { pl.bm = 64; pl.bn = 64; pl.stages = 2; pl.splitk = 4; return pl; }
    // wide projections (q|k|v, N = 3840): 960 tiles of 64 x 32 re-read A 120 times; 240 tiles of 128 x 64 with the same 4-deep ring:
    // 13.3 vs 19.8 us plain, 15.2 vs 17.5 us LayerNorm-folded (profiles/r05_sweep_ln_ns4.txt, HBM-cold weights)
    else if (wide_rule && !p.geglu && nblk(64, 32) > 640 && nblk(128, 64) >= 192) { pl.bm = 128; pl.bn = 64; pl.waves = 8; pl.stages = 4; }
  }
  if (!p.geglu) {
    const long long tiles = nblk(pl.bm, pl.bn);
    const int cap_m = p.M >= 8192 ? 2 : (p.M >= 2048 ? 4 : 8);      // bounds the fp32 slab traffic (8*M*N*split bytes)
    const int min_tiles = p.M <= 1024 ? 8 : 16;                      // k-tiles each slice keeps
    if (tiles < 512 && nk >= 4 * min_tiles) {
      int s_ = (int)((1280 + tiles - 1) / tiles);
      if (s_ > nk / min_tiles) s_ = nk / min_tiles;
      if (s_ > cap_m) s_ = cap_m;
      if (s_ < 1) s_ = 1;
      pl.splitk = s_;
    }
  }
  return pl;
}

// the shape the plan is chosen for: plan_m rows when the caller asks for batch-independent arithmetic (common.h), else the real M
inline NrGemmParams plan_view(const NrGemmParams& p) {
  NrGemmParams q = p;
  if (p.plan_m > 0 && p.plan_m < p.M) q.M = p.plan_m;
  return q;
}

// hipFuncSetAttribute is per device: one bit per device ordinal and instantiation
inline bool attr_needed(unsigned long long& mask) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (mask & bit) return false;
  mask |= bit;
  return true;
}

// returns 0 on success, 9 when the requested (tile, LayerNorm-fused) combination has no instantiation
template <int BM, int BN, int NS, int WGM, int WGN>
int launch_cfg(const NrGemmParams& p, unsigned grid, int splitk, float* partial, int m_fast, hipStream_t stream) {
  const size_t shm = (size_t)NS * (BM + BN) * 64 * sizeof(bf16);
  // LayerNorm-fused variant: instantiated for the tiles the transformer GEMMs use (nr_launch_igemm maps others onto them).
  // Its row-statistics exchange buffer lives in the DYNAMIC allocation behind the ring: a static __shared__ array next to
  // > 64 KiB of dynamic LDS made the first launch (and any hipGraph node captured from it) run with a short allocation.
  constexpr bool LN_OK = (NS == 2 && BM <= 128 && BN <= 128) || (NS == 4 && (BM * BN <= 64 * 64 || (BM == 128 && BN == 64))) ||
                         (BM == 128 && BN == 160 && WGN == 1 && NS <= 4) ||     // the row-wave 128x160 tile of the wide GEGLU projections
                         (BM == 64 && BN == 160 && NS <= 3);                     // the one-round tile of the N = 1280 Linears at M = 1024 / 2048
  if (p.ln_c) {
    if constexpr (LN_OK) {
      const size_t shm_ln = shm + (size_t)2 * WGN * BM * sizeof(float);
      static unsigned long long attr_ln = 0;
      if (attr_needed(attr_ln))
        (void)hipFuncSetAttribute((const void*)igemm_bf16_kernel<BM, BN, NS, WGM, WGN, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm_ln);
      hipLaunchKernelGGL((igemm_bf16_kernel<BM, BN, NS, WGM, WGN, true>), dim3(grid), dim3(64 * WGM * WGN), shm_ln, stream, p, splitk,
                         partial, m_fast);
      return 0;
    }
    return 9;
  }
  static unsigned long long attr_set = 0;
  if (attr_needed(attr_set)) {  // > 64 KiB of dynamic LDS needs the opt-in attribute (gfx950 has 160 KiB per CU), per device
    (void)hipFuncSetAttribute((const void*)igemm_bf16_kernel<BM, BN, NS, WGM, WGN, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    (void)hipFuncSetAttribute((const void*)igemm_bf16_kernel<BM, BN, NS, WGM, WGN, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
  }
  static const int adma_min = getenv("NR_IGEMM_ADMA_MINK") ? atoi(getenv("NR_IGEMM_ADMA_MINK")) : 24;   // k-tiles per slice; A/B switch
  const int nk_slice = (p.K / 64) / (splitk > 0 ? splitk : (splitk < 0 ? -splitk : 1));
  // plain Linears (1x1, one source) on the instantiation without the conv paths (rings up to 4 deep: the ones Linears are planned with)
  static const bool lin_on = !(getenv("NR_IGEMM_LIN") && getenv("NR_IGEMM_LIN")[0] == '0');            // A/B switch
  if constexpr (NS <= 4) {
    if (lin_on && p.ksize == 1 && !p.a1 && p.c1 == 0) {
      static unsigned long long attr_lin = 0;
      if (attr_needed(attr_lin)) {
        (void)hipFuncSetAttribute((const void*)igemm_bf16_kernel<BM, BN, NS, WGM, WGN, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        (void)hipFuncSetAttribute((const void*)igemm_bf16_kernel<BM, BN, NS, WGM, WGN, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
      }
      if (nk_slice >= adma_min)
        hipLaunchKernelGGL((igemm_bf16_kernel<BM, BN, NS, WGM, WGN, false, true, true>), dim3(grid), dim3(64 * WGM * WGN), shm, stream, p, splitk, partial, m_fast);
      else
        hipLaunchKernelGGL((igemm_bf16_kernel<BM, BN, NS, WGM, WGN, false, false, true>), dim3(grid), dim3(64 * WGM * WGN), shm, stream, p, splitk, partial, m_fast);
      return 0;
    }
  }
  if (nk_slice >= adma_min)
    hipLaunchKernelGGL((igemm_bf16_kernel<BM, BN, NS, WGM, WGN, false, true>), dim3(grid), dim3(64 * WGM * WGN), shm, stream, p, splitk,
                       partial, m_fast);
  else
    hipLaunchKernelGGL((igemm_bf16_kernel<BM, BN, NS, WGM, WGN, false, false>), dim3(grid), dim3(64 * WGM * WGN), shm, stream, p, splitk,
                       partial, m_fast);
  return 0;
}

template <int BM, int BN, int WGM, int WGN>
int launch_tile(const NrGemmParams& p, unsigned grid, const Plan& pl, float* partial, int m_fast, hipStream_t stream) {
  constexpr size_t STAGE = (size_t)(BM + BN) * 64 * sizeof(bf16);
  constexpr bool FITS4 = 4 * STAGE <= 160 * 1024;
  if (pl.stages <= 2) return launch_cfg<BM, BN, 2, WGM, WGN>(p, grid, pl.splitk, partial, m_fast, stream);
  if (pl.stages == 3 || !FITS4) return launch_cfg<BM, BN, 3, WGM, WGN>(p, grid, pl.splitk, partial, m_fast, stream);
  if (pl.stages <= 4 || BM * BN > 64 * 64) return launch_cfg<BM, BN, 4, WGM, WGN>(p, grid, pl.splitk, partial, m_fast, stream);   // deep ring
  if (pl.stages <= 6) return launch_cfg<BM, BN, 6, WGM, WGN>(p, grid, pl.splitk, partial, m_fast, stream);
  return launch_cfg<BM, BN, 8, WGM, WGN>(p, grid, pl.splitk, partial, m_fast, stream);
}

// test/tuning override: NR_IGEMM_FORCE="bm,bn,splitk,stages,order" (any field <0 keeps the heuristic's choice)
void apply_override(const NrGemmParams& p, Plan& pl, int& m_fast) {
  const char* e = getenv("NR_IGEMM_FORCE");
  if (!e) return;
  // optional filters so an in-situ A/B (bench.py under hipGraph, no per-launch host overhead) can target one shape class:
  // NR_IGEMM_FORCE_MAXM / _MINM bound p.M, NR_IGEMM_FORCE_KS selects 1x1 or 3x3 launches
  if (const char* f = getenv("NR_IGEMM_FORCE_MAXM")) if (p.M > atoi(f)) return;
  if (const char* f = getenv("NR_IGEMM_FORCE_MINM")) if (p.M < atoi(f)) return;
  if (const char* f = getenv("NR_IGEMM_FORCE_KS")) if (p.ksize != atoi(f)) return;
  if (const char* f = getenv("NR_IGEMM_FORCE_N")) if (p.N != atoi(f)) return;
  if (const char* f = getenv("NR_IGEMM_FORCE_K")) if (p.K != atoi(f)) return;
  int bm = -1, bn = -1, sk = -1, st = -1, ord = -1, wv = -1;
  sscanf(e, "%d,%d,%d,%d,%d,%d", &bm, &bn, &sk, &st, &ord, &wv);
  if (bm > 0 && bn > 0) {
    const bool ok = (bm == 128 && (bn == 160 || bn == 128 || bn == 64)) || (bm == 64 && (bn == 64 || bn == 32 || bn == 160)) || (bm == 256 && (bn == 128 || bn == 160));
    if (ok && !(p.geglu && ((bn == 160 && !(bm == 128 && wv == 41)) || bn == 32))) { pl.bm = bm; pl.bn = bn; }
  }
  if (wv == 4 || wv == 8) pl.waves = wv;
  if (pl.bm == 256 && pl.bn == 128 && wv != 4) pl.waves = 8;
  if (pl.bn == 160 || pl.bm == 64) pl.waves = 4;
  if (wv == 41 && pl.bm == 128 && pl.bn == 160) pl.waves = 41;      // 4 x 1 waves (32 x 160 each): the row-wave tile
  if (sk > 0 && !p.geglu) { const int nk = p.K / 64; pl.splitk = sk > nk ? nk : sk; }
  if (st >= 2 && st <= 8) pl.stages = st;
  if (ord >= 0) m_fast = ord;
}

}  // namespace

extern "C" int nr_rowpanel_eligible(const NrGemmParams* pp);
extern "C" int nr_launch_rowpanel(const NrGemmParams* pp, hipStream_t stream);
// gemm8p.hip: 256-row ping-pong kernel for the big launches (SparseCtrl groups, several clips per call, 32-frame clips, the VAE)
extern "C" int nr_g8p_plan(const NrGemmParams* pp);
extern "C" int nr_launch_g8p(const NrGemmParams* pp, int m_fast, hipStream_t stream);
// smallm.hip: panel-resident kernel for the M <= 512 Linears; needs the fragment-major copy of the weights (NrGemmParams::w_fm)
extern "C" int nr_smallm_eligible(const NrGemmParams* pp);
extern "C" int nr_launch_smallm(const NrGemmParams* pp, hipStream_t stream);
#ifdef NR_EXPERIMENTS
// Rejected experiments (csrc/experiments/, built only by `make experiments` into libneurons_amd_exp.so for the A/B tools; never the product):
// gemm256.hip: 256-row tiles with role-alternating wave groups for the long-K convs / Linears (NR_IGEMM256=2)
extern "C" int nr_igemm256_plan(const NrGemmParams* pp, int* bn_out, int* splitk_out);
extern "C" size_t nr_igemm256_workspace_bytes(const NrGemmParams* pp);
extern "C" int nr_launch_igemm256(const NrGemmParams* pp, float* workspace, int m_fast, int* splitk_used, hipStream_t stream);

// gemmws.hip: four MFMA waves + one LDS-DMA wave per 128 x 128 tile (NR_IGEMM_WS=1|2)
extern "C" int nr_igemm_ws_plan(const NrGemmParams* pp, int* splitk_out);
extern "C" size_t nr_igemm_ws_workspace_bytes(const NrGemmParams* pp);
extern "C" int nr_launch_igemm_ws(const NrGemmParams* pp, float* workspace, int m_fast, int* splitk_used, hipStream_t stream);
#endif

// fp32 scratch (bytes) a launch of this shape needs for split-K slabs (0 if none)
extern "C" size_t nr_igemm_workspace_bytes(const NrGemmParams* pp) {
  if (pp->out_f32 || pp->ln_c) return 0;
  if (nr_rowpanel_eligible(pp)) return 0;
  if (pp->w_fm) return 0;                      // the caller chose smallm.hip when it built this description (nr_smallm_eligible)
  if (nr_g8p_plan(pp)) return 0;
#ifdef NR_EXPERIMENTS
  if (nr_igemm_ws_plan(pp, nullptr)) return nr_igemm_ws_workspace_bytes(pp);
  if (nr_igemm256_plan(pp, nullptr, nullptr)) return nr_igemm256_workspace_bytes(pp);
#endif
  Plan pl = choose_plan(plan_view(*pp));
  int mf = 0;
  apply_override(*pp, pl, mf);
  return pl.splitk > 1 ? (size_t)pl.splitk * pp->M * pp->N * sizeof(float) : 0;
}

// Tile counters an in-launch split-K reduction of this description needs (NrGemmParams::sk_ctr), 0 when the launch is not a split-K launch of
// the tiled igemm, its tile count is not a multiple of 8 (whole tiles per XCD), or the experiment is off (experiments library + NR_SPLITK_L2=1)
extern "C" int nr_igemm_splitk_l2_tiles(const NrGemmParams* pp) {
#ifndef NR_EXPERIMENTS
  (void)pp;
  return 0;             // product library: slabs + splitk_reduce_kernel (the in-launch form lost its A/B)
#endif
  static const bool on = getenv("NR_SPLITK_L2") && getenv("NR_SPLITK_L2")[0] == '1';
  if (!on || nr_igemm_workspace_bytes(pp) == 0 || pp->out_f32) return 0;
  Plan pl = choose_plan(plan_view(*pp));
  int mf = 0;
  apply_override(*pp, pl, mf);
  if (pl.splitk <= 1) return 0;
  const int tiles = ((pp->M + pl.bm - 1) / pl.bm) * ((pp->N + pl.bn - 1) / pl.bn);
  return tiles % 8 == 0 ? tiles : 0;
}

// Host launcher.  Returns 0 on success, nonzero on unsupported shape.  `workspace` must hold
// nr_igemm_workspace_bytes() bytes when that is nonzero.
extern "C" int nr_launch_igemm(const NrGemmParams* pp, float* workspace, hipStream_t stream) {
  const NrGemmParams& p = *pp;
  const int Cin = p.c0 + p.c1;
  // K = 320 Linears on >= 4096 rows: the register-resident row-panel kernel (rowpanel.hip)
  if (p.K == p.ksize * p.ksize * Cin && nr_rowpanel_eligible(pp)) return nr_launch_rowpanel(pp, stream);
  // M <= 512 Linears with K a multiple of 640 whose weights the caller also holds fragment-major: the panel-resident kernel (smallm.hip).
  // w_fm IS the decision (made once with nr_smallm_eligible when the description was built); a shape the kernel cannot serve is an error here
  if (p.w_fm) return nr_launch_smallm(pp, stream);
  if (const int nt8 = nr_g8p_plan(pp)) {
    // tile order as below: the bigger operand is the one neighbouring tiles share; 8 x 4 blocks of tiles per XCD for weight-heavy shapes
    const double w_e = (double)p.N * p.K;
    const double a_e = (double)p.M * Cin;
    int mf = w_e > a_e ? 1 : 0;
    const int ntm_ = (p.M + 255) / 256, ntn_ = (p.N + 64 * nt8 - 1) / (64 * nt8);
    if (ntm_ >= 8 && ntn_ >= 4 && w_e >= 3.0e6) mf = 8;
    return nr_launch_g8p(pp, mf, stream);
  }
#ifdef NR_EXPERIMENTS
  if (!getenv("NR_IGEMM_FORCE") && nr_igemm_ws_plan(pp, nullptr)) {
    const double w_e = (double)p.N * p.K, a_e = (double)p.M * Cin;
    int mf = w_e > a_e ? 1 : 0;
    const int ntm_ = (p.M + 127) / 128, ntn_ = (p.N + 127) / 128;
    if (ntm_ >= 8 && ntn_ >= 4 && w_e >= 3.0e6) mf = 8;
    int used = 1;
    const int rc = nr_launch_igemm_ws(pp, workspace, mf, &used, stream);
    if (rc) return rc;
    if (used > 1) {
      const long long total = (long long)p.M * (p.N / 4);
      hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, p, used, (const float*)workspace);
    }
    return 0;
  }
  {
    int bn256 = 0, sk256 = 1;
    if (nr_igemm256_plan(pp, &bn256, &sk256)) {
      const double w_e = (double)p.N * p.K;
      const double a_e = (double)p.M * Cin * (p.ksize == 3 ? (p.stride == 2 ? 4.0 : (p.ups ? 0.25 : 1.0)) : 1.0);
      int mf = w_e > a_e ? 1 : 0;
      const int ntm_ = (p.M + 255) / 256, ntn_ = (p.N + bn256 - 1) / bn256;
      if (ntm_ >= 8 && ntn_ >= 4 && w_e >= 3.0e6) mf = 8;
      int used = 1;
      const int rc = nr_launch_igemm256(pp, workspace, mf, &used, stream);
      if (rc) return rc;
      if (used > 1) {
        const long long total = (long long)p.M * (p.N / 4);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, p, used, (const float*)workspace);
      }
      return 0;
    }
  }
#endif
  if (p.K % 64 != 0 || Cin % 64 != 0 || p.N % 32 != 0) return 1;
  if (p.a1 && (p.c0 % 64 != 0)) return 2;
  if (p.K != p.ksize * p.ksize * Cin) return 3;
  if (p.ksize != 1 && p.ksize != 3) return 4;
  if (p.tap_inner && (p.ksize != 3 || p.stride != 1 || p.ups || p.pad_tl0 || p.a1)) return 5;
  Plan pl = choose_plan(plan_view(p));
  const double w_elems = (double)p.N * p.K;
  const double a_elems = (double)p.M * Cin * (p.ksize == 3 ? (p.stride == 2 ? 4.0 : (p.ups ? 0.25 : 1.0)) : 1.0);
  int m_fast = w_elems > a_elems ? 1 : 0;
  {
    // enough tiles both ways: 8 x 8 blocks of tiles (tools/gemm_sweep.py SWEEP_SET=order: never slower than either plain order, up to
    // 12 % faster where the weights exceed one L2, and fewer L2 misses)
    static const bool grouped = !(getenv("NR_IGEMM_GROUPED") && getenv("NR_IGEMM_GROUPED")[0] == '0');
    const int ntm_ = (p.M + pl.bm - 1) / pl.bm, ntn_ = (p.N + pl.bn - 1) / pl.bn;
    static const double grouped_minw = getenv("NR_IGEMM_GROUPED_MINW") ? atof(getenv("NR_IGEMM_GROUPED_MINW")) : 3.0e6;   // weight elements (in situ: 1e6 / 3e6 / 0 = 15.85 / 15.89 / 15.80 frames/s, off 15.93; HBM 49.3 -> 44.4 GB per step at 1e6)
    if (grouped && ntm_ >= 8 && ntn_ >= 4 && w_elems >= grouped_minw) m_fast = 8;
  }
  apply_override(p, pl, m_fast);
  if (p.ln_c) {      // LayerNorm-fused: every block must see the whole row (K = C) -> no split-K; supported tiles only
    if (p.ksize != 1 || p.a1 || p.out_f32) return 8;
    pl.splitk = 1;
    if (pl.bn == 32) pl.bn = 64;             // every n-tile recomputes the row statistics: keep the n-tiles wide
    const bool t64x160 = pl.bm == 64 && pl.bn == 160;
    if (!(pl.stages == 4 && (pl.bm * pl.bn <= 64 * 64 || (pl.bm == 128 && pl.bn == 64))) && pl.waves != 41 && !(t64x160 && pl.stages <= 3)) pl.stages = 2;
    if (pl.bm > 128) { pl.bm = 128; pl.bn = 128; pl.waves = 8; }
    if (pl.bn > 128 && !(pl.bm == 128 && pl.bn == 160 && pl.waves == 41) && !t64x160) { pl.bm = 128; pl.bn = 128; pl.waves = 8; }
    if (pl.waves == 41 && pl.stages > 4) pl.stages = 4;
  }
  if (p.out_f32) {   // raw fp32 result: the kernel's slab path with a single K slice, no reduce pass
    if (p.geglu) return 7;
    pl.splitk = 1;
  }
  if (pl.splitk > 1 && !workspace) return 6;
  float* partial = p.out_f32 ? p.out_f32 : (pl.splitk > 1 ? workspace : nullptr);
  const unsigned grid = (unsigned)(((p.M + pl.bm - 1) / pl.bm) * ((p.N + pl.bn - 1) / pl.bn) * pl.splitk);
  // in-launch split-K reduction (the caller provided tile counters): all slices of a tile on one XCD needs whole tiles per XCD range
  const int sk_slices = pl.splitk;
  const bool l2red = pl.splitk > 1 && p.sk_ctr && !p.out_f32 && (grid / pl.splitk) % 8 == 0;
  if (l2red) pl.splitk = -pl.splitk;
  int rc;
  if (pl.bm == 256 && pl.bn == 160) rc = launch_tile<256, 160, 2, 2>(p, grid, pl, partial, m_fast, stream);
  else if (pl.bm == 256 && pl.waves == 4) rc = launch_tile<256, 128, 2, 2>(p, grid, pl, partial, m_fast, stream);
  else if (pl.bm == 256) rc = launch_tile<256, 128, 4, 2>(p, grid, pl, partial, m_fast, stream);
  else if (pl.bm == 128 && pl.bn == 160 && pl.waves == 41)      // row-wave tile: rings 3 / 4 deep only (108 / 144 KiB: one workgroup per CU)
    rc = pl.stages <= 3 ? launch_cfg<128, 160, 3, 4, 1>(p, grid, pl.splitk, partial, m_fast, stream)
                        : launch_cfg<128, 160, 4, 4, 1>(p, grid, pl.splitk, partial, m_fast, stream);
  else if (pl.bm == 128 && pl.bn == 160) rc = launch_tile<128, 160, 2, 2>(p, grid, pl, partial, m_fast, stream);
  else if (pl.bm == 128 && pl.bn == 128 && pl.waves == 8) rc = launch_tile<128, 128, 2, 4>(p, grid, pl, partial, m_fast, stream);
  else if (pl.bm == 128 && pl.bn == 128) rc = launch_tile<128, 128, 2, 2>(p, grid, pl, partial, m_fast, stream);
  else if (pl.bm == 128 && pl.bn == 64 && pl.waves == 8) rc = launch_tile<128, 64, 4, 2>(p, grid, pl, partial, m_fast, stream);
  else if (pl.bm == 128 && pl.bn == 64) rc = launch_tile<128, 64, 2, 2>(p, grid, pl, partial, m_fast, stream);
  else if (pl.bm == 64 && pl.bn == 160) rc = launch_tile<64, 160, 2, 2>(p, grid, pl, partial, m_fast, stream);
  else if (pl.bm == 64 && pl.bn == 32) rc = launch_tile<64, 32, 2, 2>(p, grid, pl, partial, m_fast, stream);
  else if (pl.bm == 64 && pl.bn == 64) rc = launch_tile<64, 64, 2, 2>(p, grid, pl, partial, m_fast, stream);
  else rc = 9;            // no such tile: never launch another one on a grid computed for this one
  if (rc) return rc;      // no instantiation for this (tile, LayerNorm-fused) request: fail loudly, never skip the launch
  (void)sk_slices;
  if (pl.splitk > 1) {
    const long long total = (long long)p.M * (p.N / 4);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, p, pl.splitk,
                       (const float*)partial);
  }
  return 0;
}

#ifdef NR_STAMP
extern "C" int nr_stamp_read(void* dst, size_t bytes, int clear) {
  const size_t n = bytes < sizeof(nr_stamp_buf) ? bytes : sizeof(nr_stamp_buf);
  int rc = 0;
  if (dst) rc = (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(nr_stamp_buf), n, 0, hipMemcpyDeviceToHost);
  if (clear) { void* d = nullptr; (void)hipGetSymbolAddress(&d, HIP_SYMBOL(nr_stamp_buf)); (void)hipMemset(d, 0, sizeof(nr_stamp_buf)); }
  return rc;
}
#endif
