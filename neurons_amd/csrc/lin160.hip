// Short-K Linear above the C = 320 level:  out[M][N] = x[M][K] . W[N][K]^T + bias (+ residual),  K = 640 / 1280, N a multiple of 160 --
// the `proj_in` / `to_out` Linears of the spatial and temporal transformers at C = 640 / 1280 (animatediff/models/attention.py:110-113,135-140,
// motion_module.py:146-158; CrossAttention.to_out motion_module_new.py:170-171): 30 launches per level and DDIM step that the tiled igemm serves at
// 0.33 PFLOP/s (20-21 us for 6.7 GFLOP: 10-20 k-tiles per workgroup, 38 % of a workgroup's life outside the k-loop, 640 tiles = 2.5 rounds;
// DESIGN 3a).  Same skeleton as the head kernels of round 6 (tattnw.hip / xattnw.hip), reduced to the GEMM:
//   * one 512-thread workgroup per (BM rows, 160 columns), BM = 128 (wave w: row tile w, all 10 column tiles) or 64 (wave w: row tile w & 3, five
//     column tiles): two waves per SIMD; the grid is exactly one round of the chip at the headline shapes (64 x 4 and 32 x 8 workgroups);
//   * W FRAGMENT-MAJOR ([column block][stage][k-step][tile][64 lanes][8]), streamed two k-steps (64 channels: 20 fragments + 4 KiB of padding =
//     3 DMA pieces per wave) per stage through a 3-slot LDS ring by linear LDS-DMA; the rows of x through the same ring (one 16-row x 64-byte
//     piece per row tile and k-step, chunk-permuted: conflict-free fragment reads); every workgroup starts at its own stage (L2 channel spread);
//   * accumulators as W . x^T (lane = 4 consecutive output channels of its row); epilogue: bias, residual, bf16, 8-byte stores (in place when
//     the residual is the output).
// Second kernel of this file, further down: lin128q_kernel, the REGISTER-PANEL form for the LayerNorm-folded wide projections of the same levels (FeedForward.net[0] with
// GEGLU, N = 8 C, motion_module_new.py:441-518; the spatial self-attention's fused to_q|to_k|to_v, N = 3 C, motion_module_new.py:201-230 behind norm1 of attention.py:272-285):
// the rows stay in registers for the whole launch and only W streams.
#include "common.h"
#include <cstdlib>

namespace {

typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void glds16(const void* src, unsigned lds_wave_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(lds_wave_base) : "memory", "m0");
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

#ifdef NR_STAMP
// diagnostic build only (make stamp, tools/tattnw_timeline.py lin160): shader-clock stamps of wave 0 of the first 512 workgroups.  Slots: 0 entry, 1 prologue issued,
// 2 + 3 s / 3 + 3 s / 4 + 3 s = stage s (< 40) after its DMA wait / barrier / MFMAs, 125 loop end, 126 kernel end
__device__ unsigned long long lin160_stamp_buf[512][128];
#define L1_STAMP(slot) do { if (threadIdx.x == 0 && blockIdx.x < 512 && (slot) < 128) lin160_stamp_buf[blockIdx.x][(slot)] = __builtin_amdgcn_s_memtime(); } while (0)
// panel kernel, TIMING-ONLY ablations of its stage loop (results wrong; stamp build only, compile-time template argument chosen by NR_Q_DBG at launch; bits: 1 no stage
// barrier, 2 no DMA pieces in the loop, 4 no fragment reads in the loop, 8 no MFMAs, 16 no DMA wait, 32 no block epilogue)
#define Q_DBG(bit) (DBG & (bit))
// panel kernel, fine stamps of ONE steady-state stage (g = 11) for wave 0 (rows 0 .. 255 of the buffer) and its SIMD-mate wave 4 (rows 256 .. 511): slots 64 + 3 kk = k-step kk
// begins, 65 + 3 kk = its DMA burst (if any) issued, 66 + 3 kk = its MFMAs issued; 76 / 77 / 78 = in front of the DMA wait / behind it / behind the barrier (inside k-step 3)
#ifdef NR_STAMP_FINE
#define Q_STAMPW(slot) do { if ((threadIdx.x & 63) == 0 && (wave == 0 || wave == 4) && blockIdx.x < 256) lin160_stamp_buf[blockIdx.x + (wave ? 256 : 0)][(slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define Q_STAMPW(slot) do { } while (0)      // (they perturb the stage they measure: every stamp's store drains the DMA ring; kept for relative order only)
#endif
#else
#define L1_STAMP(slot) do { } while (0)
#define Q_STAMPW(slot) do { } while (0)
#define Q_DBG(bit) 0
#endif

constexpr int L1_BN = 160, L1_NT = 10;
constexpr int L1_W_STAGE = 24 * 1024;            // 2 k-steps x 10 fragments of 1 KiB + 4 KiB pad: 3 pieces per wave
constexpr int L1_NS = 3;

struct NrLin160Params {
  const bf16* x; int lda;
  const bf16* stream;      // [N / 160][K / 64 stages][L1_W_STAGE]
  const float* bias;       // [N] or null
  const bf16* res; int ldr;   // residual [M][ldr] or null (may alias out)
  bf16* out; int ldo;
  int M, N, K;
  const float* ln_c; float ln_eps;   // GLN variant: LayerNorm folded (W' = gamma W, ln_c[n] = sum_k W'[n][k], bias holds beta . W + b); W rows (16 value | 16 gate)-interleaved
  int norot;               // 1: every workgroup walks the stages from stage 0 (results independent of the row position: NR_DETERMINISTIC_BATCH)
};

// GLN: the LayerNorm-folded GEGLU projection (FeedForward.net[0] behind norm3, motion_module_new.py:441-518): row statistics from the fragments that pass
// (a wave owns whole rows: BM = 128 only), epilogue out[m][80 cb + 16 i + ..] = v_i gelu(g_i) on the (value, gate) tile pairs a lane holds anyway
// MODE 0: plain Linear (+ bias, + residual); 1 (GLN): LayerNorm folded + GEGLU (BM = 128).  (A LayerNorm-folded PLAIN mode for the q|k|v projection at
// M = 512 -- 8 x 24 workgroups of 64 rows -- was built and measured: it costs the keyframe step 0.14 ms against the 128 x 64 igemm tile; removed.)
template <int BM, int MODE = 0>
__global__ __launch_bounds__(512) void lin160_kernel(NrLin160Params p) {
  constexpr bool GLN = MODE == 1;
  static_assert(!GLN || BM == 128, "the GEGLU variant pairs the (value, gate) tiles inside one wave");
  constexpr int RT = BM / 16;                    // row tiles: 8 / 4
  constexpr int NTW = BM == 128 ? L1_NT : L1_NT / 2;   // column tiles per wave: 10 / 5
  constexpr int A_STAGE = BM * 128;              // rows x 2 k-steps x 64 B: 16 / 8 KiB
  constexpr int A_PW = BM == 128 ? 2 : 1;        // row pieces per wave and stage
  constexpr int STAGE = L1_W_STAGE + A_STAGE;    // 40 / 32 KiB
  constexpr int PPW = 3 + A_PW;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // L1_NS stages

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int rt = BM == 128 ? wave : (wave & 3);          // this wave's row tile
  const int n0 = BM == 128 ? 0 : (wave >> 2) * NTW;     // its first column tile

  L1_STAMP(0);
  const int NCB = p.N / L1_BN, S = p.K >> 6;
  const int nrg = p.M / BM;
  int rg, cb;
  if ((nrg & 7) == 0) { const int j = blockIdx.x >> 3; cb = j % NCB; rg = (j / NCB) * 8 + (int)(blockIdx.x & 7); }   // the column blocks of a row group share an XCD
  else { cb = (int)(blockIdx.x % NCB); rg = blockIdx.x / NCB; }
  const int r0 = rg * BM;
  const int rot = p.norot ? 0 : rg % S;

  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(lptr_t)smem);
  const char* wsrc = reinterpret_cast<const char*>(p.stream) + (size_t)cb * ((size_t)S * L1_W_STAGE) + (size_t)(wave * 3) * 1024 + (size_t)lane * 16;
  // row pieces: piece (tile, kk) = 16 rows x 64 B of k-step kk; BM = 128: wave w fetches (w, 0) and (w, 1); BM = 64: wave w fetches (w & 3, w >> 2)
  const bf16* arow;
  {
    const int r = lane >> 2;
    arow = p.x + (size_t)(r0 + 16 * rt + r) * p.lda + (((lane & 3) ^ ((-(r >> 2)) & 3)) << 3);
  }
  auto issue_piece = [&](int s, int slot, int i) {
    const unsigned dst = lds0 + (unsigned)(slot * STAGE);
    int st = s + rot; if (st >= S) st -= S;
    if (i < 3) glds16(wsrc + (size_t)st * L1_W_STAGE + (size_t)i * 1024, dst + (unsigned)((wave * 3 + i) * 1024));
    else {
      const int kk = BM == 128 ? i - 3 : (wave >> 2);
      glds16(arow + 64 * st + 32 * kk, dst + (unsigned)(L1_W_STAGE + (kk * RT + rt) * 1024));
    }
  };
#pragma unroll
  for (int s = 0; s < L1_NS - 1; ++s)
#pragma unroll
    for (int i = 0; i < PPW; ++i) issue_piece(s, s, i);

  L1_STAMP(1);
  f32x4 acc[NTW];
#pragma unroll
  for (int n = 0; n < NTW; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
  float s1 = 0.f, s2 = 0.f;
  const bf16x2 one2 = {(bf16)1.0f, (bf16)1.0f};
  const unsigned wl = (unsigned)(n0 * 1024 + lane * 16);
  const unsigned al = (unsigned)(L1_W_STAGE + rt * 1024 + fr * 64 + ((fg ^ ((-(fr >> 2)) & 3)) << 4));

  int slot = 0;
  for (int s = 0; s < S; ++s) {
    if (s + 1 < S) wait_vmcnt<(L1_NS - 2) * PPW>(); else wait_vmcnt<0>();      // in flight behind stage s: the one stage issued after it
    if (s < 40) L1_STAMP(2 + 3 * s);
    __builtin_amdgcn_s_barrier();             // every wave's pieces landed; every wave has left stage s - 1 (its slot may be refilled)
    if (s < 40) L1_STAMP(3 + 3 * s);
    const int s_next = s + L1_NS - 1;
    const bool pf = s_next < S;
    int pslot = slot + L1_NS - 1; if (pslot >= L1_NS) pslot -= L1_NS;
    const unsigned char* base = smem + slot * STAGE;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const bf16x8 xa = *(const bf16x8*)(base + al + kk * (RT * 1024));
      if constexpr (GLN) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bf16x2 pr = {xa[2 * e], xa[2 * e + 1]};
          s1 = __builtin_amdgcn_fdot2_f32_bf16(pr, one2, s1, false);
          s2 = __builtin_amdgcn_fdot2_f32_bf16(pr, pr, s2, false);
        }
      }
      bf16x8 w[NTW];
#pragma unroll
      for (int n = 0; n < NTW; ++n) w[n] = *(const bf16x8*)(base + (unsigned)((kk * L1_NT + n) * 1024) + wl);
      if (pf) {
        if (kk == 0) { issue_piece(s_next, pslot, 0); issue_piece(s_next, pslot, 1); }
        else { issue_piece(s_next, pslot, 2); issue_piece(s_next, pslot, 3); if (A_PW == 2) issue_piece(s_next, pslot, 4); }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int n = 0; n < NTW; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[n], xa, acc[n], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (s < 40) L1_STAMP(4 + 3 * s);
    slot = slot + 1 == L1_NS ? 0 : slot + 1;
  }
  L1_STAMP(125);

  if constexpr (GLN) {
    auto rows_sum = [](float v) {
      auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
      v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
      auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
      return __uint_as_float(b[0]) + __uint_as_float(b[1]);
    };
    const float inv_k = 1.0f / (float)p.K;
    const float mu = rows_sum(s1) * inv_k;
    const float rstd = rsqrtf(fmaxf(rows_sum(s2) * inv_k - mu * mu, 0.f) + p.ln_eps);
    const int row = r0 + 16 * rt + fr;
    // ---- LayerNorm fold + GEGLU: tiles (2 i, 2 i + 1) of the block are the values / gates of output columns 80 cb + 16 i .. ----
    const int nb = cb * L1_BN + 4 * fg;
    bf16* orow = p.out + (size_t)row * p.ldo + cb * (L1_BN / 2) + 4 * fg;
#pragma unroll
    for (int i = 0; i < L1_NT / 2; ++i) {
      const int nv = nb + 32 * i, ng = nv + 16;
      const f32x4 cv = *(const f32x4*)(p.ln_c + nv), cg = *(const f32x4*)(p.ln_c + ng);
      const f32x4 bv = *(const f32x4*)(p.bias + nv), bg = *(const f32x4*)(p.bias + ng);
      bf16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = (acc[2 * i][e] - mu * cv[e]) * rstd + bv[e];
        const float g = (acc[2 * i + 1][e] - mu * cg[e]) * rstd + bg[e];
        o[e] = (bf16)(v * gelu_erf_fast(g));
      }
      nr_store8(orow + 16 * i, o);
    }
    L1_STAMP(126);
    return;
  }
  // ---- epilogue: lane holds out[r0 + 16 rt + fr][cb 160 + 16 (n0 + n) + 4 fg .. + 3]: 8 bytes per lane in 32-byte row segments.  v_permlane16_swap between the
  // column tiles (2 k, 2 k + 1) hands every lane 8 CONSECUTIVE channels (even lane rows: tile 2 k, channels 4 fg .. 4 fg + 7; odd rows: tile 2 k + 1,
  // channels 4 (fg - 1) ..): 16-byte residual loads and stores in 64-byte row segments (as tattn.hip); an odd last tile keeps the 8-byte form ----
  const int row = r0 + 16 * rt + fr;
  const int cbase = cb * L1_BN + 16 * n0;
  bf16* orow = p.out + (size_t)row * p.ldo + cbase;
  const bf16* rrow = p.res ? p.res + (size_t)row * p.ldr + cbase : nullptr;
#pragma unroll
  for (int k = 0; k < NTW / 2; ++k) {
    f32x4 lo = acc[2 * k], hi = acc[2 * k + 1];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(lo[e]), __float_as_uint(hi[e]), false, false);
      lo[e] = __uint_as_float(sw[0]); hi[e] = __uint_as_float(sw[1]);
    }
    const int c = 16 * (2 * k + (fg & 1)) + 4 * (fg & 2);
    if (p.bias) { lo += *(const f32x4*)(p.bias + cbase + c); hi += *(const f32x4*)(p.bias + cbase + c + 4); }
    if (rrow) {
      const bf16x8 r = *(const bf16x8*)(rrow + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) { lo[e] += (float)r[e]; hi[e] += (float)r[4 + e]; }
    }
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[e] = (bf16)lo[e]; o[4 + e] = (bf16)hi[e]; }
    nr_store16(orow + c, o);
  }
  if constexpr (NTW % 2 == 1) {
    constexpr int n = NTW - 1;
    const int c = 16 * n + 4 * fg;
    f32x4 v = acc[n];
    if (p.bias) v += *(const f32x4*)(p.bias + cbase + c);
    if (rrow) {
      const bf16x4 r = *(const bf16x4*)(rrow + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += (float)r[e];
    }
    bf16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (bf16)v[e];
    nr_store8(orow + c, o);
  }
  L1_STAMP(126);
}

// fragment-major stream from the row-major [N][K] bf16 matrix: chunk -> (column block, stage, k-step kk, fragment n, lane)
__global__ __launch_bounds__(256) void lin160_w_pack_kernel(const bf16* __restrict__ w, bf16* __restrict__ stream, int N, int K) {
  const int S = K >> 6, NCB = N / L1_BN, CH_STAGE = L1_W_STAGE / 16;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)NCB * S * CH_STAGE) return;
  const int cb = (int)(idx / ((long long)S * CH_STAGE));
  int c = (int)(idx - (long long)cb * S * CH_STAGE);
  const int st = c / CH_STAGE;
  c -= st * CH_STAGE;
  bf16x8 v = bf16x8_zero();
  if (c < 2 * L1_NT * 64) {                                    // the tail of a stage is padding
    const int kk = c / (L1_NT * 64), n = (c / 64) % L1_NT, lane = c & 63;
    v = *(const bf16x8*)(w + (size_t)(cb * L1_BN + 16 * n + (lane & 15)) * K + 64 * st + 32 * kk + 8 * (lane >> 4));
  }
  *(bf16x8*)(stream + (size_t)idx * 8) = v;
}


// ---------------------------------------------------------------------------------------------------------------------------------------------
// Row-PANEL form for the LayerNorm-folded wide projections of the C = 640 level (GEGLU projection N = 8 C, fused q|k|v N = 3 C; K = C) on 2048 .. 8192 rows.
// What bounds lin160_kernel on these shapes is the LDS read port, not the MFMAs: a wave that owns 16 rows reads one 1-KiB W fragment per MFMA
// (8 waves x 20 fragments = 160 KiB per 64-channel stage = 640 cycles at 256 B/clk against 640 MFMA cycles per SIMD), every workgroup pays prologue and
// epilogue per 160 columns, and x travels through the ring beside W.  Here:
//   * the workgroup's 128 rows stay in REGISTERS for the whole launch: wave w holds rows 32 (w & 3) .. + 31 as MFMA B fragments for all of K (2 x 20 x 4 =
//     160 VGPRs) and computes column half w >> 2 of every 128-column block -- each W fragment read from LDS feeds TWO MFMAs (half the LDS bytes per flop);
//   * only W streams: J consecutive 128-column blocks per workgroup (column group = blockIdx mod NCG, so the workgroups of an XCD share their J blocks in its
//     L2; x is read once and needs no locality) as 32-KiB stages of 128 channels (4 k-steps x 8 fragments, no padding; 4 one-KiB LDS-DMA pieces per wave)
//     through a 4-slot ring that never drains between blocks; every row group starts each block at its own stage (L2 channel spread): the register panel
//     is loaded in that rotated k order, the unrolled MFMA loop indexes it statically;
//   * fragment reads run one k-step ahead of the MFMAs (two register banks), across the stage barrier too: the barrier of stage g + 1 sits in front of the
//     LAST k-step of stage g, whose fragments are in registers by then;
//   * LayerNorm statistics from the register panel on the matrix unit (x . 1 and the diagonal of x . x^T), once; fold constants (ln_c | bias') of the workgroup's columns in an LDS table;
//   * block epilogue: out = rstd (acc - mean c) + b' (GEGLU: v . gelu(g) on the (value, gate) tile pairs a wave holds).
// KS = K / 32 (20).  Grid (M / 128) x NCG, one workgroup per CU.
// ---------------------------------------------------------------------------------------------------------------------------------------------
constexpr int Q_BN = 128, Q_NT = 8;
constexpr int Q_STAGE = 32 * 1024;               // 128 channels: 4 k-steps x 8 fragments of 1 KiB
constexpr int Q_NS = 4;                          // ring slots (a power of two)
// (Built, measured and removed: the two column halves of a workgroup TWO STAGES APART -- ring buffer g carrying tiles 0 .. 3 of stream stage g and tiles 4 .. 7 of
// stage g - 2, so that the GEGLU epilogues of the two waves of a SIMD fall 40 % of a block apart, each under its mate's MFMAs: the MFMA wave lost more issue slots to
// its mate's VALU stream than the serial epilogue costs, launch 139 k -> 168 k cycles.)

struct NrLin128QParams {
  const bf16* x; int lda;
  const bf16* stream;      // [N / 128][K / 128][4 k-steps][8 tiles][64 lanes][8]
  const float* ln_c; const float* bias; float ln_eps;
  bf16* out; int ldo;
  int M, N, J, NCG;
  int norot;               // 1: every workgroup walks the stages from stage 0 (NR_DETERMINISTIC_BATCH)
  int cgmajor;             // 1: workgroup -> XCD by column group (blockIdx mod NCG) instead of by row group
};

// KSPLIT (K = 1280): a wave cannot hold 32 rows x 1280 channels (320 VGPRs), so the two waves of a SIMD split K instead of the columns: wave (rp, ch) holds the
// k-steps 4 ch .. 4 ch + 3 of every 256-channel stage of its 32 rows (160 VGPRs again, and no row is loaded twice), the column block is 64 wide (4 tiles: both
// waves accumulate all of them over their half of K), a stage is still 32 KiB (8 k-steps x 4 fragments), the ring has 3 slots, and at the end of a block the pair
// swaps partial sums through 32 KiB of LDS (each wave hands over the two tiles the other one finishes, one extra barrier per block); the row statistics are
// combined the same way once.
template <int KS, bool GEGLU, bool KSPLIT = false, int DBG = 0>
__global__ __launch_bounds__(512) void lin128q_kernel(NrLin128QParams p) {
  constexpr int K = 32 * KS;
  constexpr int KSL = KSPLIT ? KS / 2 : KS;            // k-steps a wave holds
  constexpr int KPS = KSPLIT ? 8 : 4;                  // k-steps per stage
  constexpr int S = KS / KPS;                          // stages per column block
  constexpr int BN = KSPLIT ? 64 : Q_BN, NTB = BN / 16;   // column block, its tiles
  constexpr int NS = KSPLIT ? 3 : Q_NS;                // ring slots
  constexpr int XCH = KSPLIT ? 32 * 1024 : 0;          // partial-sum exchange behind the ring
  static_assert(KSL == 20 && S == 5, "20 k-steps per wave in 5 stages of 4");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // NS stages of Q_STAGE bytes, [the exchange area,] then float tab[2][J * BN] (ln_c | bias')
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int rp = wave & 3, ch = wave >> 2;             // ch: column half (tiles 4 ch ..) or, KSPLIT, K half (k-steps 4 ch .. of every stage)
  L1_STAMP(0);
  const int J = p.J;
  // workgroup -> (row group, column group): the column groups of a row group on ONE XCD (blockIdx mod 8), so x (read once per workgroup, all in the prologue)
  // crosses the fabric once; every XCD then streams all of W, block by block in step with its row groups (working set: NCG blocks of 160 KiB)
  // (few row groups and many column groups -- the 8 x 8 level at one clip: 16 x 16 -- turn it round, host flag: the column groups of an XCD stay in its L2 and x crosses
  // the fabric eight times: 68 MB instead of 215 MB per launch for the C = 1280 GEGLU projection)
  int rg, cg;
  const int nrg = p.M >> 7;
  if (!p.cgmajor && (nrg & 7) == 0) { const int i = blockIdx.x >> 3; cg = i % p.NCG; rg = (i / p.NCG) * 8 + (int)(blockIdx.x & 7); }
  else { cg = (int)(blockIdx.x % p.NCG); rg = (int)(blockIdx.x / p.NCG); }
  const int row0 = rg * 128 + 32 * rp + fr;            // + 16 rt
  const int rot = p.norot ? 0 : rg % S;                // first stage of every block for this row group
  const int G = J * S;                                 // stages of this workgroup (>= NS - 1: host)

  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(lptr_t)smem);
  const char* wbase = reinterpret_cast<const char*>(p.stream) + (size_t)cg * J * ((size_t)S * Q_STAGE) + (size_t)(wave * 4) * 1024 + (size_t)lane * 16;
  auto issue_piece = [&](int g, int i) {               // piece i (0 .. 3) of this wave for stage g -> slot g mod NS
    const int gs = g < G ? g : G - 1;                  // behind the last stage nobody consumes the piece: a valid source keeps the loop branch-free and its vmcnt constant
    const int j = gs / S;
    int st = gs - j * S + rot; if (st >= S) st -= S;
    glds16(wbase + ((size_t)j * S + st) * Q_STAGE + (size_t)i * 1024, lds0 + (unsigned)((g % NS) * Q_STAGE + (wave * 4 + i) * 1024));
  };
#pragma unroll
  for (int g = 0; g < NS - 1; ++g)
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_piece(g, i);

  L1_STAMP(122);
  // ---- fold constants of the workgroup's J x BN columns -> LDS table ----
  float* tab = reinterpret_cast<float*>(smem + NS * Q_STAGE + XCH);
  for (int i = tid; i < J * BN; i += 512) {
    tab[i] = p.ln_c[cg * J * BN + i];
    tab[J * BN + i] = p.bias[cg * J * BN + i];
  }

  L1_STAMP(123);
  // ---- the row panel -> registers: register k-step 4 st + kk = channel k-step KPS ((st + rot) mod S) + [4 ch +] kk of the row.  Issued BEHIND the prologue's
  // DMA pieces: the compiler's own vmcnt for these loads does not know the asm pieces, so they must be the older operations (its waits are then conservative) ----
  bf16x8 xb[2][KSL];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    const bf16* xr = p.x + (size_t)(row0 + 16 * rt) * p.lda + 8 * fg;
#pragma unroll
    for (int ks = 0; ks < KSL; ++ks) {
      int sq = (ks >> 2) + rot; if (sq >= S) sq -= S;
      const int kq = KPS * sq + (KSPLIT ? 4 * ch : 0) + (ks & 3);
      xb[rt][ks] = *(const bf16x8*)(xr + 32 * kq);
    }
  }
  auto rows_sum = [](float v) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
  };
  L1_STAMP(124);
  // row statistics on the MATRIX unit, one pass over the register panel: with A = ones, D[i][r] = sum_k x[r][k] for every i (each lane gets its row's sum);
  // with A = the fragment itself (the A and B register images of v_mfma_f32_16x16x32_bf16 coincide), D[i][r] = x_i . x_r, whose diagonal is sum_k x[r][k]^2 --
  // 80 MFMAs per wave (1.3 k cycles) where cvt / add / fma chains cost 2 560 VALU instructions and the dot unit 640 (two waves per SIMD: 20 k / 10 k cycles of prologue)
  float mu[2], rstd[2];
  {
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;
    float sx[2], sq2[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      f32x4 a1 = {0.f, 0.f, 0.f, 0.f}, a2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KSL; ++ks) {
        a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, xb[rt][ks], a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xb[rt][ks], xb[rt][ks], a2, 0, 0, 0);
      }
      // lane (fr, fg) holds D[4 fg + e][fr]: the diagonal element of row fr sits in lane group fg = fr / 4 at e = fr mod 4
      const int e_d = fr & 3;
      float d = e_d == 0 ? a2[0] : e_d == 1 ? a2[1] : e_d == 2 ? a2[2] : a2[3];
      d = (fg == (fr >> 2)) ? d : 0.f;
      sx[rt] = a1[0];
      sq2[rt] = rows_sum(d);
    }
    if constexpr (KSPLIT) {                            // the other half of K lives in the SIMD-mate (wave ^ 4)
      float* xs = reinterpret_cast<float*>(smem + NS * Q_STAGE);          // [8 waves][2 rt][2][16 rows]
      if (fg == 0) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) { xs[((wave * 2 + rt) * 2 + 0) * 16 + fr] = sx[rt]; xs[((wave * 2 + rt) * 2 + 1) * 16 + fr] = sq2[rt]; }
      }
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) { sx[rt] += xs[(((wave ^ 4) * 2 + rt) * 2 + 0) * 16 + fr]; sq2[rt] += xs[(((wave ^ 4) * 2 + rt) * 2 + 1) * 16 + fr]; }
    }
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      mu[rt] = sx[rt] * (1.0f / K);
      rstd[rt] = rsqrtf(fmaxf(sq2[rt] * (1.0f / K) - mu[rt] * mu[rt], 0.f) + p.ln_eps);
    }
  }
  L1_STAMP(127);
  wait_vmcnt<0>();                                   // the panel is the youngest: stages 0 .. NS - 2 of this wave have landed
  __builtin_amdgcn_s_barrier();                      // ... of every wave, and the table (and, KSPLIT, everyone has read the statistics exchange)
  L1_STAMP(1);

  // this wave's first fragment of a k-step: column half ch (tiles 4 ch ..) / K half ch (k-steps 4 ch .. of the stage, all 4 tiles)
  const unsigned wl = (unsigned)((KSPLIT ? 16 * ch : 4 * ch) * 1024 + lane * 16);
  // element offset of this lane's first output of a block's column 0 (32 bits: M ldo < 2^31 elements, host-checked), uniform base + lane offset addressing;
  // the tiles a wave FINISHES: 4 ch .. 4 ch + 3 / KSPLIT 2 ch, 2 ch + 1
  const unsigned oofs = (unsigned)row0 * (unsigned)p.ldo + (unsigned)((KSPLIT ? (GEGLU ? 16 * ch : 32 * ch) : (GEGLU ? 32 * ch : 64 * ch)) + 4 * fg);
  bf16x8 w[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) w[n] = *(const bf16x8*)(smem + (unsigned)(n * 1024) + wl);
  f32x4 acc[2][4];
  // block epilogue: lane holds out channels 16 t + 4 fg .. + 3 of rows row0, row0 + 16 for the tiles t it finishes of block jb of the group
  auto epilogue = [&](int jb) {
    const int cb = cg * J + jb;
    constexpr int NF = KSPLIT ? 2 : 4;                 // tiles this wave finishes
    const float* tc = tab + jb * BN + 16 * (NF * ch) + 4 * fg;
    const float* tb = tc + J * BN;
    if constexpr (KSPLIT) {
      // the pair swaps partial sums: [wave][rt][nn][lane] f32x4 -- the two tiles the OTHER wave finishes go out, its sums of this wave's two tiles come in
      f32x4* ex = reinterpret_cast<f32x4*>(smem + NS * Q_STAGE);
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int nn = 0; nn < 2; ++nn) {
          f32x4 theirs;
#pragma unroll
          for (int e = 0; e < 4; ++e) theirs[e] = ch ? acc[rt][nn][e] : acc[rt][2 + nn][e];
          ex[((wave * 2 + rt) * 2 + nn) * 64 + lane] = theirs;
        }
      __builtin_amdgcn_s_barrier();
      const f32x4 c0 = *(const f32x4*)(tc), c1 = *(const f32x4*)(tc + 16), b0 = *(const f32x4*)(tb), b1 = *(const f32x4*)(tb + 16);
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const f32x4 o0 = ex[(((wave ^ 4) * 2 + rt) * 2 + 0) * 64 + lane], o1 = ex[(((wave ^ 4) * 2 + rt) * 2 + 1) * 64 + lane];
        bf16x4 r0, r1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float t0 = ((ch ? acc[rt][2][e] : acc[rt][0][e]) + o0[e] - mu[rt] * c0[e]) * rstd[rt] + b0[e];
          const float t1 = ((ch ? acc[rt][3][e] : acc[rt][1][e]) + o1[e] - mu[rt] * c1[e]) * rstd[rt] + b1[e];
          if constexpr (GEGLU) r0[e] = (bf16)(t0 * gelu_erf_fast(t1));
          else { r0[e] = (bf16)t0; r1[e] = (bf16)t1; }
        }
        bf16* orow = p.out + (oofs + (unsigned)(16 * rt) * (unsigned)p.ldo + (unsigned)(cb * (GEGLU ? BN / 2 : BN)));
        nr_store8(orow, r0);
        if constexpr (!GEGLU) nr_store8(orow + 16, r1);
      }
    } else if constexpr (GEGLU) {
#pragma unroll
      for (int i = 0; i < NF / 2; ++i) {
        const f32x4 cv = *(const f32x4*)(tc + 32 * i), cgv = *(const f32x4*)(tc + 32 * i + 16);
        const f32x4 bv = *(const f32x4*)(tb + 32 * i), bg = *(const f32x4*)(tb + 32 * i + 16);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float v = (acc[rt][2 * i][e] - mu[rt] * cv[e]) * rstd[rt] + bv[e];
            const float gt = (acc[rt][2 * i + 1][e] - mu[rt] * cgv[e]) * rstd[rt] + bg[e];
            o[e] = (bf16)(v * gelu_erf_fast(gt));
          }
          nr_store8(p.out + (oofs + (unsigned)(16 * rt) * (unsigned)p.ldo + (unsigned)(cb * (BN / 2) + 16 * i)), o);
        }
      }
    } else {
#pragma unroll
      for (int n = 0; n < NF; ++n) {
        const f32x4 c4 = *(const f32x4*)(tc + 16 * n), b4 = *(const f32x4*)(tb + 16 * n);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (bf16)((acc[rt][n][e] - mu[rt] * c4[e]) * rstd[rt] + b4[e]);
          nr_store8(p.out + (oofs + (unsigned)(16 * rt) * (unsigned)p.ldo + (unsigned)(cb * BN + 16 * n)), o);
        }
      }
    }
  };
  // The stage loop is STRAIGHT-LINE code: no branch inside a stage (hipcc loses its lgkmcnt bookkeeping at every join and waits for ALL fragment reads in front of the
  // next MFMA pair: a k-step cost 360 cycles for 256 of MFMAs).  Hence: the fragments of the next k-step are always read (behind the last stage: stale bytes nobody uses),
  // the DMA burst and the stage barrier are unconditional (pieces behind the last stage re-fetch its bytes), the first k-step of a block accumulates onto one zero quad.
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  int g = 0, slot = 0;
  for (int j = 0; j < J; ++j) {
#pragma unroll
    for (int st = 0; st < S; ++st) {
      const int nslot = slot + 1 == NS ? 0 : slot + 1;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        // ---- where the fragments of the NEXT k-step come from (kk = 3: the next stage, behind its barrier); w[n] is refilled as soon as its two MFMAs are issued ----
        const unsigned char* nbase;
        if (kk < 3) nbase = smem + slot * Q_STAGE + (kk + 1) * (NTB * 1024);
        else {
          if (!Q_DBG(16)) wait_vmcnt<(NS - 2) * 4>();            // stage g + 1 landed: younger pieces of this wave in flight = stages g + 2 .. g + NS - 1 (4 each)
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave's reads of slot g are in registers: the slot may be refilled behind the barrier
          if (g < 40) L1_STAMP(2 + 3 * g);
          if (!Q_DBG(1)) __builtin_amdgcn_s_barrier();
          if (g < 40) L1_STAMP(3 + 3 * g);
          nbase = smem + nslot * Q_STAGE;
        }
        __builtin_amdgcn_sched_barrier(0);
        // the wave's four DMA pieces of stage g + NS - 1 in one burst at the head of the stage (staggering the two waves of a SIMD measured the same)
        if (kk == 0 && !Q_DBG(2)) {
#pragma unroll
          for (int i = 0; i < 4; ++i) issue_piece(g + NS - 1, i);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < 4; ++n) {
#pragma unroll
          for (int rt = 0; rt < 2; ++rt)
            if (!Q_DBG(8)) acc[rt][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[n], xb[rt][4 * st + kk], (st == 0 && kk == 0) ? zero4 : acc[rt][n], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (!Q_DBG(4)) w[n] = *(const bf16x8*)(nbase + (unsigned)(n * 1024) + wl);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (g < 40) L1_STAMP(4 + 3 * g);
      ++g;
      slot = nslot;
    }
    if (!Q_DBG(32)) epilogue(j);
    if constexpr (KSPLIT) {
      // the first fragments of the next block are read again BEHIND the epilogue: their registers are free across it (the exchange needs them: with the prefetched
      // set live hipcc spills panel registers and rotates the whole panel by four registers per block, 140 v_mov_b64)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int n = 0; n < 4; ++n) w[n] = *(const bf16x8*)(smem + slot * Q_STAGE + (unsigned)(n * 1024) + wl);
    }
  }
  L1_STAMP(125);
  wait_vmcnt<0>();                                   // the pieces issued behind the last stage still write this workgroup's LDS
  L1_STAMP(126);
}

// fragment-major stream of the panel kernel from the row-major [N][K] bf16 matrix: 16-byte chunk -> (column block of BN, stage of 32 KPS channels, k-step, tile, lane);
// a stage is Q_STAGE bytes for both forms (BN = 128, KPS = 4 / KSPLIT: BN = 64, KPS = 8)
__global__ __launch_bounds__(256) void lin128q_w_pack_kernel(const bf16* __restrict__ w, bf16* __restrict__ stream, int N, int K, int BN, int KPS) {
  const int S = K / (32 * KPS), CH_STAGE = Q_STAGE / 16, NTB = BN / 16;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)(N / BN) * S * CH_STAGE) return;
  const int cb = (int)(idx / ((long long)S * CH_STAGE));
  int c = (int)(idx - (long long)cb * S * CH_STAGE);
  const int st = c / CH_STAGE;
  c -= st * CH_STAGE;
  const int kk = c / (NTB * 64), n = (c / 64) % NTB, lane = c & 63;
  *(bf16x8*)(stream + (size_t)idx * 8) = *(const bf16x8*)(w + (size_t)(cb * BN + 16 * n + (lane & 15)) * K + 32 * (KPS * st + kk) + 8 * (lane >> 4));
}

unsigned long long g_l1_attr = 0;

}  // namespace

#ifdef NR_STAMP
extern "C" int nr_lin160_stamp_read(void* dst, size_t bytes, int clear) {
  const size_t n = bytes < sizeof(lin160_stamp_buf) ? bytes : sizeof(lin160_stamp_buf);
  int rc = 0;
  if (dst) rc = (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(lin160_stamp_buf), n, 0, hipMemcpyDeviceToHost);
  if (clear) { void* d = nullptr; (void)hipGetSymbolAddress(&d, HIP_SYMBOL(lin160_stamp_buf)); (void)hipMemset(d, 0, sizeof(lin160_stamp_buf)); }
  return rc;
}
#endif

extern "C" size_t nr_lin160_stream_bytes(int N, int K) { return (N % L1_BN == 0 && K % 64 == 0) ? (size_t)(N / L1_BN) * (K / 64) * L1_W_STAGE : 0; }

// The shapes this kernel is chosen for: plain Linear (one source, no GEGLU / LayerNorm fold / row vector / activation / scale), K = 640 or 1280
// (the short-K regime), N a multiple of 160, >= 2048 rows in whole 64-row groups
// PANEL rule (rows of one clip, N, K): the LayerNorm-folded wide projections (GEGLU N = 8 C, q|k|v N = 3 C) at K = C = 640 (>= 4096 rows) / 1280 (>= 2048 rows) (NR_LIN160_PANEL_MAXM: sweep aid;
// NR_LIN160_PANEL_K1280=0: the K-split form of the C = 1280 level off)
extern "C" int nr_lin160_panel_rule(int Mp, int N, int K) {
  static const bool off = (getenv("NR_LIN160") && getenv("NR_LIN160")[0] == '0') || (getenv("NR_LIN160_PANEL") && getenv("NR_LIN160_PANEL")[0] == '0');   // A/B switches
  static const bool off1280 = getenv("NR_LIN160_PANEL_K1280") && getenv("NR_LIN160_PANEL_K1280")[0] == '0';
  static const int maxm = getenv("NR_LIN160_PANEL_MAXM") ? atoi(getenv("NR_LIN160_PANEL_MAXM")) : (1 << 30);   // no row ceiling: J = 10 blocks per workgroup amortise prologue and epilogue over any number of rounds (config 4: +1.6 %)
  if (off || Mp < 2048 || Mp > maxm || N < 3 * K) return 0;
  // K = 640 needs >= 32 row groups: at 2048 rows (the sgm keyframe model's 32 x 32 level) the best partition has 160 workgroups of 20 stages and the prologue dominates
  // (keyframe 10.55 -> 10.63 ms per Euler step with it, profiles/r06_lin160_panel_ab.txt); K = 1280 fills the chip from 2048 rows on (16 row groups x 16 column groups)
  if (K == 640) return Mp >= 4096 && N % Q_BN == 0;
  if (K == 1280) return !off1280 && N % 64 == 0;
  return 0;
}
extern "C" size_t nr_lin128q_stream_bytes(int N, int K) {
  if (K == 640) return N % Q_BN == 0 ? (size_t)N * K * sizeof(bf16) : 0;
  if (K == 1280) return N % 64 == 0 ? (size_t)N * K * sizeof(bf16) : 0;
  return 0;
}
extern "C" int nr_launch_lin128q_w_pack(const bf16* w, int N, int K, bf16* stream, hipStream_t s) {
  const long long total = (long long)(nr_lin128q_stream_bytes(N, K) / 16);
  if (!total) return 1;
  hipLaunchKernelGGL(lin128q_w_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, stream, N, K, K == 1280 ? 64 : Q_BN, K == 1280 ? 8 : 4);
  return 0;
}

extern "C" int nr_lin160_eligible(const NrGemmParams* pp) {
  static const bool off = getenv("NR_LIN160") && getenv("NR_LIN160")[0] == '0';   // A/B switch
  const NrGemmParams& p = *pp;
  // PANEL: register-resident rows, W streamed
  {
    const int Mq = (p.plan_m > 0 && p.plan_m < p.M) ? p.plan_m : p.M;
    if (p.ln_c && p.bias && p.ksize == 1 && p.stride == 1 && !p.ups && !p.a1 && !p.c1 && !p.rowvec && !p.act && !p.out_f32 && !p.res &&
        p.out_scale == 1.0f && p.K == p.c0 && nr_lin160_panel_rule(Mq, p.N, p.K) && p.M % 128 == 0 && p.lda0 % 8 == 0 && p.ldo % 4 == 0 &&
        (long long)p.M * p.ldo < (1ll << 31))          // 32-bit output offsets in the kernel: beyond that the tiled igemm serves the shape
      return 4;
  }
  // GLN: the LayerNorm-folded GEGLU projection on FEW rows (the keyframe model's depth-10 levels and the 4 x 4 level of the headline: M = 512, N = 10240,
  // K = 1280): 4 x 64 workgroups = one round of the chip, each streams its 160 W rows once for 128 rows (tiled igemm: 27-32 us)
  {
    static const bool gln_off = getenv("NR_LIN160_GEGLU") && getenv("NR_LIN160_GEGLU")[0] == '0';
    const int Mg = (p.plan_m > 0 && p.plan_m < p.M) ? p.plan_m : p.M;
    if (!off && !gln_off && p.geglu && p.ln_c && p.bias && p.ksize == 1 && p.stride == 1 && !p.ups && !p.a1 && !p.c1 && !p.rowvec && !p.act && !p.out_f32 && !p.res &&
        p.out_scale == 1.0f && p.K == p.c0 && p.K == 1280 && p.N % L1_BN == 0 && p.N >= 8192 && p.M % 128 == 0 && Mg <= 1024 && (long long)(Mg / 128) * (p.N / L1_BN) >= 192 &&
        p.lda0 % 8 == 0 && p.ldo % 4 == 0)
      return 2;
  }
  if (off || p.ksize != 1 || p.stride != 1 || p.ups || p.a1 || p.c1 || p.geglu || p.ln_c || p.rowvec || p.act || p.out_f32 || p.tap_inner) return 0;
  // K = 640 / 1280 only: the long-K folded net.2 | proj_out operand (K = 3200, built as a two-source variant and measured: 54.8 vs 55 us) gains nothing
  if (p.out_scale != 1.0f || p.K != p.c0 || (p.K != 640 && p.K != 1280) || p.N % L1_BN != 0 || p.N > 1280) return 0;
  const int Mp = (p.plan_m > 0 && p.plan_m < p.M) ? p.plan_m : p.M;                 // NR_DETERMINISTIC_BATCH: the choice is made per clip
  if (p.M % 64 != 0 || Mp < 2048) return 0;
  if (p.lda0 % 8 != 0 || p.ldo % 8 != 0 || (p.res && p.ldr % 8 != 0)) return 0;
  if (Mp > 8192) return 0;           // one round of the chip at the headline shapes; beyond it (config 4: M = 16384 / 65536) the tiled igemm with its 2-3 resident workgroups per CU wins (profiles/r06_lin160_ab.txt)
  return 1;
}

extern "C" int nr_launch_lin160_w_pack(const bf16* w, int N, int K, bf16* stream, hipStream_t s) {
  const long long total = (long long)(nr_lin160_stream_bytes(N, K) / 16);
  if (!total) return 1;
  hipLaunchKernelGGL(lin160_w_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, stream, N, K);
  return 0;
}

extern "C" int nr_launch_lin160(const NrGemmParams* pp, const bf16* stream, hipStream_t s) {
  const NrGemmParams& g = *pp;
  if (!stream) return 1;
  const int Mp = (g.plan_m > 0 && g.plan_m < g.M) ? g.plan_m : g.M;
  if (g.ln_c && nr_lin160_panel_rule(Mp, g.N, g.K)) {            // PANEL form (stream = the nr_lin128q layout)
    if (!g.bias || g.M % 128 != 0 || (long long)g.M * g.ldo >= (1ll << 31)) return 1;
    const bool ksplit = g.K == 1280;                               // K = 1280: the two waves of a SIMD split K, 64-column blocks
    NrLin128QParams q;
    q.x = g.a0; q.lda = g.lda0; q.stream = stream; q.ln_c = g.ln_c; q.bias = g.bias; q.ln_eps = g.ln_eps; q.out = g.out; q.ldo = g.ldo; q.M = g.M; q.N = g.N; q.norot = g.plan_m > 0 ? 1 : 0;
    // J column blocks per workgroup (NCG = ncb / J column groups): the divisor of ncb with the fewest stage-times for the launch -- rounds of the chip x (J S stages + ~6
    // stage-times of prologue / epilogue); ties -> the larger J
    const int bn = ksplit ? 64 : Q_BN, ncq = g.N / bn, S = 5, nrg = g.M / 128, ns = ksplit ? 3 : Q_NS;
    static const int j_force = getenv("NR_LIN160_PANEL_J") ? atoi(getenv("NR_LIN160_PANEL_J")) : 0;   // sweep aid
    int J = 0; long long best = 0;
    for (int c = 1; c <= 16 && c <= ncq; ++c) {
      if (ncq % c != 0 || c * S < ns - 1) continue;
      const long long wgs = (long long)nrg * (ncq / c), cost = ((wgs + 255) / 256) * (c * S + 6);
      if (!J || cost <= best) { J = c; best = cost; }
    }
    if (j_force > 0 && ncq % j_force == 0 && j_force <= 16) J = j_force;
    if (!J) return 1;
    q.J = J; q.NCG = ncq / J;
    {
      // workgroup -> XCD: by row group (every XCD streams all of W, x crosses the fabric once) or by column group (W once or 8 / NCG times, x NCG-or-8 times): the
      // smaller fabric traffic; by column group only where blockIdx mod 8 fixes the column group
      static const int cg_force = getenv("NR_LIN160_PANEL_CGMAJOR") ? atoi(getenv("NR_LIN160_PANEL_CGMAJOR")) : -1;   // A/B aid
      const double wb = 2.0 * g.N * (double)g.K, xb = 2.0 * g.M * (double)g.K;
      const bool can = q.NCG % 8 == 0 || 8 % q.NCG == 0;
      const double by_rg = (nrg % 8 == 0 ? 8.0 : 8.0) * wb + xb;
      const double by_cg = (q.NCG % 8 == 0 ? 1.0 : 8.0 / q.NCG) * wb + (q.NCG % 8 == 0 ? 8.0 : (double)q.NCG) * xb;
      q.cgmajor = (can && by_cg < by_rg) ? 1 : 0;
      if (cg_force >= 0) q.cgmajor = cg_force && can;
    }
    const size_t shm_max = (size_t)Q_NS * Q_STAGE + 2 * 16 * Q_BN * sizeof(float);      // either form: ring [+ exchange] + the table of 16 blocks
    const size_t shm = (size_t)ns * Q_STAGE + (ksplit ? 32 * 1024 : 0) + (size_t)2 * J * bn * sizeof(float);
    int dev = 0;
    (void)hipGetDevice(&dev);
    static unsigned long long done = 0;
    if (!(done >> (dev & 63) & 1ull)) {
      const void* ks[4] = {(const void*)lin128q_kernel<20, true>, (const void*)lin128q_kernel<20, false>, (const void*)lin128q_kernel<40, true, true>, (const void*)lin128q_kernel<40, false, true>};
      for (const void* kf : ks) if (hipFuncSetAttribute(kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm_max) != hipSuccess) return 2;
      done |= 1ull << (dev & 63);
    }
    const dim3 grid((unsigned)(nrg * q.NCG));
#ifdef NR_STAMP
    if (const int bits = (!ksplit && getenv("NR_Q_DBG")) ? atoi(getenv("NR_Q_DBG")) : 0) {      // timing-only ablation arms of the GEGLU form
      auto go = [&](auto kf) { (void)hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm_max); hipLaunchKernelGGL(kf, grid, dim3(512), shm, s, q); };
      switch (bits) {
        case 1: go(lin128q_kernel<20, true, false, 1>); break;
        case 2: go(lin128q_kernel<20, true, false, 2>); break;
        case 4: go(lin128q_kernel<20, true, false, 4>); break;
        case 8: go(lin128q_kernel<20, true, false, 8>); break;
        case 14: go(lin128q_kernel<20, true, false, 14>); break;
        case 16: go(lin128q_kernel<20, true, false, 16>); break;
        case 17: go(lin128q_kernel<20, true, false, 17>); break;
        default: return 1;
      }
      return 0;
    }
#endif
    if (ksplit) {
      if (g.geglu) hipLaunchKernelGGL((lin128q_kernel<40, true, true>), grid, dim3(512), shm, s, q);
      else hipLaunchKernelGGL((lin128q_kernel<40, false, true>), grid, dim3(512), shm, s, q);
    } else {
      if (g.geglu) hipLaunchKernelGGL((lin128q_kernel<20, true>), grid, dim3(512), shm, s, q);
      else hipLaunchKernelGGL((lin128q_kernel<20, false>), grid, dim3(512), shm, s, q);
    }
    return 0;
  }
  if (g.M % 64 != 0 || g.N % L1_BN != 0 || g.K % 64 != 0 || g.K / 64 < L1_NS) return 1;
  NrLin160Params p;
  p.x = g.a0; p.lda = g.lda0; p.stream = stream; p.bias = g.bias; p.res = g.res; p.ldr = g.ldr; p.out = g.out; p.ldo = g.ldo; p.M = g.M; p.N = g.N; p.K = g.K; p.norot = g.plan_m > 0 ? 1 : 0;
  const int ncb = g.N / L1_BN;
  p.ln_c = g.ln_c; p.ln_eps = g.ln_eps;
  if (g.geglu) {
    if (!g.ln_c || !g.bias || g.M % 128 != 0) return 1;
    constexpr size_t shm = (size_t)L1_NS * (L1_W_STAGE + 128 * 128);
    int dev = 0;
    (void)hipGetDevice(&dev);
    static unsigned long long done = 0;
    if (!(done >> (dev & 63) & 1ull)) {
      if (hipFuncSetAttribute((const void*)lin160_kernel<128, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess) return 2;
      done |= 1ull << (dev & 63);
    }
    hipLaunchKernelGGL((lin160_kernel<128, 1>), dim3((unsigned)((g.M / 128) * ncb)), dim3(512), shm, s, p);
    return 0;
  }
  // 128-row tiles when they still fill the chip (and the row count allows), else 64-row tiles
  const bool big = g.M % 128 == 0 && (long long)(Mp / 128) * ncb >= 256;
  constexpr size_t shm128 = (size_t)L1_NS * (L1_W_STAGE + 128 * 128), shm64 = (size_t)L1_NS * (L1_W_STAGE + 64 * 128);
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!(g_l1_attr >> (dev & 63) & 1ull)) {
    if (hipFuncSetAttribute((const void*)lin160_kernel<128, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm128) != hipSuccess) return 2;
    if (hipFuncSetAttribute((const void*)lin160_kernel<64, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm64) != hipSuccess) return 2;
    g_l1_attr |= 1ull << (dev & 63);
  }
  if (big) hipLaunchKernelGGL((lin160_kernel<128, 0>), dim3((unsigned)((g.M / 128) * ncb)), dim3(512), shm128, s, p);
  else hipLaunchKernelGGL((lin160_kernel<64, 0>), dim3((unsigned)((g.M / 64) * ncb)), dim3(512), shm64, s, p);
  return 0;
}
