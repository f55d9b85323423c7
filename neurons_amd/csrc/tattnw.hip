// The q|k|v projection and the F x F attention of one temporal-attention block ABOVE the C = 320 level (C = 640: d = 80, C = 1280: d = 160;
// 8 heads, F = 16 frames) in one launch, one workgroup per (pixel group, head):
//
//     a[:, head]  =  softmax( q k^T / sqrt(d) ) v ,   [q | k | v] = ( LayerNorm(t) + pe[frame] ) . [Wq | Wk | Wv][head rows]^T
//
// i.e. TemporalTransformerBlock's `norm -> VersatileAttention` up to (not including) to_out (animatediff/models/motion_module.py:210-218;
// VersatileAttention.forward :270-329 with the "(b f) d c -> (b d) f c" regroup and PositionalEncoding :241-243; attention arithmetic
// motion_module_new.py:201-287).  Until round 5 these levels ran three launches per block: the LayerNorm-folded q|k|v GEMM (42 / 36 us at
// M = 8192 / 2048), the strided 16 x 16 attention core (14 us: 8-22 TFLOP/s, pure latency) and the to_out GEMM; q|k|v (M x 3C bf16) made a round
// trip through HBM.  Here q|k|v never leave the registers; to_out (+ bias + residual) stays a GEMM of its own on the attention output a
// (summing the heads inside this launch would need either all 8 heads in one workgroup -- 3.3 / 13 MB of weights streamed per 64 rows -- or a
// cross-workgroup reduction of fp32 slabs; tattn.hip can afford the former at C = 320 only).
//
// Structure (256 threads = 4 waves, one per SIMD; wave w owns MT row tiles = MT pixels x 16 frames: MT = 2 at d = 80, 1 at d = 160):
//   * the head's 3 d weight rows are W' = gamma . W (LayerNorm folded as in gemm.hip LNF: q = rstd (x W'^T - mean c) + b' + pe[f] W^T), stored
//     FRAGMENT-MAJOR ([stage = k-step][tile][64 lanes][8]: a wave's MFMA operand is one linear KiB of LDS, conflict-free without a swizzle) and
//     streamed k-step by k-step (32 channels) through an LDS ring by linear LDS-DMA; the raw rows of t come through the same ring (one piece = the
//     16 frames of a pixel x 64 bytes = one MFMA fragment, chunk-permuted per row so the fragment read is conflict-free; each wave fetches only
//     its own rows), so every vector-memory operation of the loop is a DMA piece counted by hand (s_waitcnt vmcnt) and the compiler only sees LDS
//     reads.  Ring: 3 slots of 24 KiB at d = 80 (two workgroups per CU = two waves per SIMD), 4 slots of 36 KiB at d = 160: two / three k-steps
//     of prefetch distance (the first build staged 64 channels per slot, 3 / 2 slots deep, one workgroup per CU, and waited on its DMA every
//     stage: 37 / 42 us per launch, profiles/r06_tattn_head_ab.txt);
//   * accumulators: q and k tiles as W' . x^T (lane = 4 channels of its frame), v tiles with the operands swapped (lane = 4 frames of its
//     channel = V^T as the A operand of the P.V product): 30 tiles of 16 x 16 per wave in both shapes;  row sums / sums of squares of x from the
//     fragments that pass anyway (v_dot2c_f32_bf16), reduced across the four lanes of a row;
//   * epilogue in registers: LayerNorm fold, positional-encoding row vector, bf16 pack; S^T = K Q^T, softmax over the 16 keys (v_permlane swaps
//     across the 16-lane rows), O^T = V^T P^T on v_mfma_f32_16x16x16_bf16 exactly as tattn.hip; 8-byte stores of a[row][head d + 16 g + 4 fg ..].
// Workgroup -> (pixel group, head): at C = 640 the 8 heads of a pixel group run on ONE XCD (the 164-KB row panel is fetched into one L2, every XCD
// streams all 2.4 MB of weights); at C = 1280 head h runs on XCD h (each L2 holds 1.2 MB of weights, the 5-MB activation is fetched by all).
// Algorithmic work per launch at M = 8192 / C = 640 (= M = 2048 / C = 1280): 20.1 GFLOP GEMM + 0.17 GFLOP attention.
#include "common.h"
#include <cstdlib>

namespace {

typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((ext_vector_type(4))) short s16x4;

__device__ __forceinline__ void glds16(const void* src, unsigned lds_wave_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(lds_wave_base) : "memory", "m0");
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ float xmax16(float v) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
}
__device__ __forceinline__ float xmax32(float v) {
  auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
}
__device__ __forceinline__ float xsum16(float v) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(a[0]) + __uint_as_float(a[1]);
}
__device__ __forceinline__ float xsum32(float v) {
  auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(a[0]) + __uint_as_float(a[1]);
}
__device__ __forceinline__ s16x4 pack4(const f32x4& v) {
  bf16x4 b;
#pragma unroll
  for (int e = 0; e < 4; ++e) b[e] = (bf16)v[e];
  return __builtin_bit_cast(s16x4, b);
}

constexpr int TW_HEADS = 8, TW_F = 16;

#ifdef NR_STAMP
// diagnostic build only (make stamp, tools/tattnw_timeline.py): shader-clock stamps of wave 0 of the first 512 workgroups.  Slots: 0 entry, 1 prologue
// issued, 2 + 3 s / 3 + 3 s / 4 + 3 s = stage s after its DMA wait / after its barrier / after its MFMAs were issued, 125 loop end, 126 kernel end,
// 127 the XCC_ID register (which XCD ran the workgroup)
__device__ unsigned long long tattnw_stamp_buf[512][128];
#define TW_STAMP(slot) do { if (threadIdx.x == 0 && blockIdx.x < 512 && (slot) < 128) tattnw_stamp_buf[blockIdx.x][(slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TW_STAMP(slot) do { } while (0)
#endif

template <int D> struct TW {
  static constexpr int C = TW_HEADS * D;
  static constexpr int DT = D / 16;                           // 16-channel tiles of one of q / k / v: 5 / 10
  static constexpr int MT = D == 80 ? 2 : 1;                  // row tiles (pixels) per wave
  static constexpr int NT = 3 * DT;                           // weight fragments per k-step: 15 / 30
  static constexpr int GS = D == 80 ? 5 : 6;                  // fragments per MFMA group (one LDS read batch)
  static constexpr int NG = NT / GS;                          // groups per k-step: 3 / 5
  static constexpr int ROWS_W = 16 * MT;                      // rows of t per wave: 32 / 16
  static constexpr int PIX_WG = 4 * MT;                       // pixels per workgroup: 8 / 4
  // one stage = ONE k-step (32 channels): NT weight fragments of 1 KiB, padded to whole pieces per wave, + 64 bytes of each of the workgroup's rows
  static constexpr int W_STAGE = D == 80 ? 16 * 1024 : 32 * 1024;
  static constexpr int W_PIECES = W_STAGE / 4096;             // 1-KiB DMA pieces per wave: 4 / 8
  static constexpr int A_STAGE = 4 * ROWS_W * 64;             // 8 / 4 KiB
  static constexpr int A_PIECES = MT;                         // per wave (one piece = 16 rows x 64 B = one MFMA fragment): 2 / 1
  static constexpr int STAGE = W_STAGE + A_STAGE;             // 24 / 36 KiB
  static constexpr int NS = D == 80 ? 3 : 4;                  // ring slots: 72 KiB (two workgroups per CU) / 144 KiB
  static constexpr int S = C / 32;                            // stages: 20 / 40
  static constexpr int PPW = W_PIECES + A_PIECES;             // DMA pieces per wave and stage: 6 / 9
  static constexpr int PPS = 2;                               // pieces issued behind each MFMA group (NG groups per stage: 6 / 10 slots)
  static constexpr int WG_PER_CU = D == 80 ? 2 : 1;
  // d = 160 has exactly one workgroup per CU at the headline shape (256 = 32 pixel groups x 8 heads), i.e. one wave per SIMD, and that wave spent
  // 1 170 cycles per stage issuing 9 DMA pieces + 31 fragment reads around 480 cycles of MFMA (profiles/r06_tattnw_timeline_v3.txt).  So the
  // workgroup gets four PRODUCER waves (4-7, one per SIMD beside a consumer): they issue every DMA piece and nothing else, the consumer waves
  // (0-3) only read fragments and issue MFMAs.  At d = 80 two workgroups share a CU and overlap each other instead.
  static constexpr bool SPLIT = D == 160;
  static constexpr int THREADS = SPLIT ? 512 : 256;
  static constexpr int TBL_HEAD = D == 80 ? 16 * 1024 : 32 * 1024;   // epilogue table of one head: 17 x 3 d floats (16 320 / 32 640 B) padded to whole pieces
  static constexpr int TBL_PIECES = TBL_HEAD / 4096;          // per wave: 4 / 8
  static_assert(17 * 3 * D * 4 <= TBL_HEAD && TBL_HEAD <= STAGE, "the epilogue table lands in one free ring slot");
  static_assert(NT % GS == 0 && NG * PPS >= PPW && NG * PPS >= (D == 80 ? 4 : 8) && NS >= 3 && NS <= 4 && NT * 1024 <= W_STAGE, "piece schedule / wait counts");
};

struct NrTAttnWParams {
  const bf16* t;           // [B2 * F * hw][C] residual stream (raw: LayerNorm is folded)
  bf16* out;               // [B2 * F * hw][C] attention output a (before to_out)
  int hw, nbatch;          // pixels per frame-image, CFG batch
  int xcd_mode;            // 0: the 8 heads of a pixel group share an XCD (needs pixel groups % 8 == 0); 1: head h on XCD h
  const bf16* stream;      // [8 heads][S stages][W_STAGE] fragment-major folded weights (tattnw_stream_pack_kernel)
  const float* table;      // [8 heads][TBL_HEAD bytes]: per head [17][3 d] fp32 = row 0: c[n] = sum_k W'[n][k]; row 1 + f: b'[n] + pe[f] . W[n]^T
                           // (n = the head's q | k | v rows; tattnw_table_pack_kernel), padded to whole DMA pieces
  float ln_eps;
  float scale_log2e;       // d^-0.5 * log2(e)
};

template <int D>
__global__ __launch_bounds__(TW<D>::THREADS, TW<D>::WG_PER_CU) void tattn_head_kernel(NrTAttnWParams p) {
  using T = TW<D>;
  constexpr int C = T::C, DT = T::DT, MT = T::MT, NT = T::NT, GS = T::GS, NG = T::NG;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // T::NS slots of T::STAGE bytes: [weights | rows]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = T::SPLIT && wave_id >= 4;       // DMA-only wave (d = 160); it serves the rows / pieces of consumer wave_id - 4
  const int wave = wave_id & 3;
  const int fr = lane & 15, fg = lane >> 4;

  TW_STAMP(0);
  int pg, head;
  if (p.xcd_mode == 0) { const int j = blockIdx.x >> 3; head = j & 7; pg = (j >> 3) * 8 + (int)(blockIdx.x & 7); }
  else { head = (int)(blockIdx.x & 7); pg = blockIdx.x >> 3; }
  const int groups_per_img = p.hw / T::PIX_WG;
  const int b = pg / groups_per_img;
  const int pix0 = (pg - b * groups_per_img) * T::PIX_WG + wave * MT;      // this wave's first pixel

  // ---- DMA sources.  Weight piece i of stage s: 1 KiB at head stream + s W_STAGE + (wave W_PIECES + i) KiB.  Row piece a = row tile a of the wave
  // (16 frames of one pixel) x 64 bytes: lane (r = lane >> 2, phys = lane & 3) fetches the 16-byte chunk phys ^ f(r) of row r, f(r) = (-(r >> 2)) & 3:
  // with this permutation the ds_read_b128 of the MFMA fragment (lane (fr, fg) reads chunk fg of row fr) is bank-conflict-free in every
  // 16-lane group of the instruction (the four rows r, r + 4, r + 8, r + 12 that share a bank quarter get four different slots) ----
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(lptr_t)smem);
  const char* wsrc = reinterpret_cast<const char*>(p.stream) + (size_t)head * ((size_t)T::S * T::W_STAGE) + (size_t)(wave * T::W_PIECES) * 1024 + (size_t)lane * 16;
  const bf16* arow[T::A_PIECES];
#pragma unroll
  for (int a = 0; a < T::A_PIECES; ++a) {
    const int r = lane >> 2;
    arow[a] = p.t + ((size_t)(b * TW_F + r) * p.hw + pix0 + a) * C + (((lane & 3) ^ ((-(r >> 2)) & 3)) << 3);
  }
  // Every workgroup walks the k-steps from its own starting point (rot = its pixel group's position in the image, so a row's arithmetic does not
  // depend on the batch it runs in): the 32-64 workgroups of an XCD that stream the SAME head would otherwise request the same KiB of the stream
  // from the same L2 channels in lock-step (tattn.hip rotates its heads for the same reason; first builds of this kernel waited 860 cycles per
  // 36-KiB stage = 18 B/clk per CU, profiles/r06_tattnw_timeline.txt)
  const int rot = (pg - b * groups_per_img) % T::S;
  auto issue_piece = [&](int s, int slot, int i) {
    const unsigned dst = lds0 + (unsigned)(slot * T::STAGE);
    int ks = s + rot; if (ks >= T::S) ks -= T::S;
    if (i < T::W_PIECES) glds16(wsrc + (size_t)ks * T::W_STAGE + (size_t)i * 1024, dst + (unsigned)((wave * T::W_PIECES + i) * 1024));
    else glds16(arow[i - T::W_PIECES] + 32 * ks, dst + (unsigned)(T::W_STAGE + wave * (T::ROWS_W * 64) + (i - T::W_PIECES) * 1024));
  };
  const char* tsrc = reinterpret_cast<const char*>(p.table) + (size_t)head * T::TBL_HEAD + (size_t)(wave * T::TBL_PIECES) * 1024 + (size_t)lane * 16;
  // prologue: stages 0 .. NS - 2
  if (!T::SPLIT || producer) {
#pragma unroll
    for (int s = 0; s < T::NS - 1; ++s)
#pragma unroll
      for (int i = 0; i < T::PPW; ++i) issue_piece(s, s, i);
  }
  if constexpr (T::SPLIT) {
    if (producer) {
      // ---- producer wave: per stage wait for its pieces of stage s, meet the consumers at the barrier, refill the slot they just left ----
      int pslot = T::NS - 1;
      for (int s = 0; s < T::S; ++s) {
        const int rem = T::S - 1 - s;
        if (rem >= T::NS - 2) wait_vmcnt<(T::NS - 2) * T::PPW>();
        else if (rem == 1) wait_vmcnt<T::PPW>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        const int s_next = s + T::NS - 1;
        if (s_next < T::S) {
#pragma unroll
          for (int i = 0; i < T::PPW; ++i) issue_piece(s_next, pslot, i);
        } else if (s == T::S - 1) {           // the head's epilogue table into the slot stage S - 2 just left
#pragma unroll
          for (int i = 0; i < T::TBL_PIECES; ++i) glds16(tsrc + (size_t)i * 1024, lds0 + (unsigned)(pslot * T::STAGE + (wave * T::TBL_PIECES + i) * 1024));
        }
        pslot = pslot + 1 == T::NS ? 0 : pslot + 1;
      }
      wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();           // the table is in LDS: the consumers' epilogue may read it
      return;
    }
  }

  TW_STAMP(1);
  f32x4 acc[NT][MT];
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[n][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float s1[MT], s2[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) { s1[mt] = 0.f; s2[mt] = 0.f; }
  const bf16x2 one2 = {(bf16)1.0f, (bf16)1.0f};

  // LDS read offsets of this lane: weight fragment n at n KiB + 16 lane; row fragment mt at tile mt, row fr, slot fg ^ f(fr)
  const unsigned wl = (unsigned)lane * 16;
  const unsigned al = (unsigned)(T::W_STAGE + wave * (T::ROWS_W * 64) + fr * 64 + ((fg ^ ((-(fr >> 2)) & 3)) << 4));

  int slot = 0;
  for (int s = 0; s < T::S; ++s) {
    // this wave's pieces of stage s have landed when at most the pieces of the stages issued after it are outstanding
    // (issued so far: stages <= min(s + NS - 2, S - 1); allowed in flight: min(NS - 2, S - 1 - s) stages of PPW pieces)
    if constexpr (!T::SPLIT) {
      const int rem = T::S - 1 - s;
      if (rem >= T::NS - 2) wait_vmcnt<(T::NS - 2) * T::PPW>();
      else if (rem == 1) wait_vmcnt<T::PPW>();
      else wait_vmcnt<0>();
    }
    TW_STAMP(2 + 3 * s);
    __builtin_amdgcn_s_barrier();             // every wave's pieces landed; every wave has left stage s - 1 (its slot may be refilled)
    TW_STAMP(3 + 3 * s);
    const int s_next = s + T::NS - 1;
    const bool pf = !T::SPLIT && s_next < T::S;
    int pslot = slot + T::NS - 1; if (pslot >= T::NS) pslot -= T::NS;
    const unsigned char* base = smem + slot * T::STAGE;
    bf16x8 xa[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) xa[mt] = *(const bf16x8*)(base + al + mt * 1024);
    bf16x8 wa[GS], wb[GS];
#pragma unroll
    for (int i = 0; i < GS; ++i) wa[i] = *(const bf16x8*)(base + (unsigned)(i * 1024) + wl);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bf16x2 pr = {xa[mt][2 * e], xa[mt][2 * e + 1]};
        s1[mt] = __builtin_amdgcn_fdot2_f32_bf16(pr, one2, s1[mt], false);
        s2[mt] = __builtin_amdgcn_fdot2_f32_bf16(pr, pr, s2[mt], false);
      }
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      bf16x8 (&cur)[GS] = (g & 1) ? wb : wa;
      bf16x8 (&nxt)[GS] = (g & 1) ? wa : wb;
      if (g + 1 < NG) {
#pragma unroll
        for (int i = 0; i < GS; ++i) nxt[i] = *(const bf16x8*)(base + (unsigned)(((g + 1) * GS + i) * 1024) + wl);
      }
      if (pf) {
#pragma unroll
        for (int q = 0; q < T::PPS; ++q) {
          const int i = g * T::PPS + q;
          if (i < T::PPW) issue_piece(s_next, pslot, i);
        }
      } else if (!T::SPLIT && s == T::S - 1) {             // last stage: the head's epilogue table into the slot stage S - 2 just left
#pragma unroll
        for (int q = 0; q < T::PPS; ++q) {
          const int i = g * T::PPS + q;
          if (i < T::TBL_PIECES) glds16(tsrc + (size_t)i * 1024, lds0 + (unsigned)(pslot * T::STAGE + (wave * T::TBL_PIECES + i) * 1024));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < GS; ++i) {
        const int n = g * GS + i;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          acc[n][mt] = n < 2 * DT ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur[i], xa[mt], acc[n][mt], 0, 0, 0)
                                  : __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[mt], cur[i], acc[n][mt], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    TW_STAMP(4 + 3 * s);
    slot = slot + 1 == T::NS ? 0 : slot + 1;
  }
  TW_STAMP(125);
  // the epilogue table sits in the slot "before" the last stage's: slot now = (last + 1) % NS, table slot = (last + NS - 1) % NS = (slot + NS - 2) % NS
  int tslot = slot + T::NS - 2; if (tslot >= T::NS) tslot -= T::NS;
  const float* tb = reinterpret_cast<const float*>(smem + tslot * T::STAGE);
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();

  // ---- LayerNorm statistics of the wave's rows: lane (fr, fg) holds a quarter of row fr's sums ----
  float mu[MT], rstd[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const float a = xsum32(xsum16(s1[mt])) * (1.0f / C);
    const float q = xsum32(xsum16(s2[mt])) * (1.0f / C);
    mu[mt] = a;
    rstd[mt] = rsqrtf(fmaxf(q - a * a, 0.f) + p.ln_eps);
  }

  // ---- fold epilogue + attention per row tile (pixel); the three tensors as packed bf16 MFMA operands ----
  // table columns of this lane: q / k tiles nt -> part D + 16 nt + 4 fg .. + 3; v tile nt -> 2 D + 16 nt + fr (its 4 frames 4 fg + r)
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    s16x4 qa[DT], ka[DT], va[DT];
    const float* te = tb + (1 + fr) * (3 * D);
#pragma unroll
    for (int part = 0; part < 2; ++part) {
#pragma unroll
      for (int nt = 0; nt < DT; ++nt) {
        const int col = part * D + 16 * nt + 4 * fg;
        const f32x4 c4 = *(const f32x4*)(tb + col), e4 = *(const f32x4*)(te + col);
        f32x4 v = acc[part * DT + nt][mt];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (v[e] - mu[mt] * c4[e]) * rstd[mt] + e4[e];
        if (part == 0) qa[nt] = pack4(v); else ka[nt] = pack4(v);
      }
    }
    float muf[4], rsf[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { muf[r] = __shfl(mu[mt], 4 * fg + r, 64); rsf[r] = __shfl(rstd[mt], 4 * fg + r, 64); }
#pragma unroll
    for (int nt = 0; nt < DT; ++nt) {
      const int col = 2 * D + 16 * nt + fr;
      const float cs = tb[col];
      f32x4 v = acc[2 * DT + nt][mt];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = (v[r] - muf[r] * cs) * rsf[r] + tb[(1 + 4 * fg + r) * (3 * D) + col];
      va[nt] = pack4(v);
    }
    // S^T[key 4 fg + r][query fr] = sum_c K[key][c] Q[query][c]
    f32x4 s4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nt = 0; nt < DT; ++nt) s4 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ka[nt], qa[nt], s4, 0, 0, 0);
    float mx = fmaxf(fmaxf(s4[0], s4[1]), fmaxf(s4[2], s4[3]));
    mx = xmax32(xmax16(mx));
    float l = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) { s4[r] = __builtin_amdgcn_exp2f((s4[r] - mx) * p.scale_log2e); l += s4[r]; }
    l = xsum32(xsum16(l));
    const float inv = __builtin_amdgcn_rcpf(l);
    const s16x4 pb = pack4(s4);
    // O^T[channel 16 g + 4 fg + r][query fr] = V^T P^T ; a[row of (frame fr, pixel)][head D + 16 g + 4 fg + r]
    bf16* orow = p.out + ((size_t)(b * TW_F + fr) * p.hw + pix0 + mt) * C + head * D + 4 * fg;
#pragma unroll
    for (int g = 0; g < DT; ++g) {
      const f32x4 o4 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(va[g], pb, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      bf16x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (bf16)(o4[r] * inv);
      nr_store8(orow + 16 * g, o);
    }
  }
  TW_STAMP(126);
#ifdef NR_STAMP
  if (threadIdx.x == 0 && blockIdx.x < 512) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    tattnw_stamp_buf[blockIdx.x][127] = xcc & 0xf;
  }
#endif
}

// Builds the fragment-major weight stream from the LayerNorm-folded [3 C][C] bf16 matrix (rows q | k | v).  One thread per 16-byte chunk of the stream:
// chunk -> (head, stage = k-step, tile n, lane): W'[row(n, lane & 15)][32 stage + 8 (lane >> 4) .. + 7]
template <int D>
__global__ __launch_bounds__(256) void tattnw_stream_pack_kernel(const bf16* __restrict__ w, bf16* __restrict__ stream) {
  using T = TW<D>;
  constexpr int C = T::C, CH_STAGE = T::W_STAGE / 16;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)TW_HEADS * T::S * CH_STAGE) return;
  const int head = (int)(idx / ((long long)T::S * CH_STAGE));
  int c = (int)(idx - (long long)head * T::S * CH_STAGE);
  const int s = c / CH_STAGE;
  c -= s * CH_STAGE;
  bf16x8 v = bf16x8_zero();
  if (c < T::NT * 64) {                                        // the tail of a stage is padding
    const int n = c / 64, lane = c & 63;
    const int part = n / T::DT, nt = n - part * T::DT;
    const int row = part * C + head * D + 16 * nt + (lane & 15);
    v = *(const bf16x8*)(w + (size_t)row * C + 32 * s + 8 * (lane >> 4));
  }
  *(bf16x8*)(stream + (size_t)idx * 8) = v;
}

// epilogue table: per head [17][3 d] fp32 (row 0: c, row 1 + f: b' + pe[f] W^T) from the engine's LayerNorm-fold vectors (rows q | k | v of [3 C])
template <int D>
__global__ __launch_bounds__(256) void tattnw_table_pack_kernel(const float* __restrict__ lnc, const float* __restrict__ bias, const float* __restrict__ rowvec,
                                                                float* __restrict__ table) {
  using T = TW<D>;
  constexpr int C = T::C, PER_HEAD = T::TBL_HEAD / 4;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= TW_HEADS * PER_HEAD) return;
  const int head = idx / PER_HEAD, e = idx - head * PER_HEAD;
  float v = 0.f;
  if (e < 17 * 3 * D) {
    const int row = e / (3 * D), j = e - row * (3 * D);
    const int n = (j / D) * C + head * D + (j % D);
    v = row == 0 ? lnc[n] : bias[n] + rowvec[(size_t)(row - 1) * (3 * C) + n];
  }
  table[idx] = v;
}

unsigned long long g_tw_attr = 0;

}  // namespace

#ifdef NR_STAMP
extern "C" int nr_tattnw_stamp_read(void* dst, size_t bytes, int clear) {
  const size_t n = bytes < sizeof(tattnw_stamp_buf) ? bytes : sizeof(tattnw_stamp_buf);
  int rc = 0;
  if (dst) rc = (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(tattnw_stamp_buf), n, 0, hipMemcpyDeviceToHost);
  if (clear) { void* d = nullptr; (void)hipGetSymbolAddress(&d, HIP_SYMBOL(tattnw_stamp_buf)); (void)hipMemset(d, 0, sizeof(tattnw_stamp_buf)); }
  return rc;
}
#endif

extern "C" size_t nr_tattnw_stream_bytes(int C) {
  return C == 640 ? (size_t)TW_HEADS * TW<80>::S * TW<80>::W_STAGE : (C == 1280 ? (size_t)TW_HEADS * TW<160>::S * TW<160>::W_STAGE : 0);
}

// rows: the launch's row count (deterministic-batch mode: one clip's).  At C = 1280 a launch needs >= 2048 rows: with 512 (the 4 x 4 level at one
// clip) only 64 workgroups exist, each streaming a head's 1.2 MB alone: 34 us against 27 us for the q|k|v GEMM + attention core (profiles/r06_tattn_head_ab.txt)
extern "C" int nr_tattnw_eligible(int C, int heads, int frames, int hw, long long rows) {
  static const bool off = getenv("NR_TATTN_HEAD") && getenv("NR_TATTN_HEAD")[0] == '0';   // A/B switch
  if (off || heads != TW_HEADS || frames != TW_F) return 0;
  if (C == 640) return hw % TW<80>::PIX_WG == 0;
  if (C == 1280) return hw % TW<160>::PIX_WG == 0 && rows >= 2048;
  return 0;
}

extern "C" int nr_launch_tattnw_stream_pack(const bf16* w_folded, int C, bf16* stream, hipStream_t s) {
  const long long total = (long long)(nr_tattnw_stream_bytes(C) / 16);
  if (!total) return 1;
  const dim3 grid((unsigned)((total + 255) / 256));
  if (C == 640) hipLaunchKernelGGL(tattnw_stream_pack_kernel<80>, grid, dim3(256), 0, s, w_folded, stream);
  else hipLaunchKernelGGL(tattnw_stream_pack_kernel<160>, grid, dim3(256), 0, s, w_folded, stream);
  return 0;
}

extern "C" size_t nr_tattnw_table_bytes(int C) { return C == 640 ? (size_t)TW_HEADS * TW<80>::TBL_HEAD : (C == 1280 ? (size_t)TW_HEADS * TW<160>::TBL_HEAD : 0); }

extern "C" int nr_launch_tattnw_table_pack(const float* lnc, const float* bias, const float* rowvec, int C, float* table, hipStream_t s) {
  const int total = (int)(nr_tattnw_table_bytes(C) / 4);
  if (!total) return 1;
  const dim3 grid((unsigned)((total + 255) / 256));
  if (C == 640) hipLaunchKernelGGL(tattnw_table_pack_kernel<80>, grid, dim3(256), 0, s, lnc, bias, rowvec, table);
  else hipLaunchKernelGGL(tattnw_table_pack_kernel<160>, grid, dim3(256), 0, s, lnc, bias, rowvec, table);
  return 0;
}

extern "C" int nr_launch_tattnw(const bf16* t, bf16* out, int nbatch, int hw, int C, const bf16* stream, const float* table, float ln_eps, hipStream_t s) {
  if (nbatch <= 0 || hw <= 0 || !nr_tattnw_stream_bytes(C)) return 1;
  const int pix_wg = C == 640 ? TW<80>::PIX_WG : TW<160>::PIX_WG;
  if (hw % pix_wg != 0) return 1;
  NrTAttnWParams p;
  p.t = t; p.out = out; p.hw = hw; p.nbatch = nbatch; p.stream = stream; p.table = table; p.ln_eps = ln_eps;
  const int d = C / TW_HEADS;
  p.scale_log2e = 1.4426950408889634f / sqrtf((float)d);
  const int npg = nbatch * (hw / pix_wg);
  p.xcd_mode = (C == 640 && npg % 8 == 0) ? 0 : 1;
  const size_t shm = C == 640 ? (size_t)TW<80>::NS * TW<80>::STAGE : (size_t)TW<160>::NS * TW<160>::STAGE;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!(g_tw_attr >> (dev & 63) & 1ull)) {
    if (hipFuncSetAttribute((const void*)tattn_head_kernel<80>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)TW<80>::NS * TW<80>::STAGE)) != hipSuccess) return 2;
    if (hipFuncSetAttribute((const void*)tattn_head_kernel<160>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)TW<160>::NS * TW<160>::STAGE)) != hipSuccess) return 2;
    g_tw_attr |= 1ull << (dev & 63);
  }
  const unsigned grid = (unsigned)(npg * TW_HEADS);
  if (C == 640) hipLaunchKernelGGL(tattn_head_kernel<80>, dim3(grid), dim3(TW<80>::THREADS), shm, s, p);
  else hipLaunchKernelGGL(tattn_head_kernel<160>, dim3(grid), dim3(TW<160>::THREADS), shm, s, p);
  return 0;
}
