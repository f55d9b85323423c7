// One kernel per temporal-attention block of the C = 320 level (8 heads of d = 40; F = 16 frames, or F = 32 for BASELINE config 5: see the
// template parameter of tattn_fused_kernel; the description below is the F = 16 form):
//
//     t  <-  t + to_out( softmax( q k^T / sqrt(d) ) v ) ,   [q | k | v] = ( LayerNorm(t) + pe[frame] ) . [Wq | Wk | Wv]^T
//
// i.e. TemporalTransformerBlock's `norm -> VersatileAttention -> + hidden_states` (animatediff/models/motion_module.py:210-218) with
// VersatileAttention.forward (:270-329: regroup "(b f) d c -> (b d) f c", PositionalEncoding :241-243, CrossAttention arithmetic
// motion_module_new.py:201-287, to_out[0] with bias).  Until round 3 this was three launches -- LayerNorm-folded q|k|v projection
// (43 us at M = 32768), the strided 16 x 16 attention core (25 us, HBM-bound on the 63 MB q|k|v tensor) and the to_out GEMM + residual
// (23 us) -- with q|k|v and the attention output round-tripping through HBM.  Here a workgroup owns 8 pixels x all 16 frames (128 rows
// of the "(b f) (h w) c" activation, gathered with stride h*w*C) and nothing but t leaves the chip.
//
// Structure (256-thread workgroup = 4 waves, one per SIMD; wave w owns 2 pixels = 2 MFMA row tiles of 16 frames):
//   * prologue: the wave's 32 x 320 panel of t -> registers, LayerNorm statistics, then xn = (x - mean) rstd gamma + (beta + pe[frame])
//     rounded to bf16 IN the registers (the rounding point of the un-fused LayerNorm kernel), kept as MFMA fragments (80 VGPRs).
//   * per head h, four weight stages streamed through a ring of three LDS slots by linear LDS-DMA copies (the stream is stored in
//     HBM as the LDS image, XOR-swizzled [rows][64] sub-tiles as in gemm.hip / ffpanel.hip):
//        q_h, k_h  [48 n][320 k] (40 rows + 8 zero rows):  acc = W . xn^T  ("transposed" issue: lane holds 4 channels of its frame)
//        v_h       same shape, operands SWAPPED:           acc = xn . W^T  (lane holds 4 frames of its channel = V^T as A operand)
//        o_h       [320 n][64 k-slots]: Wo[:, 40 h .. 40 h + 39] in the k order in which the attention output sits in registers
//   * attention of (pixel, head) entirely in registers with v_mfma_f32_16x16x16_bf16 (k = 16 = one accumulator tile):
//        S^T = K Q^T (3 MFMAs: 48 padded channels), softmax over the 16 keys = 4 registers x 4 lane groups (two xor-shuffles),
//        O^T = V^T P^T (3 MFMAs); every accumulator -> operand hand-over is lane-local (cdna_hip_programming.md 3, "An accumulator
//        tile as the next MFMA's operand"): no LDS round trip.
//   * out += Wo_h . O_h^T accumulates over the heads in 160 VGPRs; epilogue t + bo + acc, written in place (rows are private).
// Algorithmic work per launch at M = 32768: 26.8 GFLOP GEMM + 0.17 GFLOP attention; HBM: t in + t out = 42 MB (+ 1.1 MB of weights,
// L2-resident per XCD).
#include "common.h"
#include <cstdlib>

namespace {

typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((ext_vector_type(4))) short s16x4;

__device__ __forceinline__ void glds16(const void* src, unsigned lds_wave_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(lds_wave_base) : "memory", "m0");
}
template <int N> __device__ __forceinline__ void wait_vmcnt() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

constexpr int TA_C = 320, TA_HEADS = 8, TA_D = 40;
constexpr int TA_ROWS = 128;                    // rows per workgroup = 8 pixels x 16 frames or 4 pixels x 32 frames
constexpr int TA_QSUB = 48 * 64;                // elements of one [48][64] sub-tile of a q / k / v stage
constexpr int TA_QKV_BYTES = 32 * 1024;         // stage stride of a q / k / v stage (30 KiB image + 2 KiB pad: 8 DMA per wave)
constexpr int TA_OSUB = 64 * 64;
constexpr int TA_O_BYTES = 40 * 1024;           // o stage: five [64][64] sub-tiles (10 DMA per wave)
constexpr int TA_HEAD_BYTES = 3 * TA_QKV_BYTES + TA_O_BYTES;
constexpr int TA_SLOT = TA_O_BYTES;             // ring slot
constexpr int TA_NS = 3;

struct NrTAttnParams {
  bf16* t;                 // [B2 * F * hw][C], updated in place
  int hw, nbatch;          // pixels per frame-image, CFG batch
  int norot;               // 1: every workgroup walks the heads from head 0 (results independent of the row position: NR_DETERMINISTIC_BATCH)
  const bf16* stream;      // 8 heads x (q | k | v | o) stages (tattn_stream_pack_kernel)
  const float* gamma;      // [C] LayerNorm weight
  const float* gb;         // [F][C] LayerNorm bias + positional encoding of the frame
  const float* bo;         // [C] to_out bias
  float ln_eps;
  float scale_log2e;       // d^-0.5 * log2(e)
  int dbg;                 // timing experiments only (NR_FUSED_DBG): 1 no DMA waits, 2 no stage barriers, 4 no DMA issue (results are wrong)
};

// out-tile accumulation with the accumulator PINNED in the AGPR half of the register file (round 5, as xattn.hip: "+a": vDst = SrcC = an AGPR quad).
// Left to hipcc the 160 accumulator registers of the out tile lived in VGPRs between the heads and the MFMA groups were bracketed by v_accvgpr
// copies (584 copies per head iteration for 272 MFMAs).  The asm MFMA is invisible to the compiler's hazard bookkeeping: its A operand comes from
// LDS (s_waitcnt placed for the asm input), its B operand (ob_prev*) is written by the attention's last phase, which never sits directly in front
// of one of these MFMAs (phase 6 runs behind the last group), and the accumulator is next touched one head later or by the epilogue behind an
// explicit s_nop.  tattn.o is compiled with -amdgpu-mfma-vgpr-form so the short-lived projection / attention accumulators stay in VGPRs.
#ifndef NR_ACC_AGPR
#define NR_ACC_AGPR 1      // 0: the compiler-allocated form again (A/B arm: make variant NAME=noagpr VFLAGS=-DNR_ACC_AGPR=0)
#endif
__device__ __forceinline__ void mfma_acc_agpr(f32x4& acc, const bf16x8& a, const bf16x8& b) {
#if NR_ACC_AGPR
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
#else
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
#endif
}

// own and partner value across 16-lane rows on the VALU (round 5): v_permlane16_swap / v_permlane32_swap with both operands = v give every lane
// {own, partner} in the two results (lane ^ 16 / lane ^ 32), where __shfl_xor is a ds_bpermute round trip through the LDS -- eight of them per head
// sat on the latency chain of the attention phases.  max / + are commutative: bit-identical to the shuffle form.
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

// F = 16 (round 3): a wave owns 2 pixels, row tile mt = pixel.  F = 32 (round 4, BASELINE config 5): a wave owns ONE pixel, row tile mt = frames
// 16 mt .. 16 mt + 15; the projections and the out GEMM do not care what a row tile means, only the attention core and the addressing do.
template <int F>
__global__ __launch_bounds__(256) void tattn_fused_kernel(NrTAttnParams p) {
  static_assert(F == 16 || F == 32, "one or two MFMA row tiles of frames per pixel");
  constexpr int C = TA_C, KS = C / 32, NT2 = C / 16;
  constexpr int PIX_WG = TA_ROWS / F, PIX_WAVE = 32 / F;       // pixels per workgroup / per wave (8 / 2 or 4 / 1)
  extern __shared__ __attribute__((aligned(16))) bf16 smem[];   // TA_NS slots of 40 KiB

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;

  // ---- weight stream.  The heads are independent up to the order of the fp32 accumulation of the out tile, so every workgroup walks them
  // from its own starting head: the 32 workgroups of an XCD then read 8 different regions of the stream instead of all hammering the same
  // 32 KiB (the same few L2 channels) in lockstep.  blockIdx % 8 labels the XCD (speed only; results depend on blockIdx alone). ----
  const int head0 = p.norot ? 0 : (int)((blockIdx.x >> 3) & (TA_HEADS - 1));
  const char* wsrc = reinterpret_cast<const char*>(p.stream) + (size_t)lane * 16;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(lptr_t)smem);
  // Stage order: q k v o per head.  (Running the o stage one head late, so that its 80 MFMAs cover the shuffle / exp latency chain of the next
  // head's attention, was built and spilled 129 VGPRs: the panel, the packed q|k|v tiles and the attention state do not fit 256 VGPRs.)
  // piece i of a stage: q / k / v stages have 32 pieces of 1 KiB (8 per wave), o stages 40 (10 per wave)
  const char* pf_src = wsrc;
  unsigned pf_dst = lds0;
  int pf_n = 0;                     // pieces per wave of the stage being prefetched: 8 or 10
  auto set_prefetch = [&](int hidx, int part, int slot) {      // hidx: position in this workgroup's head order
    const int head = (head0 + hidx) & (TA_HEADS - 1);
    pf_n = part < 3 ? 8 : 10;
    pf_src = wsrc + (size_t)head * TA_HEAD_BYTES + (size_t)part * TA_QKV_BYTES + (size_t)(wave * pf_n) * 1024;
    pf_dst = lds0 + (unsigned)(slot * TA_SLOT) + (unsigned)(wave * pf_n * 1024);
  };
  auto prefetch_piece = [&](int i) { if (!(p.dbg & 4)) glds16(pf_src + i * 1024, pf_dst + (unsigned)(i * 1024)); };
  set_prefetch(0, 0, 0);
#pragma unroll
  for (int i = 0; i < 8; ++i) prefetch_piece(i);
  set_prefetch(0, 1, 1);
#pragma unroll
  for (int i = 0; i < 8; ++i) prefetch_piece(i);

  // ---- the row panel: row index in t = (b F + frame) hw + pixel; F = 16: tile mt = pixel pix0 + mt, lane row fr = frame;
  // F = 32: tile mt = frames 16 mt + fr of the wave's one pixel ----
  const int groups_per_img = p.hw / PIX_WG;
  const int b = blockIdx.x / groups_per_img;
  const int pix0 = (blockIdx.x - b * groups_per_img) * PIX_WG + wave * PIX_WAVE;
  bf16* trow[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
    trow[mt] = F == 16 ? p.t + ((size_t)(b * F + fr) * p.hw + pix0 + mt) * C : p.t + ((size_t)(b * F + 16 * mt + fr) * p.hw + pix0) * C;
  bf16x8 xb[2][KS];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) xb[mt][ks] = *(const bf16x8*)(trow[mt] + 32 * ks + 8 * fg);
  // LayerNorm (two-pass in registers: mean, then centred second moment) + positional encoding, rounded to bf16 in place
  {
    float mu[2], rstd[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      float s = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) s += (float)xb[mt][ks][e];
      s = xsum32(xsum16(s));
      mu[mt] = s * (1.0f / C);
      float q = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = (float)xb[mt][ks][e] - mu[mt]; q += d * d; }
      q = xsum32(xsum16(q));
      rstd[mt] = rsqrtf(q * (1.0f / C) + p.ln_eps);
    }
    const float* gbr = p.gb + (size_t)fr * C + 8 * fg;
    const float* gar = p.gamma + 8 * fg;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const f32x4 g0 = *(const f32x4*)(gar + 32 * ks), g1 = *(const f32x4*)(gar + 32 * ks + 4);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const float* gbm = gbr + (F == 32 ? (size_t)16 * mt * C : (size_t)0);       // positional encoding of frame 16 mt + fr
        const f32x4 b0 = *(const f32x4*)(gbm + 32 * ks), b1 = *(const f32x4*)(gbm + 32 * ks + 4);
        bf16x8 v = xb[mt][ks];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = (bf16)(((float)v[e] - mu[mt]) * rstd[mt] * g0[e] + b0[e]);
          v[4 + e] = (bf16)(((float)v[4 + e] - mu[mt]) * rstd[mt] * g1[e] + b1[e]);
        }
        xb[mt][ks] = v;
      }
    }
  }

  f32x4 oacc[NT2][2];
#pragma unroll
  for (int nt = 0; nt < NT2; ++nt)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) oacc[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

  int slot = 0;
  // stage start: this wave's pieces of the stage (landed) vs the FOLLOWING stage's pieces (next_dma of them may stay in flight)
  auto stage_wait = [&](int next_dma) {
    if (!(p.dbg & 1)) { if (next_dma == 10) wait_vmcnt<10>(); else wait_vmcnt<8>(); }
    if (!(p.dbg & 2)) __builtin_amdgcn_s_barrier();
  };
  auto next_slot = [&]() { slot = slot + 1 == TA_NS ? 0 : slot + 1; };
  auto slot_plus2 = [&]() { int x = slot + 2; return x >= TA_NS ? x - TA_NS : x; };

  // fragment of a q / k / v stage: 16 weight rows nt (0..2), k-step ks
  auto frag_qkv = [&](const bf16* sW, int nt, int ks) {
    const int row = nt * 16 + fr;
    return *(const bf16x8*)(sW + (ks >> 1) * TA_QSUB + row * 64 + ((((ks & 1) * 4 + fg) ^ (row & 7)) << 3));
  };
  // acc[nt][mt] over K = 320; SWAP: activations as the A operand (result transposed: lane = channel, registers = frames).
  // One LDS-DMA piece of the stage two ahead goes out behind each of the first pf_n k-steps.
  auto gemm_qkv = [&](const bf16* sW, s16x4 (&outp)[3][2], bool swap) {
    f32x4 acc[3][2];
#pragma unroll
    for (int nt = 0; nt < 3; ++nt)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) acc[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 w0[3], w1[3], w2[3];
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) { w0[nt] = frag_qkv(sW, nt, 0); w1[nt] = frag_qkv(sW, nt, 1); }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bf16x8 (&wc)[3] = (ks % 3 == 0) ? w0 : (ks % 3 == 1 ? w1 : w2);
      bf16x8 (&wn)[3] = (ks % 3 == 0) ? w2 : (ks % 3 == 1 ? w0 : w1);
      if (ks + 2 < KS) {
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) wn[nt] = frag_qkv(sW, nt, ks + 2);
      }
      if (ks < 8) prefetch_piece(ks);
      else if (pf_n == 10) prefetch_piece(ks);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int nt = 0; nt < 3; ++nt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
          acc[nt][mt] = swap ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(xb[mt][ks], wc[nt], acc[nt][mt], 0, 0, 0)
                             : __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[nt], xb[mt][ks], acc[nt][mt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    // the attention consumes these tiles as bf16 MFMA operands: keep them packed (12 VGPRs per tensor instead of 24)
#pragma unroll
    for (int nt = 0; nt < 3; ++nt)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) outp[nt][mt] = pack4(acc[nt][mt]);
  };
  auto frag_o = [&](const bf16* sW, int nt, int ks2) {
    const int row = (nt & 3) * 16 + fr;
    return *(const bf16x8*)(sW + (nt >> 2) * TA_OSUB + row * 64 + (((ks2 * 4 + fg) ^ (row & 7)) << 3));
  };

  // ---- attention of one head for the wave's two pixels, cut into 7 phases at its latency points (MFMA result -> cross-lane shuffle ->
  // exp -> shuffle -> rcp -> MFMA); each phase is pinned (opaque asm) so that the o stage of the previous head can put one group of 8
  // MFMAs between consecutive phases ----
  s16x4 qa[3][2], ka[3][2], va[3][2];
  constexpr int KT = F / 16;          // key tiles per query tile: 1 (F = 16: keys = the 16 frames of the tile's own pixel) or 2 (F = 32: both frame tiles)
  f32x4 at_s[KT][2], at_o[3][2];      // at_s[kt][mt]: scores of key tile kt for the queries of tile mt, then their exponentials
  float at_m[2], at_l[2];
  bf16x8 ob_prev0[2], ob_prev1[2];   // O^T of the head whose o stage runs next, as the two B fragments of that stage
  auto attn_phase = [&](int ph) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      if (ph == 0) {               // S^T[key 4 fg + r][query fr] += K[key][c] Q[query][c] over the 16 channels of tile nt
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          const int km = F == 16 ? mt : kt;                  // row tile that holds the keys
          f32x4 s4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int nt = 0; nt < 3; ++nt) s4 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ka[nt][km], qa[nt][mt], s4, 0, 0, 0);
          at_s[kt][mt] = s4;
        }
      } else if (ph == 1) {
        float mx = fmaxf(fmaxf(at_s[0][mt][0], at_s[0][mt][1]), fmaxf(at_s[0][mt][2], at_s[0][mt][3]));
        if constexpr (KT == 2) mx = fmaxf(mx, fmaxf(fmaxf(at_s[1][mt][0], at_s[1][mt][1]), fmaxf(at_s[1][mt][2], at_s[1][mt][3])));
        at_m[mt] = xmax16(mx);
      } else if (ph == 2) {
        at_m[mt] = xmax32(at_m[mt]);
      } else if (ph == 3) {
        float l = 0.f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) { at_s[kt][mt][r] = __builtin_amdgcn_exp2f((at_s[kt][mt][r] - at_m[mt]) * p.scale_log2e); l += at_s[kt][mt][r]; }
        at_l[mt] = xsum16(l);
      } else if (ph == 4) {
        at_l[mt] = xsum32(at_l[mt]);
      } else if (ph == 5) {        // O^T[channel 16 g + 4 fg + r][query fr] = V^T P^T
        if constexpr (KT == 1) {
          const s16x4 pb = pack4(at_s[0][mt]);
#pragma unroll
          for (int g = 0; g < 3; ++g)
            at_o[g][mt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(va[g][mt], pb, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        } else {
          // 32 keys = ONE 32-deep MFMA: k-slot 8 fg + j <-> key (j < 4 ? 4 fg + j : 16 + 4 fg + j - 4), i.e. the lane's own registers of key
          // tile 0 then key tile 1 -- for P (accumulator layout of S^T) and for V^T (the swapped projection leaves 4 frames of each tile per lane)
          const s16x4 p0 = pack4(at_s[0][mt]), p1 = pack4(at_s[1][mt]);
          typedef __attribute__((ext_vector_type(8))) short s16x8;
          const s16x8 pb8 = {p0[0], p0[1], p0[2], p0[3], p1[0], p1[1], p1[2], p1[3]};
#pragma unroll
          for (int g = 0; g < 3; ++g) {
            const s16x8 va8 = {va[g][0][0], va[g][0][1], va[g][0][2], va[g][0][3], va[g][1][0], va[g][1][1], va[g][1][2], va[g][1][3]};
            at_o[g][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, va8), __builtin_bit_cast(bf16x8, pb8),
                                                                  f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          }
        }
      } else {
        const float inv = __builtin_amdgcn_rcpf(at_l[mt]);
        bf16x8 b0, b1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          b0[r] = (bf16)(at_o[0][mt][r] * inv); b0[4 + r] = (bf16)(at_o[1][mt][r] * inv);   // k-slots 8 fg + j: channels {4 fg + j}, {16 + 4 fg + j}
          b1[r] = (bf16)(at_o[2][mt][r] * inv); b1[4 + r] = (bf16)0.0f;                    // k-slots 32 + 8 fg + j: channels {32 + 4 fg + j} (fg < 2)
        }
        ob_prev0[mt] = b0; ob_prev1[mt] = b1;
      }
    }
    // pin the phase
    if (ph == 0 || ph == 3) {
      asm volatile("" : "+v"(at_s[0][0]), "+v"(at_s[0][1]));
      if constexpr (KT == 2) asm volatile("" : "+v"(at_s[1][0]), "+v"(at_s[1][1]));
      if (ph == 3) asm volatile("" : "+v"(at_l[0]), "+v"(at_l[1]));
    }
    else if (ph == 1 || ph == 2) asm volatile("" : "+v"(at_m[0]), "+v"(at_m[1]));
    else if (ph == 4) asm volatile("" : "+v"(at_l[0]), "+v"(at_l[1]));
    else if (ph == 5) asm volatile("" : "+v"(at_o[0][0]), "+v"(at_o[1][0]), "+v"(at_o[2][0]), "+v"(at_o[0][1]), "+v"(at_o[1][1]), "+v"(at_o[2][1]));
    else asm volatile("" : "+v"(ob_prev0[0]), "+v"(ob_prev0[1]), "+v"(ob_prev1[0]), "+v"(ob_prev1[1]));
  };
  // o stage: out += Wo[:, head] . O^T (ob_prev*), 10 groups of 8 MFMAs; WITH_ATTN: attention phases 0..5 of the current head in front of groups 0..5, phase 6 behind the last
  auto gemm_o = [&](const bf16* sW, bool with_attn) {
    bf16x8 fa[4], fb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[i] = frag_o(sW, i, 0);
#pragma unroll
    for (int grp = 0; grp < 10; ++grp) {
      const int ks2 = grp / 5, q = grp - 5 * ks2;
      bf16x8 (&cur)[4] = (grp & 1) ? fb : fa;
      bf16x8 (&nxt)[4] = (grp & 1) ? fa : fb;
      if (grp + 1 < 10) {
        const int g2 = grp + 1, k2 = g2 / 5, q2 = g2 - 5 * k2;
#pragma unroll
        for (int i = 0; i < 4; ++i) nxt[i] = frag_o(sW, 4 * q2 + i, k2);
      }
      if (grp < 8) prefetch_piece(grp);
      __builtin_amdgcn_sched_barrier(0);
      if (with_attn && grp < 6) attn_phase(grp);      // phase 6 overwrites the B fragments this stage reads: after the last group
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int nt = 4 * q + i;
        mfma_acc_agpr(oacc[nt][0], cur[i], ks2 ? ob_prev1[0] : ob_prev0[0]);
        mfma_acc_agpr(oacc[nt][1], cur[i], ks2 ? ob_prev1[1] : ob_prev0[1]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (with_attn) attn_phase(6);
  };

  for (int it = 0; it < TA_HEADS; ++it) {
    const bool last = it + 1 == TA_HEADS;
    // ---- q stage; in flight behind it: k (8).  Prefetch: v of this head (8) ----
    stage_wait(8);
    set_prefetch(it, 2, slot_plus2());
    gemm_qkv(smem + slot * (TA_SLOT / 2), qa, false);
    next_slot();
    // ---- k stage; behind it: v (8).  Prefetch: o of this head (10) ----
    stage_wait(8);
    set_prefetch(it, 3, slot_plus2());
    gemm_qkv(smem + slot * (TA_SLOT / 2), ka, false);
    next_slot();
    // ---- v stage; behind it: o (10).  Prefetch: q of the next head (last head: a harmless re-fetch of the first head's q keeps the piece
    // issue unconditional) ----
    stage_wait(10);
    set_prefetch(last ? 0 : it + 1, 0, slot_plus2());
    gemm_qkv(smem + slot * (TA_SLOT / 2), va, true);
    next_slot();
    // ---- attention of this head (both pixels of the wave), in registers ----
#pragma unroll
    for (int ph = 0; ph < 7; ++ph) attn_phase(ph);
    // ---- o stage; behind it: q of the next head (8) or the dummy.  Prefetch: k of the next head ----
    stage_wait(8);
    set_prefetch(last ? 0 : it + 1, 1, slot_plus2());
    gemm_o(smem + slot * (TA_SLOT / 2), false);
    next_slot();
  }
  wait_vmcnt<0>();      // the tail's dummy pieces

  asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");     // the last asm MFMAs' results -> the accumulator reads below (>= 18 wait states, stated not assumed)
  // ---- epilogue: t <- t + bo + acc, in place.  In the accumulator layout a lane holds 4 channels (16 nt + 4 fg .. + 3) of row fr: 8-byte accesses in
  // 32-byte row segments, and the whole chip runs this phase at once (t in + t out = 42 MB: 23 k of the kernel's 100 k cycles, xattn.hip's timeline, same structure).
  // v_permlane16_swap between the column tiles (2 k, 2 k + 1) hands every lane 8 CONSECUTIVE channels (even rows: tile 2 k, channels 4 fg .. 4 fg + 7;
  // odd rows: tile 2 k + 1, channels 4 (fg - 1) ..): 16-byte residual loads and stores in 64-byte row segments, half the vector-memory instructions ----
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
    for (int k = 0; k < NT2 / 2; ++k) {
      f32x4 lo = oacc[2 * k][mt], hi = oacc[2 * k + 1][mt];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(lo[e]), __float_as_uint(hi[e]), false, false);
        lo[e] = __uint_as_float(sw[0]); hi[e] = __uint_as_float(sw[1]);
      }
      const int col0 = 16 * (2 * k + (fg & 1)) + 4 * (fg & 2);      // even rows: own tile's channels 4 fg ..; odd rows: the next tile's 4 (fg - 1) ..
      bf16* tp = trow[mt] + col0;
      const bf16x8 xv = *(const bf16x8*)tp;
      const f32x4 b0 = *(const f32x4*)(p.bo + col0), b1 = *(const f32x4*)(p.bo + col0 + 4);
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[e] = (bf16)(lo[e] + b0[e] + (float)xv[e]);
        o[4 + e] = (bf16)(hi[e] + b1[e] + (float)xv[4 + e]);
      }
      nr_store16(tp, o);
    }
  }
}

// Builds the weight stream from the four bf16 [C][C] matrices (rows = output features).  One thread per 16-byte chunk.
__global__ __launch_bounds__(256) void tattn_stream_pack_kernel(const bf16* __restrict__ wq, const bf16* __restrict__ wk, const bf16* __restrict__ wv,
                                                                const bf16* __restrict__ wo, bf16* __restrict__ stream) {
  constexpr int C = TA_C;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  constexpr int CH_HEAD = TA_HEAD_BYTES / 16;
  if (idx >= TA_HEADS * CH_HEAD) return;
  const int head = idx / CH_HEAD;
  int c = idx - head * CH_HEAD;
  bf16x8 v = bf16x8_zero();
  if (c < 3 * (TA_QKV_BYTES / 16)) {
    const int part = c / (TA_QKV_BYTES / 16);
    c -= part * (TA_QKV_BYTES / 16);
    if (c < 5 * 48 * 8) {                                      // [5 sub-tiles][48 rows][8 chunks]; the tail of the stage is padding
      const int sub = c / (48 * 8), row = (c / 8) % 48, phys = c & 7;
      const int lchunk = phys ^ (row & 7);
      if (row < TA_D) {
        const bf16* w = part == 0 ? wq : (part == 1 ? wk : wv);
        v = *(const bf16x8*)(w + (size_t)(head * TA_D + row) * C + 64 * sub + 8 * lchunk);
      }
    }
  } else {
    c -= 3 * (TA_QKV_BYTES / 16);
    const int sub = c >> 9, row = (c >> 3) & 63, phys = c & 7;   // [5][64 n rows][8 chunks of 8 k-slots]
    const int lchunk = phys ^ (row & 7);
    const int n = 64 * sub + row;
    const int ks2 = lchunk >> 2, fgq = lchunk & 3;
    const bf16* src = wo + (size_t)n * C + head * TA_D;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int ch = -1;
      if (ks2 == 0) ch = j < 4 ? 4 * fgq + j : 16 + 4 * fgq + (j - 4);
      else if (j < 4 && fgq < 2) ch = 32 + 4 * fgq + j;
      v[j] = ch >= 0 ? src[ch] : (bf16)0.0f;
    }
  }
  *(bf16x8*)(stream + (size_t)idx * 8) = v;
}

unsigned long long g_ta_attr = 0;

}  // namespace

extern "C" size_t nr_tattn_stream_bytes(void) { return (size_t)TA_HEADS * TA_HEAD_BYTES; }

extern "C" int nr_tattn_fused_eligible(int C, int heads, int frames, int hw, long long rows) {
  static const bool off = getenv("NR_TATTN_FUSED") && getenv("NR_TATTN_FUSED")[0] == '0';   // A/B switch
  return !off && C == TA_C && heads == TA_HEADS && (frames == 16 || frames == 32) && hw % (TA_ROWS / frames) == 0 && rows >= 4096;
}

extern "C" int nr_launch_tattn_stream_pack(const bf16* wq, const bf16* wk, const bf16* wv, const bf16* wo, bf16* stream, hipStream_t s) {
  const int total = TA_HEADS * (TA_HEAD_BYTES / 16);
  hipLaunchKernelGGL(tattn_stream_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, s, wq, wk, wv, wo, stream);
  return 0;
}

extern "C" int nr_launch_tattn_fused(bf16* t, int nbatch, int frames, int hw, const bf16* stream, const float* gamma, const float* gb, const float* bo,
                                     float ln_eps, int norot, hipStream_t s) {
  if (nbatch <= 0 || hw <= 0 || (frames != 16 && frames != 32) || hw % (TA_ROWS / frames) != 0) return 1;
  NrTAttnParams p;
  p.t = t; p.hw = hw; p.nbatch = nbatch; p.stream = stream; p.gamma = gamma; p.gb = gb; p.bo = bo; p.ln_eps = ln_eps; p.norot = norot;
  p.scale_log2e = 1.4426950408889634f / sqrtf((float)TA_D);
  static const int dbg = getenv("NR_FUSED_DBG") ? atoi(getenv("NR_FUSED_DBG")) : 0;
  p.dbg = dbg;
  constexpr size_t shm = (size_t)TA_NS * TA_SLOT;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!(g_ta_attr >> (dev & 63) & 1ull)) {
    if (hipFuncSetAttribute((const void*)tattn_fused_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess) return 2;
    if (hipFuncSetAttribute((const void*)tattn_fused_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess) return 2;
    g_ta_attr |= 1ull << (dev & 63);
  }
  const unsigned grid = (unsigned)(nbatch * (hw / (TA_ROWS / frames)));
  if (frames == 16) hipLaunchKernelGGL(tattn_fused_kernel<16>, dim3(grid), dim3(256), shm, s, p);
  else hipLaunchKernelGGL(tattn_fused_kernel<32>, dim3(grid), dim3(256), shm, s, p);
  return 0;
}
