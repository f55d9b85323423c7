// Fused FeedForward(GEGLU) + proj_out block of the 32x32 level (C = 320):
//
//     out = x + bc + [t | g] . Wc^T ,   g = GEGLU( LayerNorm(t) . W1^T + b1 )
//
// i.e. BasicTransformerBlock's / TemporalTransformerBlock's last FeedForward with its residual add, followed by the transformer's
// proj_out with ITS residual add (reference: animatediff/models/attention.py:129-140,297-299; motion_module.py:150-158,219-221;
// FeedForward / GEGLU motion_module_new.py:441-518), in ONE launch.  Wc = [Wpo | Wpo Wff2] and bc = bpo + Wpo bff2 are the folded
// net.2 + proj_out of engine.hip (w_fold_ff_proj); W1 is the plain net.0 weight in the value/gate row interleave of w_geglu (the kernel
// applies the LayerNorm gamma / beta to its register panel explicitly, it is NOT folded into W1).  Until round 3 this was two launches (GEGLU projection 84 us + K = 5C GEMM 53 us at M = 32768) with the 4C-wide hidden
// activation written to and re-read from HBM (2 x 84 MB per block); here the hidden activation never leaves the registers.
//
// Structure (256-thread workgroup = 4 waves, one per SIMD, 128 rows per workgroup):
//   * wave w keeps rows [32w, 32w+32) x K = 320 of t in registers as MFMA B fragments (v_mfma_f32_16x16x32_bf16, "transposed" issue as
//     in gemm.hip / rowpanel.hip: weights = A operand), 80 VGPRs, read from HBM once; LayerNorm statistics come from that panel.
//   * the out tile [32 rows][320] of the wave stays in 160 accumulator VGPRs for the whole launch.
//   * ALL weights arrive as one pre-arranged stream of 65 stages of 40 KiB (engine: ff_stream): 5 stages [320 n][64 k] of Wpo for the
//     t-part, then 20 x { W1 chunk 2j, W1 chunk 2j+1 ([64 n][320 k] each: 32 hidden units, 16 value | 16 gate rows interleaved),
//     Wc g-piece j ([320 n][64 k]) }.  Every stage is five [64][64] sub-tiles with the XOR-swizzled 16-byte chunks of gemm.hip, stored in
//     HBM exactly as the LDS image, so staging is a linear LDS-DMA copy (global_load_lds_dwordx4, 10 per wave and stage) through a ring
//     of three stages: two stages in flight, one barrier per stage.
//   * a W1 chunk gives acc1 = 4 x 2 fragments; its GEGLU epilogue (LN apply, bias, exact-erf GELU gate) leaves 32 hidden units x 32 rows
//     as ONE bf16x8 B fragment per 16-row tile: lane (row fr, quad fg) holds hidden {4fg..4fg+3} and {16+4fg..16+4fg+3} -- the k order of the
//     g-piece columns is permuted to exactly that at weight-conversion time, so the accumulator -> operand hand-over is lane-local
//     (no LDS round trip, no shuffles).
// Algorithmic work per launch at M = 32768: 87 GFLOP; HBM: t + x + out = 63 MB (+ 2.6 MB of weights, L2-resident per XCD).
#include "common.h"
#include <cstdlib>

namespace {

typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void glds16(const void* src, unsigned lds_wave_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(lds_wave_base) : "memory", "m0");
}
template <int N> __device__ __forceinline__ void wait_vmcnt() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

constexpr int FF_C = 320;
constexpr int FF_ROWS = 128;                 // rows per workgroup: 4 waves x 32 rows (MT = 2) or 8 waves x 16 rows (MT = 1)
constexpr int FF_SUB = 64 * 64;              // elements of one [64][64] sub-tile
constexpr int FF_STAGE = 5 * FF_SUB;         // elements of one stage (40 KiB)
constexpr int FF_NS = 3;                     // ring depth
constexpr int FF_TSTAGES = FF_C / 64;        // 5: t-part of the second GEMM
constexpr int FF_PAIRS = (4 * FF_C) / 64;    // 20: pairs of W1 chunks (64 hidden units per pair)
constexpr int FF_NSTAGES = FF_TSTAGES + 3 * FF_PAIRS;   // 65
constexpr int FF_PIECES = FF_STAGE * 2 / 1024;          // 40 LDS-DMA pieces (1 KiB) per stage: 10 per wave with 4 waves, 5 with 8

struct NrFFParams {
  const bf16* t; int ldt;        // [M][C] residual stream inside the transformer (pre-LayerNorm)
  const bf16* x; int ldx;        // [M][C] transformer input (outer residual)
  bf16* out; int ldo;            // [M][C]
  int M;
  int norot;                     // 1: every workgroup walks the weight stream from triple 0 (results independent of the row position: NR_DETERMINISTIC_BATCH)
  const bf16* stream;            // 65 stages x 40 KiB (ff_stream_pack_kernel)
  const float* gamma;            // [C] LayerNorm weight
  const float* beta;             // [C] LayerNorm bias
  const float* b1;               // [8C] net.0 bias in the value/gate-interleaved row order of the weight
  const float* bc;               // [C]  bpo + Wpo bff2
  float ln_eps;
  int dbg;               // timing experiments only (NR_FUSED_DBG): 1 no DMA waits, 2 no stage barriers, 4 no DMA issue (results are wrong)
};

// exact-erf GELU gate times the value, two outputs at a time (Abramowitz-Stegun 7.1.25 as gelu_erf_fast of common.h)
__device__ __forceinline__ float geglu1(float v, float g) { return v * gelu_erf_fast(g); }

// MT = 16-row tiles per wave.  MT = 2: four waves, one per SIMD (round 3).  MT = 1 (round 4): EIGHT waves of 16 rows, two per SIMD: the same
// 128-row workgroup, weight stream and LDS ring, but the matrix pipe of a SIMD now has a second wave to take MFMAs from while the first is in
// its GELU / packing VALU work (round 3 measured the one-wave form issue-bound: 83 k MFMA + 71 k VALU + 59 k wait cycles, nothing overlapping).
// Every weight fragment read from LDS then feeds one MFMA instead of two (the LDS array becomes as busy as the matrix pipe), which is the price.
template <int MT>
__global__ __launch_bounds__(MT == 2 ? 256 : 512, MT == 2 ? 1 : 2) void ff_fused_kernel(NrFFParams p) {
  constexpr int C = FF_C, KS = C / 32, NT2 = C / 16;       // 10 k-steps of the panel, 20 16-column groups of the output
  constexpr int NW = FF_ROWS / (16 * MT);                  // waves per workgroup
  constexpr int FF_DMA = FF_PIECES / NW;                   // LDS-DMA instructions per wave and stage
  constexpr int NTHR = 64 * NW;
  extern __shared__ __attribute__((aligned(16))) bf16 smem[];   // FF_NS stages, then b1 (8C floats)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;

  // ---- weight stream.  The 20 (W1, W1, g-piece) triples are independent up to the order of the fp32 accumulation, so every workgroup
  // walks them from its own starting triple: the 32 workgroups of an XCD then read 20 different regions of the stream instead of all
  // hammering the same 40 KiB (the same few L2 channels) in lockstep.  blockIdx % 8 labels the XCD (speed only; results depend on blockIdx
  // alone, so they are reproducible run to run). ----
  const int pair0 = p.norot ? 0 : (int)((blockIdx.x >> 3) % FF_PAIRS);
  const char* wsrc = reinterpret_cast<const char*>(p.stream) + (size_t)wave * (FF_DMA * 1024) + (size_t)lane * 16;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(lptr_t)smem) + (unsigned)wave * (FF_DMA * 1024);
  // logical stage s (order of execution) -> stage of the stream
  auto phys_stage = [&](int s) {
    if (s < FF_TSTAGES) return s;
    const int q = (s - FF_TSTAGES) / 3, r = (s - FF_TSTAGES) - 3 * q;
    int pr = pair0 + q;
    pr = pr >= FF_PAIRS ? pr - FF_PAIRS : pr;
    return FF_TSTAGES + 3 * pr + r;
  };
  auto issue_piece = [&](int s, int slot, int i) {      // one of the 10 LDS-DMA instructions of logical stage s
    const char* src = wsrc + (size_t)phys_stage(s) * (FF_STAGE * 2);
    glds16(src + i * 1024, lds0 + (unsigned)(slot * FF_STAGE * 2) + (unsigned)(i * 1024));
  };
#pragma unroll
  for (int i = 0; i < FF_DMA; ++i) issue_piece(0, 0, i);
#pragma unroll
  for (int i = 0; i < FF_DMA; ++i) issue_piece(1, 1, i);
  // the net.0 bias (8C floats) lives in LDS behind the ring: a global load inside the stage loop would make hipcc wait for it with a
  // vmcnt that also drains the LDS-DMA issued before it
  float* sB1 = reinterpret_cast<float*>(smem + FF_NS * FF_STAGE);
  for (int i = tid * 4; i < 8 * C; i += NTHR * 4) *(f32x4*)(sB1 + i) = *(const f32x4*)(p.b1 + i);

  // ---- the row panel: lane holds row (16 mt + fr) of its wave's 16 MT rows, k = 32 ks + 8 fg .. +7 ----
  const int mrow0 = blockIdx.x * FF_ROWS + wave * (16 * MT);
  bf16x8 xb[MT][KS];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    int m = mrow0 + 16 * mt + fr;
    m = m < p.M ? m : p.M - 1;
    const bf16* ap = p.t + (size_t)m * p.ldt + 8 * fg;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) xb[mt][ks] = *(const bf16x8*)(ap + 32 * ks);
  }
  __syncthreads();      // b1 is visible to every wave before the first chunk (the stage barriers are raw s_barrier)

  f32x4 oacc[NT2][MT];
#pragma unroll
  for (int nt = 0; nt < NT2; ++nt)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) oacc[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- stage sequencing: ring slot and waits are runtime values, everything that indexes registers is unrolled ----
  int s = 0, ring = 0;
  const char* pf_src = wsrc;
  unsigned pf_dst = lds0;
  auto stage_begin = [&]() -> const bf16* {
    // stage s must have landed: this wave's 10 DMA pieces of stage s + 1 (issued during stage s - 1) may stay outstanding.  No other
    // vector-memory operation is issued inside the loop, so the count is exact.
    if (!(p.dbg & 1)) { if (s + 1 < FF_NSTAGES) wait_vmcnt<FF_DMA>(); else wait_vmcnt<0>(); }
    if (!(p.dbg & 2)) __builtin_amdgcn_s_barrier();            // everyone's pieces of stage s landed; everyone is done reading stage s - 1
    // where the pieces of stage s + 2 come from / go to (the slot stage s - 1 occupied).  The last two stages re-fetch the final stage into
    // that free slot, so the piece issue stays unconditional (a branch per piece would cut the MFMA groups into separate scheduling regions)
    const int s2 = s + 2 < FF_NSTAGES ? s + 2 : FF_NSTAGES - 1;
    int sl = ring + 2;
    sl = sl >= FF_NS ? sl - FF_NS : sl;
    pf_src = wsrc + (size_t)phys_stage(s2) * (FF_STAGE * 2);
    pf_dst = lds0 + (unsigned)(sl * FF_STAGE * 2);
    return smem + ring * FF_STAGE;
  };
  // piece i of stage s + 2 goes out behind MFMA group i of stage s: spreads the ~100-cycle issue cost of an LDS-DMA piece over the stage
  // instead of stalling its head
  auto prefetch_piece = [&](int i) { if (i < FF_DMA && !(p.dbg & 4)) glds16(pf_src + i * 1024, pf_dst + (unsigned)(i * 1024)); };
  auto stage_end = [&]() { ++s; ring = ring + 1 == FF_NS ? 0 : ring + 1; };

  auto frag_n320 = [&](const bf16* sW, int nt, int ks2) {
    const int row = (nt & 3) * 16 + fr;
    return *(const bf16x8*)(sW + (nt >> 2) * FF_SUB + row * 64 + (((ks2 * 4 + fg) ^ (row & 7)) << 3));
  };
  // one 4-nt group of a [320 n][64 k] stage: 4 fragment reads (issued one group ahead by the caller) -> 8 MFMAs
  // out += A(stage [320 n][64 k]) x B(b0 | b1 per mt): 10 groups of (4 reads, 8 MFMAs); EPI(grp) is VALU work the caller wants
  // interleaved with group grp's MFMAs (the previous chunk's GEGLU epilogue)
  auto gemm_n320 = [&](const bf16* sW, const bf16x8 (&b0)[MT], const bf16x8 (&b1)[MT], auto&& epi) {
    bf16x8 fa[4], fb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[i] = frag_n320(sW, i, 0);
#pragma unroll
    for (int grp = 0; grp < 10; ++grp) {            // group = (ks2, 4 consecutive nt): grp = 5 ks2 + q
      const int ks2 = grp / 5, q = grp - 5 * ks2;
      bf16x8 (&cur)[4] = (grp & 1) ? fb : fa;
      bf16x8 (&nxt)[4] = (grp & 1) ? fa : fb;
      if (grp + 1 < 10) {
        const int g2 = grp + 1, k2 = g2 / 5, q2 = g2 - 5 * k2;
#pragma unroll
        for (int i = 0; i < 4; ++i) nxt[i] = frag_n320(sW, 4 * q2 + i, k2);
      }
      prefetch_piece(grp);
      __builtin_amdgcn_sched_barrier(0);
      epi(grp);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int nt = 4 * q + i;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) oacc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur[i], ks2 ? b1[mt] : b0[mt], oacc[nt][mt], 0, 0, 0);
      }
      // one MFMA, then the VALU / transcendental instructions that fit under it, once per MFMA of the group
#pragma unroll
      for (int i = 0; i < 4 * MT; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x402, 4, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto no_epi = [](int) {};

  // ---- t-part: out += t[:, 64 ts .. 64 ts + 63] . Wpo[:, same]^T  (raw t) ----
#pragma unroll
  for (int ts = 0; ts < FF_TSTAGES; ++ts) {
    const bf16* sW = stage_begin();
    bf16x8 b0[MT], b1[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) { b0[mt] = xb[mt][2 * ts]; b1[mt] = xb[mt][2 * ts + 1]; }
    gemm_n320(sW, b0, b1, no_epi);
    stage_end();
  }

  // ---- LayerNorm of the panel in registers (two-pass: mean, centred second moment), rounded to bf16 exactly where the un-fused LayerNorm
  // kernel rounds; gamma / beta come from global memory once (the two stages in flight are simply waited for here) ----
  {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      float sm = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) sm += (float)xb[mt][ks][e];
      sm += __shfl_xor(sm, 16, 64);
      sm += __shfl_xor(sm, 32, 64);
      const float mu = sm * (1.0f / C);
      float q = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = (float)xb[mt][ks][e] - mu; q += d * d; }
      q += __shfl_xor(q, 16, 64);
      q += __shfl_xor(q, 32, 64);
      const float rstd = rsqrtf(q * (1.0f / C) + p.ln_eps);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const f32x4 g0 = *(const f32x4*)(p.gamma + 32 * ks + 8 * fg), g1 = *(const f32x4*)(p.gamma + 32 * ks + 8 * fg + 4);
        const f32x4 e0 = *(const f32x4*)(p.beta + 32 * ks + 8 * fg), e1 = *(const f32x4*)(p.beta + 32 * ks + 8 * fg + 4);
        bf16x8 v = xb[mt][ks];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = (bf16)(((float)v[e] - mu) * rstd * g0[e] + e0[e]);
          v[4 + e] = (bf16)(((float)v[4 + e] - mu) * rstd * g1[e] + e1[e]);
        }
        xb[mt][ks] = v;
      }
    }
  }

  // W1 chunk: acc = b1 + xn . W1[chunk]^T.  EPI(ks) = VALU work interleaved with k-step ks (the previous chunk's GEGLU epilogue)
  auto gemm_w1 = [&](const bf16* sW, int chunk, f32x4 (&acc)[4][MT], auto&& epi) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const f32x4 bv = *(const f32x4*)(sB1 + chunk * 64 + 16 * nt + 4 * fg);      // lane's 4 columns of tile nt: the bias is the C operand
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = bv;
    }
    auto read_w = [&](bf16x8 (&wf)[4], int ks) {
      const int t = ks >> 1, k2 = ks & 1;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int row = nt * 16 + fr;
        wf[nt] = *(const bf16x8*)(sW + t * FF_SUB + row * 64 + (((k2 * 4 + fg) ^ (row & 7)) << 3));
      }
    };
    bf16x8 wf0[4], wf1[4], wf2[4];
    read_w(wf0, 0);
    read_w(wf1, 1);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bf16x8 (&wc)[4] = (ks % 3 == 0) ? wf0 : (ks % 3 == 1 ? wf1 : wf2);
      bf16x8 (&wn)[4] = (ks % 3 == 0) ? wf2 : (ks % 3 == 1 ? wf0 : wf1);
      if (ks + 2 < KS) read_w(wn, ks + 2);
      prefetch_piece(ks);
      __builtin_amdgcn_sched_barrier(0);
      epi(ks);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[nt], xb[mt][ks], acc[nt][mt], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4 * MT; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x402, 4, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // GEGLU epilogue of one chunk, cut into 4 MT slices of 2 outputs: slice i -> (mt = i >> 2, h = (i >> 1) & 1, e0 = 2 (i & 1)).
  // Lane holds rows (16 mt + fr); value tile 2 h, gate tile 2 h + 1, columns 4 fg + e.
  constexpr int NSL = 4 * MT;
  float gt[MT][8];
  auto epi_slice = [&](const f32x4 (&acc)[4][MT], int i) {
    const int mt = i >> 2, h = (i >> 1) & 1, e0 = 2 * (i & 1);
    float r0 = geglu1(acc[2 * h][mt][e0], acc[2 * h + 1][mt][e0]);
    float r1 = geglu1(acc[2 * h][mt][e0 + 1], acc[2 * h + 1][mt][e0 + 1]);
    // pin the slice where it is written: without the opaque statement LLVM sinks the whole epilogue (pure arithmetic) down to its first
    // use, the operand packing in front of the g-piece, where no MFMA is left to hide it
    asm volatile("" : "+v"(r0), "+v"(r1));
    gt[mt][4 * h + e0] = r0;
    gt[mt][4 * h + e0 + 1] = r1;
  };
  auto pack_g = [&](bf16x8& dst, int mt) {
#pragma unroll
    for (int e = 0; e < 8; ++e) dst[e] = (bf16)gt[mt][e];
  };

  bf16x8 gB[2][MT];           // [k-step of the g-piece = W1 chunk parity][mt]
  f32x4 accA[4][MT], accB[4][MT];
  for (int q = 0; q < FF_PAIRS; ++q) {
    int pair = pair0 + q;
    pair = pair >= FF_PAIRS ? pair - FF_PAIRS : pair;
    // ---- chunk A: MFMAs only ----
    {
      const bf16* sW = stage_begin();
      gemm_w1(sW, 2 * pair, accA, no_epi);
      stage_end();
    }
    // ---- chunk B: MFMAs with chunk A's GEGLU epilogue interleaved (slices 0..7 behind k-steps 0..7) ----
    {
      const bf16* sW = stage_begin();
      gemm_w1(sW, 2 * pair + 1, accB, [&](int ks) { if (ks < NSL) epi_slice(accA, ks); });
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) pack_g(gB[0][mt], mt);
      stage_end();
    }
    // ---- g-piece: out += g[:, 64 pair .. +63] . (Wpo Wff2)[:, same]^T.  k-step 0 (groups 0..4) needs chunk A only: chunk B's epilogue
    // rides on those groups (two slices each on groups 0..3) ----
    {
      const bf16* sW = stage_begin();
      bf16x8 fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = frag_n320(sW, i, 0);
#pragma unroll
      for (int grp = 0; grp < 10; ++grp) {
        const int ks2 = grp / 5, qq = grp - 5 * ks2;
        bf16x8 (&cur)[4] = (grp & 1) ? fb : fa;
        bf16x8 (&nxt)[4] = (grp & 1) ? fa : fb;
        if (grp + 1 < 10) {
          const int g2 = grp + 1, k2 = g2 / 5, q2 = g2 - 5 * k2;
#pragma unroll
          for (int i = 0; i < 4; ++i) nxt[i] = frag_n320(sW, 4 * q2 + i, k2);
        }
        prefetch_piece(grp);
        __builtin_amdgcn_sched_barrier(0);
        // chunk B's slices ride on groups 0..3: two per group with 32-row waves, one per group with 16-row waves
        if (grp < 4) {
#pragma unroll
          for (int i = 0; i < MT; ++i) epi_slice(accB, MT * grp + i);
        }
        if (grp == 4) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) pack_g(gB[1][mt], mt);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int nt = 4 * qq + i;
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) oacc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur[i], gB[ks2][mt], oacc[nt][mt], 0, 0, 0);
        }
        if (grp < 4) {
#pragma unroll
          for (int i = 0; i < 4 * MT; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x402, 8, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      stage_end();
    }
  }

  wait_vmcnt<0>();      // the tail's dummy pieces
#ifndef NR_FF_EPI16
#define NR_FF_EPI16 1
#endif
#if NR_FF_EPI16
  // ---- epilogue: out = x + bc + acc.  In the accumulator layout a lane holds rows (16 mt + fr), columns 16 nt + 4 fg .. +3: 8-byte accesses in 32-byte
  // row segments.  Round 5 (as xattn.hip / tattn.hip): v_permlane16_swap between the column tiles (2 k, 2 k + 1) hands every lane 8 CONSECUTIVE columns
  // (even lane rows: tile 2 k, columns 4 fg .. 4 fg + 7; odd: tile 2 k + 1, columns 4 (fg - 1) ..), so residual loads and stores are 16 bytes per lane in
  // 64-byte row segments and half as many.  The swap partners (fg ^ 1) hold the same row, so the row guard stays lane-consistent; same arithmetic ----
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = mrow0 + 16 * mt + fr;
    const bool ok = m < p.M;
#pragma unroll
    for (int k = 0; k < NT2 / 2; ++k) {
      f32x4 lo = oacc[2 * k][mt], hi = oacc[2 * k + 1][mt];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(lo[e]), __float_as_uint(hi[e]), false, false);
        lo[e] = __uint_as_float(sw[0]); hi[e] = __uint_as_float(sw[1]);
      }
      const int col0 = 16 * (2 * k + (fg & 1)) + 4 * (fg & 2);
      if (ok) {
        const bf16x8 xv = *(const bf16x8*)(p.x + (size_t)m * p.ldx + col0);
        const f32x4 b0 = *(const f32x4*)(p.bc + col0), b1 = *(const f32x4*)(p.bc + col0 + 4);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[e] = (bf16)(lo[e] + b0[e] + (float)xv[e]);
          o[4 + e] = (bf16)(hi[e] + b1[e] + (float)xv[4 + e]);
        }
        nr_store16(p.out + (size_t)m * p.ldo + col0, o);
      }
    }
  }
}
#else
  // ---- epilogue: out = x + bc + acc, lane holds rows (16 mt + fr), columns 16 nt + 4 fg .. +3 (A/B arm: -DNR_FF_EPI16=0) ----
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = mrow0 + 16 * mt + fr;
    if (m >= p.M) continue;
    const bf16* xr = p.x + (size_t)m * p.ldx + 4 * fg;
    bf16* orow = p.out + (size_t)m * p.ldo + 4 * fg;
#pragma unroll
    for (int nt = 0; nt < NT2; ++nt) {
      const bf16x4 xv = *(const bf16x4*)(xr + 16 * nt);
      const f32x4 b = *(const f32x4*)(p.bc + 16 * nt + 4 * fg);
      bf16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (bf16)(oacc[nt][mt][e] + b[e] + (float)xv[e]);
      nr_store8(orow + 16 * nt, o);
    }
  }
}
#endif

// Builds the 65-stage weight stream from W1 ([8C][C] bf16, value/gate-interleaved rows: engine w_geglu) and
// Wc ([C][5C] bf16 = [Wpo | Wpo Wff2], engine w_fold_ff_proj).  One thread per 16-byte chunk of the stream.
__global__ __launch_bounds__(256) void ff_stream_pack_kernel(const bf16* __restrict__ w1, const bf16* __restrict__ wc, bf16* __restrict__ stream) {
  constexpr int C = FF_C;
  const int idx = blockIdx.x * 256 + threadIdx.x;           // chunk index over the whole stream
  constexpr int CH_PER_STAGE = FF_STAGE / 8;
  if (idx >= FF_NSTAGES * CH_PER_STAGE) return;
  const int s = idx / CH_PER_STAGE, c = idx - s * CH_PER_STAGE;
  const int sub = c >> 9, row = (c >> 3) & 63, phys = c & 7;
  const int lchunk = phys ^ (row & 7);                      // logical 16-byte chunk (8 k values) stored at this physical slot
  bf16x8 v;
  if (s < FF_TSTAGES) {                                      // Wc[n = 64 sub + row][k = 64 s + 8 lchunk ..]
    v = *(const bf16x8*)(wc + (size_t)(64 * sub + row) * (5 * C) + 64 * s + 8 * lchunk);
  } else {
    const int r = (s - FF_TSTAGES) % 3, pair = (s - FF_TSTAGES) / 3;
    if (r < 2) {                                             // W1'[n = 64 chunk + row][k = 64 sub + 8 lchunk ..]
      const int chunk = 2 * pair + r;
      v = *(const bf16x8*)(w1 + (size_t)(64 * chunk + row) * C + 64 * sub + 8 * lchunk);
    } else {                                                 // g-piece: k position 32 ks2 + 8 fg + j  <->  hidden 32 (2 pair + ks2) + h(fg, j)
      const int ks2 = lchunk >> 2, fgq = lchunk & 3;
      const bf16* src = wc + (size_t)(64 * sub + row) * (5 * C) + C + 32 * (2 * pair + ks2);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = src[j < 4 ? 4 * fgq + j : 16 + 4 * fgq + (j - 4)];
    }
  }
  *(bf16x8*)(stream + (size_t)idx * 8) = v;
}

unsigned long long g_ff_attr = 0;
int g_ff_waves = -1;       // NR_FF_WAVES: 8 (default: 16-row waves, two per SIMD) or 4 (the round-3 form); nr_ff_set_waves overrides (A/B, tests)

}  // namespace

extern "C" void nr_ff_set_waves(int waves) { g_ff_waves = waves == 4 ? 4 : 8; }
extern "C" size_t nr_ff_stream_bytes(int C) { return C == FF_C ? (size_t)FF_NSTAGES * FF_STAGE * sizeof(bf16) : 0; }

extern "C" int nr_ff_fused_eligible(int C, long long M) {
  static const bool off = getenv("NR_FF_FUSED") && getenv("NR_FF_FUSED")[0] == '0';   // A/B switch
  return !off && C == FF_C && M >= 4096;
}

extern "C" int nr_launch_ff_stream_pack(const bf16* w1, const bf16* wc, bf16* stream, hipStream_t s) {
  const int total = FF_NSTAGES * (FF_STAGE / 8);
  hipLaunchKernelGGL(ff_stream_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, s, w1, wc, stream);
  return 0;
}

extern "C" int nr_launch_ff_fused(const bf16* t, int ldt, const bf16* x, int ldx, bf16* out, int ldo, int M, const bf16* stream,
                                  const float* gamma, const float* beta, const float* b1, const float* bc, float ln_eps, int norot, hipStream_t s) {
  if (M <= 0 || ldt % 8 != 0 || ldx % 8 != 0 || ldo % 8 != 0) return 1;
  NrFFParams p;
  p.t = t; p.ldt = ldt; p.x = x; p.ldx = ldx; p.out = out; p.ldo = ldo; p.M = M; p.stream = stream; p.gamma = gamma; p.beta = beta; p.b1 = b1; p.bc = bc;
  p.ln_eps = ln_eps; p.norot = norot;
  static const int dbg = getenv("NR_FUSED_DBG") ? atoi(getenv("NR_FUSED_DBG")) : 0;
  p.dbg = dbg;
  constexpr size_t shm = (size_t)FF_NS * FF_STAGE * sizeof(bf16) + (size_t)8 * FF_C * sizeof(float);
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!(g_ff_attr >> (dev & 63) & 1ull)) {
    if (hipFuncSetAttribute((const void*)ff_fused_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess) return 2;
    if (hipFuncSetAttribute((const void*)ff_fused_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess) return 2;
    g_ff_attr |= 1ull << (dev & 63);
  }
  if (g_ff_waves < 0) g_ff_waves = getenv("NR_FF_WAVES") ? atoi(getenv("NR_FF_WAVES")) : 8;
  const unsigned grid = (unsigned)((M + FF_ROWS - 1) / FF_ROWS);
  if (g_ff_waves == 4) hipLaunchKernelGGL(ff_fused_kernel<2>, dim3(grid), dim3(256), shm, s, p);
  else hipLaunchKernelGGL(ff_fused_kernel<1>, dim3(grid), dim3(512), shm, s, p);
  return 0;
}
