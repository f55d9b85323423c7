// Row-panel GEMM for the short-K Linears of the 32x32 level (K = C = 320): out = epilogue( [LayerNorm(+PE)](A) . W^T ).
//
// The transformer Linears with K = C (proj_in/out, to_q|k|v, to_out, GEGLU FF1; reference: animatediff/models/attention.py:
// 107-129,256-300, motion_module.py:134-158,210-222, motion_module_new.py:181-193,441-518) have only 5 k-tiles at C = 320:
// a tiled GEMM spends its time in prologues/epilogues and re-reads the activation panel once per n-tile.  Here a
// 512-thread workgroup owns 256 rows and keeps them IN REGISTERS for the whole launch:
//   * wave w holds rows [32w, 32w+32) x all K as MFMA B-operand fragments (v_mfma_f32_16x16x32_bf16, "transposed" issue as in
//     gemm.hip: weights = A operand, activations = B operand): 80 VGPRs at C = 320.  The activation panel is read from HBM
//     exactly once per workgroup (nsplit workgroups share a panel: placed on one XCD so the second read is an L2 hit).
//   * LayerNorm is folded into the GEMM as in gemm.hip (gamma-scaled weights, out = rstd (acc - mean c) + b'); the row statistics
//     come from the register panel once per workgroup (v_dot2_f32_bf16), not once per n-tile; the temporal positional encoding
//     (motion_module.py:241-243,274-278) enters as the fp32 row vector pe[f].W^T in the epilogue: no LayerNorm pass over HBM.
//   * the workgroup then walks its range of 64-column chunks of W: each chunk ([64 n][C k] = 40 KiB) is one LDS-DMA stage in a
//     ring of three (two chunks in flight, one barrier per chunk = 80 MFMAs per wave), laid out as C/64 [64][64] sub-tiles with
//     the XOR-swizzled 16-byte chunks of gemm.hip (conflict-free ds_read_b128).  Per byte of W staged the 8 waves issue 256 rows
//     of MFMAs (intensity 256 flop/B: half the L2->LDS fill rate a 256x256 tile needs); LDS reads are 50 % of the MFMA time.
//   * epilogue per chunk in fragment layout (bias, scale, quick_gelu, GEGLU, residual: one rounding to bf16), then a
//     wave-private LDS transpose so that every global store is 16 bytes per lane over whole 128-byte row segments.
//   All vector-memory instructions of an iteration have fixed counts (raw buffer loads/stores with range checking instead of
//   branches), which is what makes the counted s_waitcnt in front of the barrier valid.
#include "common.h"
#include <cstdlib>

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// LDS-DMA (global_load_lds_dwordx4) as inline asm: the compiler must NOT see it.  A builtin LDS-DMA is a pending LDS write to
// hipcc, which then puts s_waitcnt vmcnt(0) in front of the next ds_read that may alias it, i.e. at the top of every iteration,
// draining the chunks in flight AND the previous iteration's output stores (measured: 4.2 us per chunk instead of ~1.3).  Hidden
// in asm, only the counted wait + barrier below order the DMA against the fragment reads (cdna_hip_programming.md 5.7).  M0 (the
// wave-uniform LDS destination) is written and restored inside the same statement.
__device__ __forceinline__ void glds16(const void* src, unsigned lds_wave_base) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src), "s"(lds_wave_base) : "memory");
}

// Output store as inline asm, for the same reason: hipcc keeps the data registers of a store it issued locked until the store has
// RETIRED (it puts s_waitcnt vmcnt(0) in front of the next write to them, i.e. right after the next barrier), which serialises
// every chunk behind the previous chunk's HBM writes.  The hardware only needs the registers for two wait states (s_nop 1).
// rsrc: the four descriptor words in SGPRs; off: byte offset per lane (range-checked: out-of-range lanes are dropped).
__device__ __forceinline__ void buffer_store16(const u32x4& v, const u32x4& rsrc, unsigned off) {
#if NR_STORE_WT
  asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen sc1\n\ts_nop 1" : : "v"(v), "v"(off), "s"(rsrc) : "memory");
#else
  asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" : : "v"(v), "v"(off), "s"(rsrc) : "memory");
#endif
}

// s_waitcnt vmcnt(n) for a wave-uniform runtime n in [0, 15]
__device__ __forceinline__ void wait_vmcnt_dyn(int n) {
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
    case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
  }
}

__device__ __attribute__((aligned(16))) const float rp_zero_bias[4096] = {0.f};   // stands in for a null bias (N <= 4096)

constexpr int RP_ROWS = 256;          // rows per workgroup (8 waves x 32)
constexpr int RP_NS = 3;              // W ring depth (chunks)
constexpr unsigned RP_OOB = 0x7fffff00u;   // byte offset beyond any buffer: the range check drops the lane

// GEGLU gate: gelu_erf_fast of common.h (A&S 7.1.25; the epilogue is what bounds this kernel at K = 320)
__device__ __forceinline__ float gelu_gate(float x) { return gelu_erf_fast(x); }

// LN: LayerNorm folded (p.ln_c); RV: fp32 row-vector term (p.rowvec); RES: residual add (p.res); GEGLU: value * gelu(gate) epilogue
template <int C, bool LN, bool RV, bool RES, bool GEGLU>
__global__ __launch_bounds__(512) void rowpanel_kernel(NrRowPanelParams p) {
  constexpr int KS = C / 32;          // 32-deep MFMA k-steps
  constexpr int KT = C / 64;          // [64][64] sub-tiles per chunk
  constexpr int SUB = 64 * 64;        // elements of one sub-tile
  constexpr int STAGE = KT * SUB;     // elements of one chunk of W
  constexpr int NQ = GEGLU ? 2 : 4;   // 16-column output groups per chunk
  constexpr int NSTORE = GEGLU ? 2 : 4;   // 16-byte store instructions per chunk and wave
  extern __shared__ __attribute__((aligned(16))) bf16 smem[];   // RP_NS * STAGE ring, then 8 x 2048-element wave scratch

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;          // waves w and w + 4 share a SIMD: the two groups run half an iteration apart (see below)
  const int fr = lane & 15, fg = lane >> 4;
  const int lr = lane >> 3, lp = lane & 7;

  // workgroups that share a row panel differ by 8 in blockIdx (same XCD under round-robin placement: speed only)
  int mblk, part;
  {
    const int b = blockIdx.x, ns = p.nsplit, g8 = 8 * ns;
    if ((int)gridDim.x % g8 == 0) { const int q = b / g8, r = b - q * g8; mblk = q * 8 + (r & 7); part = r >> 3; }
    else { mblk = b / ns; part = b - mblk * ns; }
  }
  const int NC = p.N >> 6;
  const int c_begin = (int)(((long long)NC * part) / p.nsplit);
  const int c_end = (int)(((long long)NC * (part + 1)) / p.nsplit);
  const int nc = c_end - c_begin;

  // ---- W staging: wave w moves rows [8w, 8w+8) of every sub-tile; the 16-byte chunk a lane fetches is XOR-swizzled ----
  const bf16* wnext = p.w + (size_t)(c_begin * 64 + wave * 8 + lr) * C + ((lp ^ lr) << 3);
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(lptr_t)(smem + wave * 8 * 64));
  auto issue = [&](int buf) {
    const unsigned dst = lds0 + (unsigned)(buf * STAGE * (int)sizeof(bf16));
#pragma unroll
    for (int t = 0; t < KT; ++t) glds16(wnext + 64 * t, dst + (unsigned)(t * SUB * (int)sizeof(bf16)));
    wnext += (size_t)64 * C;
  };
  if (nc > 0) issue(0);
  if (nc > 1) issue(1);

  // ---- the row panel: lane holds row (16 mt + fr) of its wave's 32 rows, k = 32 ks + 8 fg .. +7 ----
  const int mrow0 = mblk * RP_ROWS + wave * 32;
  bf16x8 xb[2][KS];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    int m = mrow0 + 16 * mt + fr;
    m = m < p.M ? m : p.M - 1;
    const bf16* ap = p.a + (size_t)m * p.lda + 8 * fg;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) xb[mt][ks] = *(const bf16x8*)(ap + 32 * ks);
  }
  // LayerNorm folded into the GEMM exactly as in gemm.hip (LNF): w holds gamma-scaled rows, ln_c[n] = sum_k w[n][k], bias holds
  // b + beta.W; the epilogue applies out = rstd_m (acc - mean_m ln_c[n]) + bias.  The row statistics come straight from the packed
  // bf16 pairs of the register panel (v_dot2_f32_bf16, fp32 accumulation; var = E[x^2] - mean^2), once per workgroup.
  float rs[2] = {1.f, 1.f}, mr[2] = {0.f, 0.f};      // rstd_m and mean_m * rstd_m
  if constexpr (LN) {
    const bf16x2 one2 = {(bf16)1.0f, (bf16)1.0f};
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bf16x2 pr = {xb[mt][ks][2 * e], xb[mt][ks][2 * e + 1]};
          s = __builtin_amdgcn_fdot2_f32_bf16(pr, one2, s, false);
          q = __builtin_amdgcn_fdot2_f32_bf16(pr, pr, q, false);
        }
      s += __shfl_xor(s, 16, 64); q += __shfl_xor(q, 16, 64);
      s += __shfl_xor(s, 32, 64); q += __shfl_xor(q, 32, 64);
      const float mu = s * (1.0f / C);
      rs[mt] = rsqrtf(fmaxf(q * (1.0f / C) - mu * mu, 0.f) + p.ln_eps);
      mr[mt] = mu * rs[mt];
    }
  }
  if (p.out_scale != 1.0f) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) { rs[mt] *= p.out_scale; mr[mt] *= p.out_scale; }
  }
  // row of the fp32 row-vector term (temporal PE pushed through the projection: engine pe_projection) for this lane's rows
  const float* rvp[2] = {nullptr, nullptr};
  if constexpr (RV) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      int m = mrow0 + 16 * mt + fr;
      m = m < p.M ? m : p.M - 1;
      int r = m / p.rowvec_div;
      if (p.rowvec_mod) r %= p.rowvec_mod;
      rvp[mt] = p.rowvec + (size_t)r * p.rowvec_ld + 4 * fg;
    }
  }

  // ---- loop-invariant per-lane addresses ----
  const int ncol_out = GEGLU ? p.N >> 1 : p.N;
  u32x4 rs_out;      // descriptor words through readfirstlane: provably wave-uniform (an "s" operand of the asm store)
  {
    const unsigned long long v = (unsigned long long)p.out;
    rs_out[0] = __builtin_amdgcn_readfirstlane((unsigned)v);
    rs_out[1] = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32) & 0xffffu);     // stride 0
    rs_out[2] = (unsigned)__builtin_amdgcn_readfirstlane((int)(((size_t)(p.M - 1) * p.ldo + ncol_out) * 2));
    rs_out[3] = 0x00020000u;
  }
  auto uniform_ptr = [](const void* q) {
    const unsigned long long v = (unsigned long long)q;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (void*)(((unsigned long long)hi << 32) | lo);
  };
  const int res_bytes = __builtin_amdgcn_readfirstlane((int)(((size_t)(p.M - 1) * (RES ? p.ldr : p.ldo) + ncol_out) * 2));
  const auto rs_res = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(RES ? (const void*)p.res : (const void*)p.out), 0, res_bytes, 0x00020000);
  unsigned res_off[2];       // byte offset of (row, column 4 fg) in the residual; out-of-range rows sit beyond the buffer
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int m = mrow0 + 16 * mt + fr;
    res_off[mt] = m < p.M ? (unsigned)(((size_t)m * p.ldr + 4 * fg) * 2) : RP_OOB;
  }
  bf16* scratch = smem + RP_NS * STAGE + wave * 2048;
  // wave-private transpose tile [32 rows][64 columns] bf16, 16-byte chunks XOR-swizzled with (row & 7)
  bf16* sc_wr[2][NQ];        // where the fragment of (mt, q) goes: row 16 mt + fr, columns 16 q + 4 fg .. +3
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int row = 16 * mt + fr, chunk = 2 * q + (fg >> 1);
      sc_wr[mt][q] = scratch + row * 64 + ((chunk ^ (row & 7)) << 3) + ((fg & 1) << 2);
    }
  const bf16* sc_rd[NSTORE];  // 16 bytes per lane, whole 64- / 128-byte row segments per store instruction
  unsigned out_off[NSTORE];
#pragma unroll
  for (int j = 0; j < NSTORE; ++j) {
    const int idx = lane + 64 * j;
    const int row = GEGLU ? idx >> 2 : idx >> 3, chunk = GEGLU ? idx & 3 : idx & 7;
    sc_rd[j] = scratch + row * 64 + ((chunk ^ (row & 7)) << 3);
    const int m = mrow0 + row;
    out_off[j] = m < p.M ? (unsigned)(((size_t)m * p.ldo + chunk * 8) * 2) : RP_OOB;
  }
  // both sides of the select are global-address-space pointers: a generic (flat) pointer here makes the bias loads flat_load,
  // which hipcc can only order with s_waitcnt vmcnt(0)
  typedef const __attribute__((address_space(1))) float* gfp_t;
  typedef const __attribute__((address_space(1))) f32x4* gf4p_t;
  const gfp_t bias = p.bias ? (gfp_t)p.bias : (gfp_t)rp_zero_bias;

  // ---- phases.  The two waves of a SIMD (w and w + 4, one from each group) alternate roles: while group 0 issues the 80 MFMAs
  // of chunk i, group 1 runs the VALU epilogue of chunk i-1, and vice versa, so the matrix pipe and the vector ALU of every SIMD
  // are busy at the same time (with both waves in the same role the epilogue time is simply added to the MFMA time: measured
  // 3.4 us per chunk for 1.1 us of MFMA work).  Group g runs MFMA(i) in phase 2 i + g and epilogue(i) in phase 2 i + g + 1; one
  // barrier per phase.  Ring safety: chunk i is first read in phase 2 i, so EVERY wave retires its DMA pieces of chunk i before
  // that phase's barrier; chunk i+2 overwrites the buffer of chunk i-1, last read in phase 2 i - 1, and is issued in phases
  // 2 i (group 0) and 2 i + 1 (group 1), i.e. behind the barrier that follows that read. ----
  f32x4 acc[4][2], bv[4], cv[4], rv[2][4];
  u32x2 rr[2][4];
  const int nphase = 2 * nc + 1;
  for (int ph = 0; ph <= nphase; ++ph) {
    if (ph == nphase && grp == 0) break;                 // group 1 needs one more phase (no barrier) for its last epilogue
    if (ph < nphase) {
      if ((ph & 1) == 0 && (ph >> 1) < nc) {
        // Vector-memory operations retire in issue order.  After its DMA of chunk i a wave has issued at least
        // [NSTORE stores][KT DMA of chunk i+1] (plus epilogue loads): waiting until at most that many operations are
        // outstanding retires chunk i without draining chunk i+1 or the latest output stores.
        const int i = ph >> 1;
        if (i == 0 || (p.dbg & 4)) wait_vmcnt_dyn(0);
        else wait_vmcnt_dyn(i + 1 < nc ? KT + NSTORE : NSTORE);
      }
      __builtin_amdgcn_s_barrier();
    }
    const int q = ph - grp;
    if (q >= 0 && (q & 1) == 0 && (q >> 1) < nc) {
      // =============================== MFMA phase of chunk i ===============================
      const int i = q >> 1;
      const int cur = i % RP_NS;
      const int cg = c_begin + i;
      const int nw0 = cg * 64;                              // first W row / bias index of the chunk
      if (i + 2 < nc && !(p.dbg & 2)) issue((i + 2) % RP_NS);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) acc[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
      const bf16* sW = smem + cur * STAGE;
      // Fragment reads run ONE k-step ahead of the MFMAs (two register sets of 4 fragments): left to itself hipcc schedules
      // "1 ds_read, s_waitcnt lgkmcnt(0), 2 MFMAs", an exposed LDS round trip per MFMA pair.  sched_barrier(0) fences pin the
      // order [4 reads of k-step s+1][8 MFMAs of k-step s].
      auto read_w = [&](bf16x8 (&wf)[4], int ks) {
        const int t = ks >> 1, k2 = ks & 1;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const int row = nt * 16 + fr;
          wf[nt] = *(const bf16x8*)(sW + t * SUB + row * 64 + (((k2 * 4 + fg) ^ (row & 7)) << 3));
        }
      };
      // three register sets: the reads of k-step s+2 are issued before the MFMAs of k-step s.  One k-step of MFMAs (128 cycles)
      // does not cover the LDS latency when the four waves of a group read in lockstep (measured 30 cycles per MFMA with the
      // reads one k-step ahead).  The LN + row-vector instantiation has no registers left for the third set.
      constexpr int PF = (LN && RV) ? 1 : 2;       // k-steps the reads run ahead
      bf16x8 wf0[4], wf1[4], wf2[4];
      read_w(wf0, 0);
      if constexpr (PF == 2) read_w(wf1, 1);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (ks == 4) {
          // the epilogue's loads go out mid-chunk: the remaining MFMAs and the partner's phase cover their latency
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) bv[nt] = *(gf4p_t)(bias + nw0 + 16 * nt + 4 * fg);
          if constexpr (LN) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) cv[nt] = *(const f32x4*)(p.ln_c + nw0 + 16 * nt + 4 * fg);
          }
          if constexpr (RV) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
              for (int nt = 0; nt < 4; ++nt) rv[mt][nt] = *(const f32x4*)(rvp[mt] + nw0 + 16 * nt);
          }
          if constexpr (RES) {
            const unsigned cb = (unsigned)((GEGLU ? cg * 32 : cg * 64) * 2);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
              for (int qq = 0; qq < NQ; ++qq) rr[mt][qq] = __builtin_amdgcn_raw_buffer_load_b64(rs_res, res_off[mt] + cb + 32 * qq, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (PF == 2) {
          bf16x8 (&wc)[4] = (ks % 3 == 0) ? wf0 : (ks % 3 == 1 ? wf1 : wf2);
          bf16x8 (&wn)[4] = (ks % 3 == 0) ? wf2 : (ks % 3 == 1 ? wf0 : wf1);
          if (ks + 2 < KS) read_w(wn, ks + 2);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
              acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[nt], xb[mt][ks], acc[nt][mt], 0, 0, 0);
        } else {
          bf16x8 (&wc)[4] = (ks & 1) ? wf1 : wf0;
          bf16x8 (&wn)[4] = (ks & 1) ? wf0 : wf1;
          if (ks + 1 < KS) read_w(wn, ks + 1);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
              acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[nt], xb[mt][ks], acc[nt][mt], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if (q >= 1 && (q & 1) == 1 && ((q - 1) >> 1) < nc) {
      // =============================== epilogue phase of chunk i ===============================
      const int i = (q - 1) >> 1;
      const int cg = c_begin + i;
      const unsigned no0b = (unsigned)((GEGLU ? cg * 32 : cg * 64) * 2);     // byte offset of the chunk's first output column
      // The loads of this chunk were issued a whole phase ago.  As the BUILTIN the wait is visible to hipcc, which then carries
      // no "load still pending" state across the loop back-edge (it would pay for that with vmcnt waits behind the next barrier).
      __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) only
      // lane holds out[row 16 mt + fr][col 16 nt + 4 fg + r]:  out = rstd (acc - mean c) + bias (+ rowvec) (+ residual)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int qq = 0; qq < NQ; ++qq) {
          f32x4 o4;
          if constexpr (!GEGLU) {
            f32x4 t = bv[qq];
            if (p.out_scale != 1.0f) t *= p.out_scale;
            if constexpr (LN) t -= cv[qq] * mr[mt];
            if constexpr (RV) t += p.out_scale != 1.0f ? rv[mt][qq] * p.out_scale : rv[mt][qq];
            o4 = acc[qq][mt] * rs[mt] + t;
          } else {                                            // W rows are (16 value | 16 gate)-interleaved (engine: w_geglu)
            f32x4 tv = bv[2 * qq], tg = bv[2 * qq + 1];
            if constexpr (LN) { tv -= cv[2 * qq] * mr[mt]; tg -= cv[2 * qq + 1] * mr[mt]; }
            const f32x4 v = acc[2 * qq][mt] * rs[mt] + tv, g = acc[2 * qq + 1][mt] * rs[mt] + tg;
#pragma unroll
            for (int e = 0; e < 4; ++e) o4[e] = v[e] * gelu_gate(g[e]);
          }
          if constexpr (RES) {
            const bf16x4 r = __builtin_bit_cast(bf16x4, rr[mt][qq]);
#pragma unroll
            for (int e = 0; e < 4; ++e) o4[e] += (float)r[e];
          }
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (bf16)o4[e];
          *(bf16x4*)sc_wr[mt][qq] = o;
        }
      }
      // wave-private transpose.  LDS operations of one wave execute in order; the empty asm only keeps the compiler from moving
      // the reads above the writes
      asm volatile("" ::: "memory");
#pragma unroll
      for (int j = 0; j < NSTORE; ++j) {
        const u32x4 v = *(const u32x4*)sc_rd[j];
        unsigned off = out_off[j] + no0b;
        if (p.dbg & 1) off = RP_OOB;
        buffer_store16(v, rs_out, off);
      }
    }
  }
}

unsigned long long g_attr_mask = 0;   // per-device opt-in for > 64 KiB of dynamic LDS

}  // namespace

// shapes this kernel serves (the caller falls back to the tiled igemm otherwise)
extern "C" int nr_rowpanel_eligible(const NrGemmParams* pp) {
  const NrGemmParams& p = *pp;
  static const bool off = getenv("NR_ROWPANEL") && getenv("NR_ROWPANEL")[0] == '0';   // A/B switch
  if (off) return 0;
  if (p.ksize != 1 || p.a1 || p.c1 != 0 || p.out_f32) return 0;
  if (p.K != 320 || p.N % 64 != 0 || p.N > 4096 || (p.plan_m > 0 && p.plan_m < p.M ? p.plan_m : p.M) < 4096) return 0;
  if (p.act || (p.geglu && p.rowvec)) return 0;
  if (p.ln_c && p.rowvec && p.res) return 0;            // the one epilogue combination that does not fit 256 VGPRs (and never occurs)
  if (p.lda0 % 8 != 0 || p.ldo % 8 != 0 || (p.res && p.ldr % 8 != 0)) return 0;
  if (p.rowvec && (p.rowvec_div <= 0 || p.rowvec_ld % 4 != 0)) return 0;
  const int ncol = p.geglu ? p.N / 2 : p.N;
  if (((size_t)p.M * (size_t)(p.ldo > p.ldr ? p.ldo : p.ldr) + ncol) * 2 >= 0x7fffff00ull) return 0;   // 32-bit buffer offsets
  return 1;
}

extern "C" int nr_launch_rowpanel(const NrGemmParams* pp, hipStream_t stream) {
  const NrGemmParams& g = *pp;
  if (!nr_rowpanel_eligible(pp)) return 1;
  NrRowPanelParams p;
  p.a = g.a0; p.lda = g.lda0; p.w = g.w; p.M = g.M; p.N = g.N; p.bias = g.bias;
  p.ln_c = g.ln_c; p.ln_eps = g.ln_eps;
  p.rowvec = g.rowvec; p.rowvec_div = g.rowvec_div > 0 ? g.rowvec_div : 1; p.rowvec_mod = g.rowvec_mod; p.rowvec_ld = g.rowvec_ld;
  p.res = g.res; p.ldr = g.ldr; p.out = g.out; p.ldo = g.ldo; p.out_scale = g.out_scale; p.geglu = g.geglu; p.act = g.act;
  const int mblocks = (g.M + RP_ROWS - 1) / RP_ROWS;
  const int NC = g.N / 64;
  int ns = (256 + mblocks - 1) / mblocks;       // fill the 256 CUs
  if (ns > NC) ns = NC;
  if (ns < 1) ns = 1;
  p.nsplit = ns;
  static const int dbg = getenv("NR_RP_DBG") ? atoi(getenv("NR_RP_DBG")) : 0;
  p.dbg = dbg;
  if (getenv("NR_RP_NSPLIT")) { p.nsplit = ns = atoi(getenv("NR_RP_NSPLIT")); }
  constexpr size_t shm = (size_t)(RP_NS * 5 * 64 * 64 + 8 * 2048) * sizeof(bf16);
  int dev = 0;
  (void)hipGetDevice(&dev);
  const bool first = !(g_attr_mask >> (dev & 63) & 1ull);
  typedef void (*kern_t)(NrRowPanelParams);
  // index: LN 8 | RV 4 | RES 2 | GEGLU 1 (GEGLU never carries a row vector; LN + RV + RES does not fit 256 VGPRs and never occurs)
  static const kern_t ks[16] = {
      rowpanel_kernel<320, false, false, false, false>, rowpanel_kernel<320, false, false, false, true>,
      rowpanel_kernel<320, false, false, true, false>,  rowpanel_kernel<320, false, false, true, true>,
      rowpanel_kernel<320, false, true, false, false>,  nullptr,
      rowpanel_kernel<320, false, true, true, false>,   nullptr,
      rowpanel_kernel<320, true, false, false, false>,  rowpanel_kernel<320, true, false, false, true>,
      rowpanel_kernel<320, true, false, true, false>,   rowpanel_kernel<320, true, false, true, true>,
      rowpanel_kernel<320, true, true, false, false>,   nullptr,
      nullptr,                                          nullptr};
  if (first) {    // opt every instantiation of this device into > 64 KiB of dynamic LDS once
    for (auto k : ks)
      if (k && hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess) return 2;
    g_attr_mask |= 1ull << (dev & 63);
  }
  const kern_t k = ks[(g.ln_c ? 8 : 0) | (g.rowvec ? 4 : 0) | (g.res ? 2 : 0) | (g.geglu ? 1 : 0)];
  if (!k) return 3;
  hipLaunchKernelGGL(k, dim3((unsigned)(mblocks * ns)), dim3(512), shm, stream, p);
  return 0;
}
