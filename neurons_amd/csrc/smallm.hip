// Panel-resident GEMM for the small-M Linears (M <= 512: the 4x4 level of the AnimateDiff U-Net, the depth-10 16x16 level of the sgm
// keyframe UNetModel with CFG batch 2): out = epilogue( [LayerNorm](A) . W^T ), K a multiple of 640.  Reference operators: the
// nn.Linear layers of BasicTransformerBlock (animatediff/models/attention.py:256-300; generative_models/sgm/modules/attention.py:
// 551-572 BasicTransformerBlock, :340-400 CrossAttention to_q/to_k/to_v/to_out, :60-100 FeedForward / GEGLU), proj_in / proj_out
// (:690-742) and the temporal transformer Linears (motion_module.py:134-158,210-222).
//
// Why a second GEMM kernel.  On these shapes the tiled igemm is bound by neither MFMA nor HBM: a workgroup's k-tile waits for one LDS-DMA
// round trip with at most 3 tiles (48 KB) in flight per CU, ~20 B/clk per CU of ingest, and a 320-tile launch takes two rounds of workgroups
// (profiles/r05_igemm_timeline_smallm_fm.txt: 9.6 us kernel span for M = 512, N = K = 1280 = 1.7 GFLOP).  This kernel keeps ~280 KB in flight
// per CU by using the register file as the landing zone of the weights:
//   * the weights are stored FRAGMENT-MAJOR (NrGemmParams::w_fm, packed once at plan time by nr_launch_smallm_w_pack): the MFMA A-operand
//     of (16 W rows, 32 k) is one contiguous KiB, so a wave's weight load touches 8 cache lines instead of 64 (58-80 B/clk per CU against
//     16-21 for the same fragment read from row-major weights, profiles/r05_dma_issue_reg_arms.txt);
//   * a workgroup owns 32 rows x (16 NT) W rows (NT = 5 or 4 n-tiles; two wave groups split every 640-deep K chunk), all of K, and J
//     such column slabs one after the other: the grid is (M / 32) x G <= 256 workgroups = ONE round on the chip, the 32 x K activation
//     panel is read once per workgroup (LDS, 40-KiB chunks in a 3-slot ring; resident when K <= 1920), and only weights stream;
//   * per 640-deep chunk and wave: 10 contiguous 1-KiB weight loads into one of two register banks (each register refilled with the weights of
//     two chunks ahead right behind the MFMAs that read it), 20 MFMAs, ONE barrier;
//   * LayerNorm fold (row statistics from the panel, first slab), GEGLU (value / gate n-tiles on neighbouring waves), bias / row vector /
//     scale / quick_gelu / residual epilogue in fragment layout, as the igemm's.
//   * two-source operands [a0 | a1] (the skip concat as a 1x1 GEMM, the [t | g] operand of the folded net.2|proj_out): whole 640-deep chunks
//     from either source.
// Shipped plans have ONE slab per workgroup (N <= 1280 at M = 512): 1.1-1.66x the tiled igemm per launch; several slabs (N >= 1920) lose to
// the igemm's bigger tiles and stay there (NR_SMALLM=2 keeps them for the tests).  What bounds it (profiles/r05_smallm_timeline.txt): the ISSUE
// of 100 KB of weight loads + 40 KB of panel DMA per chunk through the CU's ~64 B/clk address path (2.2-2.5 k cycles for 320 MFMA cycles per
// wave), with the wait + barrier of a chunk (1.0-1.9 k) not overlapped.
// Round 3 had rejected the row-major form of this kernel (0.62-0.68x the igemm: 40 fragment-shaped loads took 8,400 cycles to ISSUE);
// DESIGN_HISTORY.md has that story, profiles/r05_smallm_fm_ab.txt the first fragment-major A/B.
#include "common.h"
#include <cstdlib>

namespace {

__device__ __attribute__((aligned(16))) const unsigned int smallm_zero16[4] = {0u, 0u, 0u, 0u};
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void glds16_asm(const void* src, unsigned lds_wave_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(lds_wave_base) : "memory", "m0");
}
template <int N> __device__ __forceinline__ void wait_vmcnt_c() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }

__device__ __forceinline__ size_t rowvec_row(const NrGemmParams& p, int m) {
  int r = m / p.rowvec_div;
  if (p.rowvec_mod) r %= p.rowvec_mod;
  return (size_t)r * p.rowvec_ld;
}

#ifdef NR_STAMP
// diagnostic build only (make stamp, tools/igemm_timeline.py): shader-clock stamps of wave 0 of the first 512 workgroups
// (slots 6 / 7: the chip-wide 100 MHz counter at entry / exit)
__device__ unsigned long long smallm_stamp_buf[512][40];      // 8 + 3 t: step t < 10 after its wait / after its barrier / after its MFMAs
#define SM_STAMP_AT(slot) do { if (threadIdx.x == 0 && blockIdx.x < 512 && (slot) < 40) smallm_stamp_buf[blockIdx.x][(slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#define SM_STAMP_RT(slot) do { if (threadIdx.x == 0 && blockIdx.x < 512) smallm_stamp_buf[blockIdx.x][(slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define SM_STAMP_AT(slot) do { } while (0)
#define SM_STAMP_RT(slot) do { } while (0)
#endif

constexpr int SM_KSW = 10;                    // k-steps (32 deep) of one wave per 640-deep chunk: two wave groups share the 20
constexpr int SM_SUBB = 32 * 64 * 2;          // one [32 rows][64 k] sub-tile of the panel
constexpr int SM_CHB = 10 * SM_SUBB;          // one 640-deep chunk of the panel: 40 KiB
constexpr int SM_NSLOT = 3;
constexpr int SM_SCRATCH = 20 * 1024;         // accumulator hand-over, LayerNorm statistics, GEGLU exchange

// NT: n-tiles (16 W rows) per slab = waves per wave group; LN: LayerNorm folded; GEGLU: value * gelu(gate)
template <int NT, bool LN, bool GEGLU>
__global__ __launch_bounds__(128 * NT) void smallm_kernel(NrGemmParams p_arg, int J, int C) {
  constexpr int NW = 2 * NT;                  // waves
  constexpr int PP = 40 / NW;                 // 1-KiB LDS-DMA pieces of a panel chunk per wave (40 per chunk)
  static_assert(40 % NW == 0, "whole pieces per wave");
  static_assert(!GEGLU || NT % 2 == 0, "value / gate n-tiles in pairs");
  const NrGemmParams p = nr_pin_params(p_arg);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // SM_NSLOT chunks, then SM_SCRATCH bytes

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  SM_STAMP_AT(0);
  SM_STAMP_RT(6);
  const int kq = wave >= NT ? 1 : 0;          // which half of every chunk's k-steps
  const int nt = wave - kq * NT;              // which n-tile of the slab
  const int fr = lane & 15, fg = lane >> 4;
  const int lr = lane >> 3, lp = lane & 7;
  const int ntm = (p.M + 31) >> 5;
  int bid;
  {   // XCD-aware remap (bijective): consecutive ids on one XCD, so the row tiles that share a column group's weights share an L2
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7, local = orig >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
  }
  // (an 8 row-tile x 4 column-group block per XCD instead of 16 x 2 -- less panel, more weights fetched per XCD -- was worth 1-2 % at K >= 5120 only)
  const int g = bid / ntm;                    // column group: slabs g J .. g J + J - 1
  const int m0 = (bid - g * ntm) * 32;
  const int T = J * C;                        // steps: (slab, chunk), chunk fastest
  const bool streaming = C > SM_NSLOT;        // else the whole panel stays resident in its C slots
  const int KB = p.K >> 5;

  // ---- weights: k-step ks of n-tile Tn is the KiB at ((Tn KB + ks) 64 + lane) 8 elements ----
  const bf16* wlane = p.w_fm + ((size_t)(g * J * NT + nt) * KB + kq * SM_KSW) * 512 + lane * 8;
  const size_t wslab = (size_t)NT * KB * 512;                         // elements from one slab's n-tile to the next slab's
  bf16x8 wa[SM_KSW], wb[SM_KSW];
  auto w_of = [&](int t) {                    // this wave's first KiB of step t = (slab t / C, chunk t % C)
    const int j = t / C, c = t - j * C;
    return wlane + (size_t)j * wslab + (size_t)c * (20 * 512);
  };
  auto issue_w = [&](bf16x8 (&w)[SM_KSW], int t) {
    const bf16* q = w_of(t);
#pragma unroll
    for (int ks = 0; ks < SM_KSW; ++ks) w[ks] = *(const bf16x8*)(q + ks * 512);
  };

  // ---- panel: piece q of a chunk = rows 8 (q & 3) .. + 7 of sub-tile q >> 2, lane (lr, lp) fetches the 16-byte chunk lp ^ lr of row lr ----
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(lptr_t)smem);
  auto issue_panel = [&](int t) {
    const int c = t % C;
    const int slot = streaming ? t % SM_NSLOT : c;
#pragma unroll
    for (int i = 0; i < PP; ++i) {
      const int q = wave * PP + i;
      const int am = m0 + 8 * (q & 3) + lr;
      const int k0 = c * 640 + (q >> 2) * 64 + ((lp ^ lr) << 3);
      // two-source operand [a0 | a1] (skip concat as a 1x1 GEMM, the [t | g] operand of the folded FeedForward): whole chunks from either
      const bf16* row = k0 < p.c0 ? p.a0 + (size_t)am * p.lda0 + k0 : p.a1 + (size_t)am * p.lda1 + (k0 - p.c0);
      const bf16* src = am < p.M ? row : (const bf16*)smallm_zero16;
      glds16_asm(src, lds0 + (unsigned)(slot * SM_CHB + q * 1024));
    }
  };

  f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  float ln_s1 = 0.f, ln_s2 = 0.f;               // LN: this thread's part of row (tid & 31)'s sums over the whole K
  float rs[2] = {1.f, 1.f}, mu[2] = {0.f, 0.f};

  // scratch behind the ring
  float* fsm = reinterpret_cast<float*>(smem + SM_NSLOT * SM_CHB);
  f32x4* red = reinterpret_cast<f32x4*>(fsm);                        // [NT][2 mt][64 lanes] f32x4 <= 10 KiB
  float* st = fsm + 2560;                                            // [NW * 2 parts][32 rows][2] <= 5 KiB
  f32x4* ex = reinterpret_cast<f32x4*>(fsm + 3840);                  // [NT / 2 pairs][2 mt][64 lanes] f32x4 <= 4 KiB

  // fragment of rows (fr, fr + 16), k-step kk of the chunk: sub-tile kk >> 1, 16-byte chunk 4 (kk & 1) + fg, XOR-swizzled with row & 7
  const int xo0 = (fg ^ (fr & 7)) << 4, xo1 = ((4 + fg) ^ (fr & 7)) << 4;
  const unsigned char* xlane = smem + fr * 128 + kq * (5 * SM_SUBB);

  // ---- prologue: everything the first two steps need is requested right after kernel entry ----
  issue_w(wa, 0);
  issue_panel(0);
  // (a barrier here, so that every wave's chunk-0 requests queue before anybody's chunk-1 requests, moves 1 k cycles from the first wait into the
  // issue phase and nothing else: profiles/r05_smallm_timeline.txt)
  if (T > 1) { issue_panel(1); issue_w(wb, 1); }
  SM_STAMP_AT(1);
  bool pnext = T > 1;                         // the panel chunk of step t + 1 was issued after step t's weights

  // Two register banks, two steps of weights in flight: bank t & 1 holds step t's weights and is refilled with step t + 2's, register by
  // register, right behind the MFMAs that read it (one step ahead was not enough: ~20 B/clk per CU, the tiled igemm's rate)
  auto step = [&](bf16x8 (&wcur)[SM_KSW], int t) __attribute__((always_inline)) {
    const bool has_next = t + 1 < T;
    if (t == 0) SM_STAMP_AT(2);
    // this step's weights and panel chunk are older than: [the next panel chunk] + [the next step's weights]
    if (has_next) { if (pnext) wait_vmcnt_c<SM_KSW + PP>(); else wait_vmcnt_c<SM_KSW>(); }
    else wait_vmcnt_c<0>();
    SM_STAMP_AT(8 + 3 * t);
    __builtin_amdgcn_s_barrier();             // every wave's pieces landed; every wave has left step t - 1 (its slot may be refilled)
    if (t == 0) SM_STAMP_AT(3);
    SM_STAMP_AT(9 + 3 * t);
    pnext = (t + 2 < T) && (streaming || t + 2 < C);
    if (pnext) issue_panel(t + 2);
    const int j = t / C, c = t - j * C;
    const int slot = streaming ? t % SM_NSLOT : c;
    const bool slab_end = c == C - 1;
    const int nq = ((g * J + j) * NT + nt) * 16 + 4 * fg;            // this lane's 4 W rows / output columns
    f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f}, cv = f32x4{0.f, 0.f, 0.f, 0.f}, rvv[2];
    bf16x4 rr[2];
    if (slab_end && kq == 0) {                // epilogue operands, fetched behind this chunk's MFMAs
      if (p.bias) bv = *(const f32x4*)(p.bias + nq);
      if constexpr (LN) cv = *(const f32x4*)(p.ln_c + nq);
      if constexpr (!GEGLU && !LN) {          // (LayerNorm-folded: fetched in the epilogue, the 168-register budget of 10 waves is full)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const int m = m0 + 16 * mt + fr;
          const int mc = m < p.M ? m : p.M - 1;
          if (p.rowvec) rvv[mt] = *(const f32x4*)(p.rowvec + rowvec_row(p, mc) + nq);
          if (p.res) rr[mt] = *(const bf16x4*)(p.res + (size_t)mc * p.ldr + nq);
        }
      }
    }
    if constexpr (LN) {
      if (j == 0) {   // first slab: row statistics straight from the chunk that just landed; thread (row tid & 31, part tid >> 5) reads 80 / NPART chunks
        constexpr int NPART = NW * 2;         // threads per row
        const int r = tid & 31, q0 = tid >> 5;
        const unsigned char* base = smem + slot * SM_CHB + r * 128;
        const bf16x2 one2 = {(bf16)1.0f, (bf16)1.0f};
#pragma unroll
        for (int i = 0; i < 80 / NPART; ++i) {
          const int cc = q0 + i * NPART;
          const bf16x8 v = *(const bf16x8*)(base + (cc >> 3) * SM_SUBB + (((cc & 7) ^ (r & 7)) << 4));
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const bf16x2 pr = {v[2 * e], v[2 * e + 1]};
            ln_s1 = __builtin_amdgcn_fdot2_f32_bf16(pr, one2, ln_s1, false);
            ln_s2 = __builtin_amdgcn_fdot2_f32_bf16(pr, pr, ln_s2, false);
          }
        }
      }
    }
    const unsigned char* xb = xlane + slot * SM_CHB;
    bf16x8 xa[2], xc2[2];
    auto read_x = [&](bf16x8 (&x)[2], int ks) {
      const unsigned char* q = xb + (ks >> 1) * SM_SUBB + ((ks & 1) ? xo1 : xo0);
      x[0] = *(const bf16x8*)q;
      x[1] = *(const bf16x8*)(q + 16 * 128);
    };
    read_x(xa, 0);
    const bool refill = t + 2 < T;
    const bf16* wref = w_of(refill ? t + 2 : t);
#pragma unroll
    for (int ks = 0; ks < SM_KSW; ++ks) {
      bf16x8 (&xc)[2] = (ks & 1) ? xc2 : xa;
      bf16x8 (&xn)[2] = (ks & 1) ? xa : xc2;
#ifdef SM_ABLATE_X        // timing experiment (WRONG results): a quarter of the LDS fragment reads
      if (ks + 1 < SM_KSW && ((ks + 1) & 3) == 0) read_x(xn, ks + 1); else { xn[0] = xc[0]; xn[1] = xc[1]; }
#else
      if (ks + 1 < SM_KSW) read_x(xn, ks + 1);
#endif
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wcur[ks], xc[0], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wcur[ks], xc[1], acc[1], 0, 0, 0);
      if (refill) wcur[ks] = *(const bf16x8*)(wref + ks * 512);
    }
    SM_STAMP_AT(10 + 3 * t);
    if (!slab_end) return;
    if (t == T - 1) SM_STAMP_AT(4);

    // ---- slab epilogue: the second wave group hands its accumulators over (fixed order), lane holds out[m0 + 16 mt + fr][nq + e] = acc[mt][e] ----
    if (kq == 1) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) red[(nt * 2 + mt) * 64 + lane] = acc[mt];
    }
    if constexpr (LN) {
      if (j == 0) { st[tid * 2] = ln_s1; st[tid * 2 + 1] = ln_s2; }      // [part][32 rows][2]
    }
    __syncthreads();
    if (kq == 0) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) acc[mt] += red[(nt * 2 + mt) * 64 + lane];
    }
    if constexpr (LN) {
      if (j == 0) {
        const float inv = 1.0f / (float)p.K;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          float a = 0.f, b = 0.f;
#pragma unroll 2                              // (all 4 NW reads at once cost 80 registers on top of the two weight banks)
          for (int q = 0; q < NW * 2; ++q) { a += st[(q * 32 + 16 * mt + fr) * 2]; b += st[(q * 32 + 16 * mt + fr) * 2 + 1]; }
          mu[mt] = a * inv;
          rs[mt] = rsqrtf(fmaxf(b * inv - mu[mt] * mu[mt], 0.f) + p.ln_eps);
        }
      }
    }
    if constexpr (GEGLU) {
      // W rows 32 i .. 32 i + 15 are the values of output columns 16 i .., rows 32 i + 16 .. their gates: even n-tiles hold values, odd gates;
      // the gate wave hands gelu(g) to its value neighbour through LDS
      f32x4 v[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        v[mt] = acc[mt];
        if constexpr (LN) v[mt] = (v[mt] - cv * mu[mt]) * rs[mt];
        v[mt] += bv;
      }
      if (kq == 0 && (nt & 1)) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          f32x4 gl;
#pragma unroll
          for (int e = 0; e < 4; ++e) gl[e] = gelu_erf_fast(v[mt][e]);
          ex[((nt >> 1) * 2 + mt) * 64 + lane] = gl;
        }
      }
      __syncthreads();
      if (kq == 0 && !(nt & 1)) {
        const int oc = ((((g * J + j) * NT + nt) * 16) >> 1) + 4 * fg;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const int m = m0 + 16 * mt + fr;
          if (m >= p.M) continue;
          const f32x4 gl = ex[((nt >> 1) * 2 + mt) * 64 + lane];
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (bf16)(v[mt][e] * gl[e]);
          *(bf16x4*)(p.out + (size_t)m * p.ldo + oc) = o;
        }
      }
    } else if (kq == 0) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int m = m0 + 16 * mt + fr;
        if (m >= p.M) continue;
        f32x4 v = acc[mt];
        if constexpr (LN) {
          v = (v - cv * mu[mt]) * rs[mt];
          if (p.rowvec) rvv[mt] = *(const f32x4*)(p.rowvec + rowvec_row(p, m) + nq);
          if (p.res) rr[mt] = *(const bf16x4*)(p.res + (size_t)m * p.ldr + nq);
        }
        v += bv;
        if (p.rowvec) v += rvv[mt];
        v *= p.out_scale;
        if (p.act == 1) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = quick_gelu_f(v[e]);
        }
        if (p.res) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += (float)rr[mt][e];
        }
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16)v[e];
        *(bf16x4*)(p.out + (size_t)m * p.ldo + nq) = o;
      }
    }
    acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
    acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
  };

  for (int t = 0; t < T; t += 2) {
    step(wa, t);
    if (t + 1 < T) step(wb, t + 1);
  }
#ifdef NR_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  SM_STAMP_AT(5);
  SM_STAMP_RT(7);
}

// row-major [N][K] -> fragment-major [N/16][K/32][64][8]: one thread per 16-byte chunk of the destination
__global__ __launch_bounds__(256) void smallm_w_pack_kernel(const bf16* __restrict__ w, bf16* __restrict__ out, int N, int K) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)N * K / 8) return;
  const int lane = (int)(idx & 63);
  const size_t blk = idx >> 6;
  const int kb = K >> 5;
  const int Tn = (int)(blk / kb), ks = (int)(blk - (size_t)Tn * kb);
  *(bf16x8*)(out + idx * 8) = *(const bf16x8*)(w + (size_t)(16 * Tn + (lane & 15)) * K + 32 * ks + 8 * (lane >> 4));
}

struct SmallmPlan { int nt, G, J, C; };

// shapes this kernel serves and how: nt n-tiles per slab, G column groups x J slabs each, C chunks of 640 along K
// decided: the caller already chose this kernel (w_fm set at plan time): no environment is read, only the shape rules apply, so a launch can
// never disagree with the plan that sized its workspace
bool smallm_plan(const NrGemmParams& p, SmallmPlan* out, bool decided) {
  int mode = 2;
  if (!decided) {
    if (getenv("NR_IGEMM_FORCE")) return false;                                       // plan sweeps of the tiled igemm (tools/gemm_sweep.py)
    const char* me = getenv("NR_SMALLM");                                             // read when the CHOICE is made (engine: plan time; nr_op_* hooks: per call)
    mode = me ? atoi(me) : 1;                                                         // 0 off, 1 M <= 512 (default), 2 every eligible launch
    if (!mode) return false;
  }
  if (p.ksize != 1 || p.stride != 1 || p.ups || p.out_f32 || p.tap_inner) return false;
  if (p.a1 ? (p.c0 % 640 != 0 || p.c1 % 640 != 0 || p.lda1 % 8 != 0 || p.ln_c) : p.c1 != 0) return false;      // second source: whole 640-deep chunks
  if (p.K != p.c0 + p.c1 || p.K % 640 != 0 || p.N % 16 != 0 || p.M < 1) return false;
  if (p.lda0 % 8 != 0 || p.ldo % 4 != 0 || (p.res && p.ldr % 4 != 0)) return false;
  if (p.rowvec && (p.rowvec_div <= 0 || p.rowvec_ld % 4 != 0)) return false;
  if (p.geglu && (p.rowvec || p.res || p.act || p.out_scale != 1.0f)) return false;
  const int Mp = (p.plan_m > 0 && p.plan_m < p.M) ? p.plan_m : p.M;                   // NR_DETERMINISTIC_BATCH: the choice is made per clip
  if (mode == 1 && Mp > 512) return false;
  const int C = p.K / 640;
  if (C > 16) return false;
  const int ntm = (Mp + 31) / 32, tiles = p.N / 16;      // planned on one clip's rows; the launch covers all of M with the same (nt, G, J)
  // the most workgroups that still fit ONE round of the chip (256 CUs, one workgroup each); ties: the wider slab
  SmallmPlan best{0, 0, 0, C};
  int best_wg = 0;
  for (int nt : {5, 4}) {
    if (tiles % nt != 0 || (p.geglu && nt % 2 != 0)) continue;
    const int slabs = tiles / nt;
    for (int G = 1; G <= slabs; ++G) {
      if (slabs % G != 0) continue;
      const int wg = ntm * G;
      if (wg > 256 && G > 1) break;
      if (wg > best_wg) { best_wg = wg; best = SmallmPlan{nt, G, slabs / G, C}; }
    }
  }
  if (!best.nt) return false;
  if (best.J * C > 64) return false;                                                  // a workgroup's serial depth: leave huge N x K to the tiled igemm
  // Several slabs per workgroup (N >= 1920 at M = 512) lose to the tiled igemm: every CU then streams its whole column group's weights
  // (32-row tiles: 16 x the weight bytes over the chip), 680 KB per CU at N = 3840 against the igemm's 490 KB with 128 x 64 tiles
  // (profiles/r05_smallm_ab.txt: 0.82-1.02x); one slab per workgroup wins 1.1-1.66x.  NR_SMALLM=2 keeps them (tests, the A/B tool).
  if (mode == 1 && best.J > 1) return false;
  if (out) *out = best;
  return true;
}

typedef void (*smallm_kern_t)(NrGemmParams, int, int);
template <int NT> smallm_kern_t smallm_pick(bool ln, bool geglu) {
  if constexpr (NT % 2 == 0) {
    return ln ? (geglu ? smallm_kernel<NT, true, true> : smallm_kernel<NT, true, false>)
              : (geglu ? smallm_kernel<NT, false, true> : smallm_kernel<NT, false, false>);
  } else {
    return ln ? smallm_kernel<NT, true, false> : smallm_kernel<NT, false, false>;
  }
}

}  // namespace

// The CHOICE, made once per launch description: 1 when this shape should run on the panel-resident kernel.  The caller then packs fragment-major
// weights and sets NrGemmParams::w_fm, and nr_launch_igemm / nr_igemm_workspace_bytes follow w_fm alone (no environment read at launch time)
extern "C" int nr_smallm_eligible(const NrGemmParams* pp) { return smallm_plan(*pp, nullptr, false) ? 1 : 0; }

extern "C" int nr_launch_smallm(const NrGemmParams* pp, hipStream_t stream) {
  const NrGemmParams& p = *pp;
  SmallmPlan pl;
  if (!p.w_fm || !smallm_plan(p, &pl, true)) return 1;
  const bool ln = p.ln_c != nullptr, gg = p.geglu != 0;
  smallm_kern_t k = pl.nt == 5 ? smallm_pick<5>(ln, gg) : smallm_pick<4>(ln, gg);
  const size_t shm = (size_t)SM_NSLOT * SM_CHB + SM_SCRATCH;           // 140 KiB: one workgroup per CU
  static unsigned long long attr_done[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // per instantiation: one bit per device ordinal
  {
    int dev = 0;
    (void)hipGetDevice(&dev);
    unsigned long long& mask = attr_done[(pl.nt == 5 ? 0 : 4) + (ln ? 2 : 0) + (gg ? 1 : 0)];
    if (!(mask & (1ull << (dev & 63)))) {
      if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess) return 2;
      mask |= 1ull << (dev & 63);
    }
  }
  const unsigned grid = (unsigned)(((p.M + 31) / 32) * pl.G);
  hipLaunchKernelGGL(k, dim3(grid), dim3(128 * pl.nt), shm, stream, p, pl.J, pl.C);
  return 0;
}

// w [N][K] row-major -> out in the fragment-major order NrGemmParams::w_fm documents (N % 16 == 0, K % 32 == 0)
extern "C" int nr_launch_smallm_w_pack(const void* w, void* out, int N, int K, hipStream_t stream) {
  if (N % 16 != 0 || K % 32 != 0) return 1;
  const size_t chunks = (size_t)N * K / 8;
  hipLaunchKernelGGL(smallm_w_pack_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, stream, (const bf16*)w, (bf16*)out, N, K);
  return 0;
}

#ifdef NR_STAMP
extern "C" int nr_smallm_stamp_read(void* dst, size_t bytes, int clear) {
  const size_t n = bytes < sizeof(smallm_stamp_buf) ? bytes : sizeof(smallm_stamp_buf);
  int rc = 0;
  if (dst) rc = (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(smallm_stamp_buf), n, 0, hipMemcpyDeviceToHost);
  if (clear) { void* d = nullptr; (void)hipGetSymbolAddress(&d, HIP_SYMBOL(smallm_stamp_buf)); (void)hipMemset(d, 0, sizeof(smallm_stamp_buf)); }
  return rc;
}
#endif
