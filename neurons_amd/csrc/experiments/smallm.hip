// Panel-resident GEMM for the small-M Linears (M <= 512: the 4x4 level of the AnimateDiff U-Net, the depth-10 16x16 level of the sgm
// keyframe UNetModel with CFG batch 2): out = epilogue( [LayerNorm](A) . W^T ), K = C or a multiple of it.  Reference operators: the
// nn.Linear layers of BasicTransformerBlock (animatediff/models/attention.py:256-300; generative_models/sgm/modules/attention.py:
// 551-572 BasicTransformerBlock, :340-400 CrossAttention to_q/to_k/to_v/to_out, :60-100 FeedForward / GEGLU), proj_in / proj_out
// (:690-742) and the temporal transformer Linears (motion_module.py:134-158,210-222).
//
// STATUS (round 3): REJECTED, not part of the product library (built only by `make -C neurons_amd/csrc experiments`).  Results equal the
// tiled igemm's (bit-equal without LayerNorm fold at KSPLIT 1), but it is slower: 14.7 vs 11.2 us on M = 512, N = K = 1280 inside a
// replayed graph with HBM-cold weights (profiles/r03_smallm_ab.txt), 19.5 vs 12.4 ms per Euler step on the sgm keyframe path and 19.07 vs
// 18.41 ms per DDIM step on the video path (same box, profiles/r03_smallm_insitu_ab.log).
//
// The idea.  In-kernel stamps (tools/igemm_timeline.py, profiles/r03_igemm_timeline_smallm.txt) of the tiled igemm on M = 512,
// N = K = 1280: 4,400 cycles from kernel entry to the first DMA, then 20 k-tiles x 670 cycles (3 LDS-DMA pieces, 6 fragment reads,
// 4 MFMAs and one barrier per tile and wave; the same 670 with the weights L2-warm or streamed from HBM), 2,100 cycles of epilogue:
// 20 k cycles per workgroup for 1,280 cycles of MFMA work.  So: no k-loop of staged tiles at all --
//   * a workgroup owns 32 rows x 64 W rows.  The 32 x K activation panel (K <= 1280 per pass: 80 KiB) goes to LDS in ONE burst of
//     LDS-DMA, and every wave loads ITS 16 x K weight slab straight from global memory into REGISTERS as MFMA A-operand fragments
//     (lane (fr, fg) holds W[n + fr][32 ks + 8 fg .. + 7]): the whole working set is requested right after kernel entry;
//   * one counted s_waitcnt + one barrier, then 2 K / 64 MFMAs per wave with two ds_read_b128 per k-step between them;
//   * K > 1280: passes of 1280 over the same accumulators (weights of the next pass re-issued into each register behind its MFMAs);
//   * KSPLIT wave groups share the k-steps of a pass (accumulators combined through LDS in fixed order); LayerNorm fold, GEGLU
//     (value / gate rows on neighbouring waves), bias / row vector / scale / quick_gelu / residual epilogue in fragment layout.
// What the stamps of THIS kernel showed: the MFMA loop is 3,500 cycles as planned, but ISSUING the loads takes 12,000: 20 LDS-DMA
// pieces in ~3,600 cycles and 40 global_load_dwordx4 in ~8,400 (~210 cycles per 1-KiB wave instruction), the same with 4, 8 or 16 waves
// per workgroup and with 80 or 320 workgroups on the chip: a CU takes ~20 B/clk of fragment-shaped loads (64-byte row segments)
// and ~40 B/clk of LDS-DMA (128-byte segments), far below the 64 B/clk L1 figure, whatever the number of waves.  The tiled igemm moves
// the same bytes (245 KB per workgroup) at the same ~18-20 B/clk; it is bound by that fill rate too, not by its k-loop overhead, and
// several co-resident workgroups hide its latencies better than one big burst does.  A small-M GEMM on this chip costs
// bytes-per-CU / ~20-40 B/clk (a 2,560-output share per CU = 250-300 KB: 6-12 k cycles), which is why hipBLASLt is no faster either.
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
__device__ unsigned long long smallm_stamp_buf[512][8];
#define SM_STAMP_AT(slot) do { if (threadIdx.x == 0 && blockIdx.x < 512) smallm_stamp_buf[blockIdx.x][(slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SM_STAMP_AT(slot) do { } while (0)
#endif

// KS: 32-deep k-steps per pass (40: K multiple of 1280; 20: K multiple of 640); KSPLIT: wave groups that share the k-steps of a pass
// (4 KSPLIT waves per workgroup: a wave's vector-memory issue rate, ~200 cycles per 1-KiB instruction, is what bounds this kernel, so the
// loads are spread over as many waves as the CU holds); LN: LayerNorm folded (single pass); GEGLU: value * gelu(gate)
template <int KS, int KSPLIT, bool LN, bool GEGLU>
__global__ __launch_bounds__(256 * KSPLIT) void smallm_kernel(NrGemmParams p, int npass) {
  constexpr int KT = KS / 2;                  // [32][64] sub-tiles of the panel
  constexpr int KSW = KS / KSPLIT;            // k-steps of one wave per pass
  constexpr int KTW = KT / KSPLIT;            // LDS-DMA pieces of one wave per pass
  constexpr int SUBB = 32 * 64 * 2;           // bytes of one sub-tile
  static_assert(KS % (2 * KSPLIT) == 0, "whole sub-tiles per wave group");
  extern __shared__ __attribute__((aligned(16))) bf16 smem[];   // KT sub-tiles; reused by the epilogue exchanges (needs >= 32 KiB)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  SM_STAMP_AT(0);
  const int wq = wave & 3;                    // which 16 W rows
  const int kq = wave >> 2;                   // which part of the k-steps
  const int fr = lane & 15, fg = lane >> 4;
  const int lr = lane >> 3, lp = lane & 7;
  const int ntm = (p.M + 31) >> 5;
  int bid;
  {   // XCD-aware remap (bijective): consecutive ids on one XCD, so the row tiles that share a weight slab share an L2
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7, local = orig >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
  }
  const int nslab = bid / ntm;
  const int m0 = (bid - nslab * ntm) * 32;
  const int nw = nslab * 64 + 16 * wq;        // first of this wave's 16 W rows

  // ---- sources: wave (wq, kq) stages rows 8 wq .. 8 wq + 7 of the sub-tiles of k-part kq; lane (lr, lp) fetches the 16-byte chunk lp ^ lr ----
  const int am = m0 + 8 * wq + lr;
  const bool a_ok = am < p.M;
  const bf16* ap = a_ok ? p.a0 + (size_t)am * p.lda0 + kq * (KTW * 64) + ((lp ^ lr) << 3) : (const bf16*)smallm_zero16;
  const int ainc = a_ok ? 64 : 0, ainc_pass = a_ok ? (KT - KTW) * 64 : 0;
  const bf16* wrow = p.w + (size_t)(nw + fr) * p.K + kq * (KSW * 32) + 8 * fg;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(lptr_t)smem) + (unsigned)(kq * KTW * SUBB + wq * 8 * 64 * 2);

  // fragment of rows (fr, fr + 16), k-step ks of the pass: sub-tile ks >> 1, 16-byte chunk 4 (ks & 1) + fg, XOR-swizzled with row & 7
  const char* xbase = reinterpret_cast<const char*>(smem) + kq * (KTW * SUBB) + fr * 128;
  const int xo0 = (fg ^ (fr & 7)) << 4, xo1 = ((4 + fg) ^ (fr & 7)) << 4;
  auto read_x = [&](bf16x8 (&x)[2], int ks) {     // ks: k-step within this wave's part
    const char* q = xbase + (ks >> 1) * SUBB + ((ks & 1) ? xo1 : xo0);
    x[0] = *(const bf16x8*)q;
    x[1] = *(const bf16x8*)(q + 16 * 128);
  };

  f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  bf16x8 wreg[KSW];
  float ln_s1[2] = {0.f, 0.f}, ln_s2[2] = {0.f, 0.f};

  // epilogue operands (waves kq == 0), fetched while the last pass computes
  const int nq = nw + 4 * fg;                 // this lane's 4 W rows / output columns
  f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f}, cv = f32x4{0.f, 0.f, 0.f, 0.f}, rvv[2];
  bf16x4 rr[2];

  for (int s = 0; s < npass; ++s) {
    if (s > 0) __builtin_amdgcn_s_barrier();          // every wave has consumed the previous panel
#pragma unroll
    for (int j = 0; j < KTW; ++j) { glds16_asm(ap, lds0 + (unsigned)(j * SUBB)); ap += ainc; }
    ap += ainc_pass;
    if (s == 0) SM_STAMP_AT(1);
    if (s == 0) {
#pragma unroll
      for (int ks = 0; ks < KSW; ++ks) wreg[ks] = *(const bf16x8*)(wrow + 32 * ks);
      SM_STAMP_AT(2);
      wait_vmcnt_c<KSW>();                             // the panel pieces are older than the KSW weight loads
    } else {
      wait_vmcnt_c<0>();                               // the pieces are the youngest; this pass's weights were issued a pass ago
    }
    __builtin_amdgcn_s_barrier();
    if (s == 0) SM_STAMP_AT(3);
    const bool last = s + 1 == npass;
    if (last && kq == 0) {
      if (p.bias) bv = *(const f32x4*)(p.bias + nq);
      if constexpr (LN) cv = *(const f32x4*)(p.ln_c + nq);
      if constexpr (!GEGLU) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const int m = m0 + 16 * mt + fr;
          const int mc = m < p.M ? m : p.M - 1;
          if (p.rowvec) rvv[mt] = *(const f32x4*)(p.rowvec + rowvec_row(p, mc) + nq);
          if (p.res) rr[mt] = *(const bf16x4*)(p.res + (size_t)mc * p.ldr + nq);
        }
      }
    }
    const bf16* wnext = wrow + (size_t)(s + 1) * (KS * 32);
    bf16x8 xa[2], xb[2];
    read_x(xa, 0);
#pragma unroll
    for (int ks = 0; ks < KSW; ++ks) {
      bf16x8 (&xc)[2] = (ks & 1) ? xb : xa;
      bf16x8 (&xn)[2] = (ks & 1) ? xa : xb;
      if (ks + 1 < KSW) read_x(xn, ks + 1);
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[ks], xc[0], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[ks], xc[1], acc[1], 0, 0, 0);
      if (!last) wreg[ks] = *(const bf16x8*)(wnext + 32 * ks);
      if constexpr (LN) {
        if ((ks & 3) == wq) {                          // this wave's quarter of its k-part of the row statistics
          const bf16x2 one2 = {(bf16)1.0f, (bf16)1.0f};
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const bf16x2 pr = {xc[mt][2 * e], xc[mt][2 * e + 1]};
              ln_s1[mt] = __builtin_amdgcn_fdot2_f32_bf16(pr, one2, ln_s1[mt], false);
              ln_s2[mt] = __builtin_amdgcn_fdot2_f32_bf16(pr, pr, ln_s2[mt], false);
            }
        }
      }
    }
  }
  SM_STAMP_AT(4);

  // ---- combine the k-parts (fixed order), then the epilogue on the waves kq == 0: lane holds out[m0 + 16 mt + fr][nq + r] = acc[mt][r] ----
  float* fsm = reinterpret_cast<float*>(smem);
  f32x4* red = reinterpret_cast<f32x4*>(fsm);                     // [4 wq][3 kq][2 mt][64 lanes] f32x4 = 24 KiB
  float* st = fsm + 6144;                                          // [16 waves][32 rows][2] = 4 KiB
  f32x4* ex = reinterpret_cast<f32x4*>(fsm + 7168) + (wq >> 1) * 128;   // [2 pairs][2 mt][64 lanes] f32x4 = 4 KiB
  float rs[2] = {1.f, 1.f}, mu[2] = {0.f, 0.f};
  if constexpr (LN) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      ln_s1[mt] += __shfl_xor(ln_s1[mt], 16, 64); ln_s2[mt] += __shfl_xor(ln_s2[mt], 16, 64);
      ln_s1[mt] += __shfl_xor(ln_s1[mt], 32, 64); ln_s2[mt] += __shfl_xor(ln_s2[mt], 32, 64);
    }
  }
  if constexpr (KSPLIT > 1 || LN || GEGLU) __syncthreads();       // the panel is free
  if constexpr (KSPLIT > 1) {
    if (kq > 0) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) red[((wq * 3 + kq - 1) * 2 + mt) * 64 + lane] = acc[mt];
    }
  }
  if constexpr (LN) {
    if (fg == 0) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) { st[(wave * 32 + 16 * mt + fr) * 2] = ln_s1[mt]; st[(wave * 32 + 16 * mt + fr) * 2 + 1] = ln_s2[mt]; }
    }
  }
  if constexpr (KSPLIT > 1 || LN) __syncthreads();
  const bool owner = kq == 0;                                      // the waves that hold the combined accumulators and store
  if constexpr (KSPLIT > 1) {
    if (owner) {
#pragma unroll
      for (int k = 1; k < KSPLIT; ++k)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) acc[mt] += red[((wq * 3 + k - 1) * 2 + mt) * 64 + lane];
    }
  }
  if constexpr (LN) {
    const float inv = 1.0f / (float)p.K;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int w = 0; w < 4 * KSPLIT; ++w) { a += st[(w * 32 + 16 * mt + fr) * 2]; b += st[(w * 32 + 16 * mt + fr) * 2 + 1]; }
      mu[mt] = a * inv;
      rs[mt] = rsqrtf(fmaxf(b * inv - mu[mt] * mu[mt], 0.f) + p.ln_eps);
    }
  }
  if constexpr (GEGLU) {
    // W rows 32 j .. 32 j + 15 are the values of output columns 16 j .., rows 32 j + 16 .. their gates: even wq hold values, odd gates;
    // the gate wave hands gelu(g) to its value neighbour through LDS
    f32x4 v[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      v[mt] = acc[mt];
      if constexpr (LN) v[mt] = (v[mt] - cv * mu[mt]) * rs[mt];
      v[mt] += bv;
    }
    if (owner && (wq & 1)) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        f32x4 g;
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] = gelu_erf_fast(v[mt][e]);
        ex[mt * 64 + lane] = g;
      }
    }
    __syncthreads();
    if (owner && !(wq & 1)) {
      const int oc = ((nslab * 64 + 16 * wq) >> 1) + 4 * fg;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int m = m0 + 16 * mt + fr;
        if (m >= p.M) continue;
        const f32x4 g = ex[mt * 64 + lane];
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16)(v[mt][e] * g[e]);
        *(bf16x4*)(p.out + (size_t)m * p.ldo + oc) = o;
      }
    }
  } else if (owner) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int m = m0 + 16 * mt + fr;
      if (m >= p.M) continue;
      f32x4 v = acc[mt];
      if constexpr (LN) v = (v - cv * mu[mt]) * rs[mt];
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
#ifdef NR_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  SM_STAMP_AT(5);
}

}  // namespace

// shapes this kernel serves; ks_out: k-steps per pass (20 / 40), npass_out: passes over K
extern "C" int nr_smallm_plan(const NrGemmParams* pp, int* ks_out, int* npass_out) {
  const NrGemmParams& p = *pp;
  const int mode = getenv("NR_SMALLM") ? atoi(getenv("NR_SMALLM")) : 0;       // 0 off (default), 1 M <= 512, 2 every eligible launch
  if (!mode) return 0;
  if (p.ksize != 1 || p.a1 || p.c1 != 0 || p.stride != 1 || p.ups || p.out_f32 || p.tap_inner) return 0;
  if (p.K % 640 != 0 || p.N % 64 != 0 || p.M < 1) return 0;
  if (p.lda0 % 8 != 0 || p.ldo % 4 != 0 || (p.res && p.ldr % 4 != 0)) return 0;
  if (p.rowvec && (p.rowvec_div <= 0 || p.rowvec_ld % 4 != 0)) return 0;
  if (p.geglu && (p.rowvec || p.res || p.act || p.out_scale != 1.0f)) return 0;
  const int ks = p.K % 1280 == 0 ? 40 : 20;
  const int npass = p.K / (32 * ks);
  if (p.ln_c && npass != 1) return 0;
  if (mode == 1 && (p.M > 512 || npass > 8)) return 0;
  if (ks_out) *ks_out = ks;
  if (npass_out) *npass_out = npass;
  return 1;
}

namespace {
typedef void (*smallm_kern_t)(NrGemmParams, int);
template <int KS, int KSPLIT> smallm_kern_t smallm_pick(bool ln, bool geglu) {
  return ln ? (geglu ? smallm_kernel<KS, KSPLIT, true, true> : smallm_kernel<KS, KSPLIT, true, false>)
            : (geglu ? smallm_kernel<KS, KSPLIT, false, true> : smallm_kernel<KS, KSPLIT, false, false>);
}
}  // namespace

extern "C" int nr_launch_smallm(const NrGemmParams* pp, hipStream_t stream) {
  const NrGemmParams& p = *pp;
  int ks = 0, npass = 0;
  if (!nr_smallm_plan(pp, &ks, &npass)) return 1;
  int split = ks == 40 ? 4 : 2;                       // 10 k-steps per wave either way: 16 / 8 waves per workgroup
  if (const char* e = getenv("NR_SMALLM_KSPLIT")) { const int v = atoi(e); if (v == 1 || v == 2 || (v == 4 && ks == 40)) split = v; }
  const bool ln = p.ln_c != nullptr, gg = p.geglu != 0;
  smallm_kern_t k = nullptr;
  if (ks == 40) k = split == 4 ? smallm_pick<40, 4>(ln, gg) : (split == 2 ? smallm_pick<40, 2>(ln, gg) : smallm_pick<40, 1>(ln, gg));
  else k = split == 2 ? smallm_pick<20, 2>(ln, gg) : smallm_pick<20, 1>(ln, gg);
  const size_t shm = (size_t)(ks / 2) * 4096;         // 80 / 40 KiB (>= the 32 KiB the epilogue exchanges use)
  if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess) return 2;
  const unsigned grid = (unsigned)(((p.M + 31) / 32) * (p.N / 64));
  hipLaunchKernelGGL(k, dim3(grid), dim3(256 * split), shm, stream, p, npass);
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
