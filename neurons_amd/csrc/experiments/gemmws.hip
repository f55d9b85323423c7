// Producer / consumer ("wave-specialised") 128 x 128 implicit-GEMM tile: NCONS MFMA waves + NPROD waves that do nothing but issue the LDS-DMA.
// Same operator as gemm.hip for its tap-inner 3x3 convs (InflatedConv3d of the ResnetBlock3D: animatediff/models/resnet.py:10-18,
// 182-212) and single-source 1x1 convs / Linears without LayerNorm fold or GEGLU.
//
// STATUS (round 3, profiles/r03_gemmws_ab.txt): REJECTED, not part of the product library (`make -C neurons_amd/csrc experiments`,
// NR_IGEMM_WS=2, NR_IGEMM_WS_NCONS / _NPROD / _NS).  Bit-identical to gemm.hip (same summation order) and 0.6-0.85x its speed: 0.39x
// with one producer wave, 0.54x with two, 0.65-0.8x with four; eight consumers instead of four, three or four LDS stages, and issuing
// the DMA while the consumers read their fragments change nothing.
//
// The idea.  tools/ubench/dma_issue.hip: a CU accepts one 1-KiB LDS-DMA piece per ~18 cycles (55-74 B/clk) however many waves issue, and a
// wave that issues a piece is BLOCKED until the piece is accepted: 74 cycles per piece with 4 issuing waves on the CU, 145 with 8.  In
// gemm.hip every wave issues its 8-9 pieces per k-tile itself in front of its 32-40 MFMAs, and a k-tile pair takes 3,400 cycles where the
// matrix pipe needs 1,280 and the fill path 1,300 (profiles/r03_igemm_kloop_ablation.txt).  Here the queueing is moved off the MFMA waves:
//   * the consumer waves only read fragments and issue MFMAs; the producer waves own the source pointers (running 64-bit pointers advanced
//     by wave-uniform increments: +128 B per k-tile for W and for 1x1 A, the difference of the tap offsets for tap-inner 3x3 A, padding /
//     tail rows switched to a zero word by a per-row 9-bit tap mask);
//   * two raw barriers per k-tile: A(kt) "tile kt has landed" (the producers arrive after a counted s_waitcnt vmcnt) and B(kt) "every
//     consumer holds tile kt's fragments in registers"; NS LDS stages.
// What it showed: a LONE wave issues only one piece per ~80 cycles (address select + m0 + the instruction), so the CU's 18 cycles per piece
// need >= 4 waves issuing side by side -- which is what gemm.hip's eight compute waves per CU already are.  With four producers the
// iteration still takes ~1,400 cycles per 128 x 128 k-tile (one workgroup per CU: 12 waves x 92 VGPRs or 8 x 140), bound by the
// producers' own instruction stream (8 pieces x (~75 blocked + ~8 VALU) + the pointer advance) and the two barriers, not by the MFMAs.
// Two co-resident workgroups whose waves both issue and compute (gemm.hip) remain the better arrangement at these tile sizes.
// Epilogue (consumers only, fragment layout): bias, fp32 row vector, scale, quick_gelu, residual; deterministic split-K slabs for the
// reduce kernel of gemm.hip.
#include "common.h"
#include <cstdlib>

namespace {

__device__ __attribute__((aligned(16))) const unsigned int nrws_zero16[4] = {0u, 0u, 0u, 0u};
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void glds16_asm(const void* src, unsigned lds_wave_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(lds_wave_base) : "memory", "m0");
}
__device__ __forceinline__ size_t rowvec_row(const NrGemmParams& p, int m) {
  int r = m / p.rowvec_div;
  if (p.rowvec_mod) r %= p.rowvec_mod;
  return (size_t)r * p.rowvec_ld;
}
__device__ __forceinline__ int fdiv_small(int a, int d) {      // exact for 0 <= a < 2^22 (gemm.hip)
  int q = (int)((float)a * __builtin_amdgcn_rcpf((float)d));
  const int r = a - q * d;
  q += (r >= d ? 1 : 0) - (r < 0 ? 1 : 0);
  return q;
}

#ifdef NR_STAMP
// diagnostic build (make experiments STAMP=1, tools/ws_timeline.py): per workgroup, cycles summed over the k-loop of producer 0 and consumer 0,
// split at the two barriers: [role][0] waiting at A, [1] issue / fragment reads, [2] waiting at B, [3] landing wait / MFMAs
__device__ unsigned long long ws_stamp_buf[512][2][4];
#define WS_T() __builtin_amdgcn_s_memtime()
#else
#define WS_T() 0ull
#endif

// NPROD: LDS-DMA waves per workgroup (a lone wave issues one piece per ~80 cycles; the CU accepts one per ~18 from four); NS: LDS stages
// NCONS: MFMA waves (4: 64 x 64 outputs each, 8: 64 x 32)
template <int NCONS, int NPROD, int NS>
__global__ __launch_bounds__(64 * (NCONS + NPROD)) void igemm_ws_kernel(NrGemmParams p, int splitk, float* partial, int m_fast) {
  constexpr int BM = 128, BN = 128, BK = 64;
  constexpr int WNC = BN / (NCONS / 2);               // columns of a consumer wave's sub-tile
  constexpr int MT = 4, NT = WNC / 16;                // accumulator tiles of a consumer wave: 64 rows x WNC columns
  constexpr int NA = BM / 8, NB = BN / 8;             // LDS-DMA pieces of the A / W tile
  constexpr int NAP = NA / NPROD, NBP = NB / NPROD;   // pieces per producer wave and k-tile
  constexpr int PP = NAP + NBP;
  constexpr int TILE = (BM + BN) * BK;                // elements per LDS stage
  extern __shared__ __attribute__((aligned(16))) bf16 smem[];     // NS stages

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntn = (p.N + BN - 1) / BN;
  const int ntm = (p.M + BM - 1) / BM;
  int bid;
  {   // XCD-aware remap (bijective), as gemm.hip
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7, local = orig >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
  }
  const int slice = splitk > 1 ? fdiv_small(bid, ntn * ntm) : 0;
  bid -= slice * ntn * ntm;
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
  const int m0 = bm * BM, n0 = bn * BN;
  const int nk_total = p.K / BK;
  int kt_begin = 0, kt_end = nk_total;
  if (splitk > 1) { kt_begin = fdiv_small(nk_total * slice, splitk); kt_end = fdiv_small(nk_total * (slice + 1), splitk); }
  const int nk = kt_end - kt_begin;

  if (wave >= NCONS) {
    const int pq = wave - NCONS;                            // producer index: pieces j = pq + NPROD * jj
    // ================================================= producer =================================================
    const int lr = lane >> 3, lp = lane & 7;
    const int lchunk = (lp ^ lr) << 3;
    const bool ti = p.ksize == 3;
    const char* zsrc = (const char*)nrws_zero16;
    // k-tile kt (in the weight's K order): 1x1: channels 64 kt ..; tap-inner 3x3: chunk kt / 9, tap kt % 9
    int st_tap = 0, st_c = 0;
    if (ti) { const int q9 = fdiv_small(kt_begin, 9); st_tap = kt_begin - 9 * q9; st_c = q9 * BK; } else { st_c = kt_begin * BK; }
    auto tap_delta = [&](int t) -> long long {          // byte offset of tap t from the centre pixel
      const int ky = fdiv_small(t, 3), kx = t - 3 * ky;
      return ((long long)(ky - 1) * p.W + (kx - 1)) * p.lda0 * (long long)sizeof(bf16);
    };
    const char* ap[NAP];         // running pointer of row 8 j + lr: centre pixel + current tap + current chunk (valid memory or not)
    unsigned amask[NAP];         // bit t: tap t of this row lies inside the image (1x1: bit 0 = row < M)
    {
      const long long d0 = ti ? tap_delta(st_tap) : 0;
#pragma unroll
      for (int jj = 0; jj < NAP; ++jj) {
        const int j = pq + NPROD * jj;
        const int m = m0 + 8 * j + lr;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        unsigned msk = 0;
        long long centre;
        if (ti) {
          const int ohw = p.OH * p.OW;
          const int n = fdiv_small(mm, ohw);
          const int r = mm - n * ohw;
          const int oy = fdiv_small(r, p.OW), ox = r - oy * p.OW;
          centre = (((long long)n * p.H + oy) * p.W + ox) * p.lda0;
#pragma unroll
          for (int t = 0; t < 9; ++t) {
            const int iy = oy + t / 3 - 1, ix = ox + t % 3 - 1;
            if (ok && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) msk |= 1u << t;
          }
        } else {
          centre = (long long)mm * p.lda0;
          msk = ok ? 1u : 0u;
        }
        ap[jj] = reinterpret_cast<const char*>(p.a0 + centre + st_c + lchunk) + d0;
        amask[jj] = msk;
      }
    }
    const char* wp[NBP];
#pragma unroll
    for (int ii = 0; ii < NBP; ++ii) {
      const int i = pq + NPROD * ii;
      const int n = n0 + 8 * i + lr;
      wp[ii] = n < p.N ? reinterpret_cast<const char*>(p.w + (size_t)n * p.K + (size_t)kt_begin * BK + lchunk) : nullptr;
    }
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(lptr_t)smem);
    auto issue_tile = [&](int buf) {                    // the tile the running pointers stand on -> stage buf; then advance them
      const unsigned sa = lds_base + (unsigned)(buf * TILE * (int)sizeof(bf16));
      const unsigned sb = sa + (unsigned)(BM * BK * (int)sizeof(bf16));
      const unsigned tapbit = ti ? (unsigned)st_tap : 0u;
#pragma unroll
      for (int jj = 0; jj < NAP; ++jj) {
        const char* src = ((amask[jj] >> tapbit) & 1u) ? ap[jj] : zsrc;
        glds16_asm(src, sa + (unsigned)((pq + NPROD * jj) * 8 * BK * (int)sizeof(bf16)));
      }
#pragma unroll
      for (int ii = 0; ii < NBP; ++ii) {
        const char* src = wp[ii] ? wp[ii] : zsrc;
        glds16_asm(src, sb + (unsigned)((pq + NPROD * ii) * 8 * BK * (int)sizeof(bf16)));
      }
      // advance: W and 1x1 A by one k-tile; tap-inner A by the difference of the tap offsets (next chunk behind tap 8)
      long long da = BK * (long long)sizeof(bf16);
      if (ti) {
        const int nt = st_tap == 8 ? 0 : st_tap + 1;
        da = tap_delta(nt) - tap_delta(st_tap) + (st_tap == 8 ? BK * (long long)sizeof(bf16) : 0);
        st_tap = nt;
      }
#pragma unroll
      for (int jj = 0; jj < NAP; ++jj) ap[jj] += da;
#pragma unroll
      for (int ii = 0; ii < NBP; ++ii) if (wp[ii]) wp[ii] += BK * sizeof(bf16);
    };
    // Tiles 0 .. NS-2 go out at once.  In iteration kt the tile kt + NS - 1 is issued between A(kt) and B(kt), i.e. WHILE the consumers read
    // tile kt's fragments, into the stage of tile kt - 1 (released at B(kt - 1), which this wave has passed); the wait in front of
    // A(kt + 1) leaves the NS - 2 younger tiles in flight.
    constexpr int VMW2 = (NS - 2) * PP > 63 ? 63 : (NS - 2) * PP;
#pragma unroll
    for (int t = 0; t < NS - 1; ++t) if (t < nk) issue_tile(t);
    if (nk >= NS - 1) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(VMW2) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    static_assert(NS >= 3, "the single-barrier protocol refills the stage released one iteration earlier");
    int buf = NS - 1;
    unsigned long long acc_t[4] = {0, 0, 0, 0};
    unsigned long long t0 = WS_T();
    for (int kt = 0; kt < nk; ++kt) {
      // X(kt): tile kt has landed (this wave waited for it below) and every consumer has left tile kt - 1 (its fragment reads were
      // drained before its MFMAs), so the stage of tile kt - 1 is free for tile kt + NS - 1
      __builtin_amdgcn_s_barrier();
      const unsigned long long t1 = WS_T();
      const bool more = kt + NS - 1 < nk;
      if (more) issue_tile(buf);
      buf = buf + 1 == NS ? 0 : buf + 1;
      const unsigned long long t2 = WS_T();
      if (more) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(VMW2) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned long long t4 = WS_T();
      acc_t[0] += t1 - t0; acc_t[1] += t2 - t1; acc_t[3] += t4 - t2;
      t0 = t4;
    }
#ifdef NR_STAMP
    if (pq == 0 && lane == 0 && blockIdx.x < 512) for (int i = 0; i < 4; ++i) ws_stamp_buf[blockIdx.x][0][i] = acc_t[i];
#endif
    return;
  }

  // ================================================= consumers =================================================
  const int wm = wave & 1, wn = wave >> 1;
  const int fr = lane & 15, fg = lane >> 4;
  f32x4 acc[NT][MT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // fragment addresses inside a stage (rows fixed per lane; the XOR swizzle of gemm.hip)
  int offw[2][NT], offx[2][MT];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
    for (int i = 0; i < NT; ++i) { const int row = wn * WNC + i * 16 + fr; offw[ks][i] = BM * BK + row * BK + (((4 * ks + fg) ^ (row & 7)) << 3); }
#pragma unroll
    for (int j = 0; j < MT; ++j) { const int row = wm * 64 + j * 16 + fr; offx[ks][j] = row * BK + (((4 * ks + fg) ^ (row & 7)) << 3); }
  }
  int cbuf = 0;
  unsigned long long acc_t[4] = {0, 0, 0, 0};
  unsigned long long t0 = WS_T();
  for (int kt = 0; kt < nk; ++kt) {
    __builtin_amdgcn_s_barrier();                       // X(kt): tile kt has landed
    const unsigned long long t1 = WS_T();
    const bf16* st = smem + cbuf * TILE;
    cbuf = cbuf + 1 == NS ? 0 : cbuf + 1;
    bf16x8 wf[2][NT], xf[2][MT];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int i = 0; i < NT; ++i) wf[ks][i] = *(const bf16x8*)(st + offw[ks][i]);
#pragma unroll
      for (int j = 0; j < MT; ++j) xf[ks][j] = *(const bf16x8*)(st + offx[ks][j]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t2 = WS_T();
    const unsigned long long t3 = t2;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][i], xf[ks][j], acc[i][j], 0, 0, 0);
#ifdef NR_STAMP
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t4 = WS_T();
    acc_t[0] += t1 - t0; acc_t[1] += t2 - t1; acc_t[2] += t3 - t2; acc_t[3] += t4 - t3;
    t0 = t4;
#endif
  }
#ifdef NR_STAMP
  if (wave == 0 && lane == 0 && blockIdx.x < 512) for (int i = 0; i < 4; ++i) ws_stamp_buf[blockIdx.x][1][i] = acc_t[i];
#endif

  // ---- epilogue: lane holds out[m = .. + fr][n = .. + 4 fg + r] ----
  if (partial) {
    float* slab = partial + (size_t)slice * p.M * p.N;
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      const int m = m0 + wm * 64 + j * 16 + fr;
      if (m >= p.M) continue;
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const int n = n0 + wn * WNC + i * 16 + 4 * fg;
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
      const int n = n0 + wn * WNC + i * 16 + 4 * fg;
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

}  // namespace

// shapes this kernel serves; splitk_out: its K split
extern "C" int nr_igemm_ws_plan(const NrGemmParams* pp, int* splitk_out) {
  const NrGemmParams& p = *pp;
  const int mode = getenv("NR_IGEMM_WS") ? atoi(getenv("NR_IGEMM_WS")) : 0;     // 0 off, 1 heuristic, 2 every eligible launch
  if (!mode) return 0;
  if (p.ln_c || p.geglu || p.out_f32 || p.a1 || p.c1 != 0 || p.stride != 1 || p.ups || p.pad_tl0) return 0;
  if (!(p.ksize == 1 || (p.ksize == 3 && p.tap_inner))) return 0;
  if (p.K % 64 != 0 || p.c0 % 64 != 0 || p.K != p.ksize * p.ksize * p.c0 || p.N % 32 != 0 || p.lda0 % 8 != 0) return 0;
  const int nk = p.K / 64;
  if (mode == 1 && (p.M < 2048 || nk < 16)) return 0;
  const long long tiles = (long long)((p.M + 127) / 128) * ((p.N + 127) / 128);
  int sk = 1;
  if (tiles < 400) {                                           // two workgroups per CU: fill 512 slots with K slices of >= 12 k-tiles
    sk = (int)((512 + tiles - 1) / tiles);
    if (sk > nk / 12) sk = nk / 12;
    if (sk < 1) sk = 1;
    if (sk > 16) sk = 16;
  }
  if (const char* f = getenv("NR_IGEMM_WS_SPLITK")) sk = atoi(f) > 0 ? atoi(f) : sk;
  if (splitk_out) *splitk_out = sk;
  return 1;
}

extern "C" size_t nr_igemm_ws_workspace_bytes(const NrGemmParams* pp) {
  int sk;
  if (!nr_igemm_ws_plan(pp, &sk)) return 0;
  return sk > 1 ? (size_t)sk * pp->M * pp->N * sizeof(float) : 0;
}

// returns 0 on success; the caller runs the split-K reduce kernel of gemm.hip when *splitk_used > 1
extern "C" int nr_launch_igemm_ws(const NrGemmParams* pp, float* workspace, int m_fast, int* splitk_used, hipStream_t stream) {
  int sk;
  if (!nr_igemm_ws_plan(pp, &sk)) return 1;
  if (sk > 1 && !workspace) return 6;
  const NrGemmParams& p = *pp;
  const unsigned grid = (unsigned)(((p.M + 127) / 128) * ((p.N + 127) / 128) * sk);
  float* partial = sk > 1 ? workspace : nullptr;
  const int nprod = getenv("NR_IGEMM_WS_NPROD") ? atoi(getenv("NR_IGEMM_WS_NPROD")) : 2;
  const int ns = getenv("NR_IGEMM_WS_NS") ? atoi(getenv("NR_IGEMM_WS_NS")) : 3;
  const int ncons = getenv("NR_IGEMM_WS_NCONS") ? atoi(getenv("NR_IGEMM_WS_NCONS")) : 8;
  typedef void (*kern_t)(NrGemmParams, int, float*, int);
  kern_t k = nullptr;
  int np = 4, nc = 8;
  const int nsl = ns <= 3 ? 3 : 4;
  if (ncons == 4) {
    nc = 4;
    if (nprod <= 2) { np = 2; k = nsl == 3 ? igemm_ws_kernel<4, 2, 3> : igemm_ws_kernel<4, 2, 4>; }
    else if (nprod <= 4) { np = 4; k = nsl == 3 ? igemm_ws_kernel<4, 4, 3> : igemm_ws_kernel<4, 4, 4>; }
    else { np = 8; k = nsl == 3 ? igemm_ws_kernel<4, 8, 3> : igemm_ws_kernel<4, 8, 4>; }
  } else {
    if (nprod <= 2) { np = 2; k = nsl == 3 ? igemm_ws_kernel<8, 2, 3> : igemm_ws_kernel<8, 2, 4>; }
    else if (nprod <= 4) { np = 4; k = nsl == 3 ? igemm_ws_kernel<8, 4, 3> : igemm_ws_kernel<8, 4, 4>; }
    else { np = 8; k = nsl == 3 ? igemm_ws_kernel<8, 8, 3> : igemm_ws_kernel<8, 8, 4>; }
  }
  const size_t shm = (size_t)nsl * (128 + 128) * 64 * 2;
  if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess) return 2;
  hipLaunchKernelGGL(k, dim3(grid), dim3(64 * (nc + np)), shm, stream, p, sk, partial, m_fast);
  if (splitk_used) *splitk_used = sk;
  return 0;
}

#ifdef NR_STAMP
extern "C" int nr_ws_stamp_read(void* dst, size_t bytes, int clear) {
  const size_t n = bytes < sizeof(ws_stamp_buf) ? bytes : sizeof(ws_stamp_buf);
  int rc = 0;
  if (dst) rc = (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(ws_stamp_buf), n, 0, hipMemcpyDeviceToHost);
  if (clear) { void* d = nullptr; (void)hipGetSymbolAddress(&d, HIP_SYMBOL(ws_stamp_buf)); (void)hipMemset(d, 0, sizeof(ws_stamp_buf)); }
  return rc;
}
#endif
