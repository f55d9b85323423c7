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
#else
#define L1_STAMP(slot) do { } while (0)
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
extern "C" int nr_lin160_eligible(const NrGemmParams* pp) {
  static const bool off = getenv("NR_LIN160") && getenv("NR_LIN160")[0] == '0';   // A/B switch
  const NrGemmParams& p = *pp;
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
  if (!stream || g.M % 64 != 0 || g.N % L1_BN != 0 || g.K % 64 != 0 || g.K / 64 < L1_NS) return 1;
  NrLin160Params p;
  p.x = g.a0; p.lda = g.lda0; p.stream = stream; p.bias = g.bias; p.res = g.res; p.ldr = g.ldr; p.out = g.out; p.ldo = g.ldo; p.M = g.M; p.N = g.N; p.K = g.K; p.norot = g.plan_m > 0 ? 1 : 0;
  const int Mp = (g.plan_m > 0 && g.plan_m < g.M) ? g.plan_m : g.M;
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
