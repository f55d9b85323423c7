// Fused FeedForward(GEGLU) + proj_out block of the 32x32 level (C = 320):
//
//     out = x + bc + [t | g] . Wc^T ,   g = GEGLU( LayerNorm(t) . W1^T + b1 )
//
// i.e. BasicTransformerBlock's / TemporalTransformerBlock's last FeedForward with its residual add, followed by the transformer's
// proj_out with ITS residual add (reference: animatediff/models/attention.py:129-140,297-299; motion_module.py:150-158,219-221;
// FeedForward / GEGLU motion_module_new.py:441-518), in ONE launch.  Wc = [Wpo | Wpo Wff2] and bc = bpo + Wpo bff2 are the folded
// net.2 + proj_out of engine.hip (w_fold_ff_proj); W1 carries the LayerNorm fold of w_ln_linear (gamma-scaled rows, value/gate
// interleave).  Until round 3 this was two launches (GEGLU projection 84 us + K = 5C GEMM 53 us at M = 32768) with the 4C-wide hidden
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
constexpr int FF_ROWS = 128;                 // rows per workgroup (4 waves x 32)
constexpr int FF_SUB = 64 * 64;              // elements of one [64][64] sub-tile
constexpr int FF_STAGE = 5 * FF_SUB;         // elements of one stage (40 KiB)
constexpr int FF_NS = 3;                     // ring depth
constexpr int FF_TSTAGES = FF_C / 64;        // 5: t-part of the second GEMM
constexpr int FF_PAIRS = (4 * FF_C) / 64;    // 20: pairs of W1 chunks (64 hidden units per pair)
constexpr int FF_NSTAGES = FF_TSTAGES + 3 * FF_PAIRS;   // 65
constexpr int FF_DMA = FF_STAGE * 2 / (4 * 1024);       // 10 LDS-DMA instructions per wave and stage

struct NrFFParams {
  const bf16* t; int ldt;        // [M][C] residual stream inside the transformer (pre-LayerNorm)
  const bf16* x; int ldx;        // [M][C] transformer input (outer residual)
  bf16* out; int ldo;            // [M][C]
  int M;
  const bf16* stream;            // 65 stages x 40 KiB (ff_stream_pack_kernel)
  const float* c1;               // [8C] sum_k W1'[n][k]     (LayerNorm fold, value/gate-interleaved row order)
  const float* b1;               // [8C] b1 + beta . W1
  const float* bc;               // [C]  bpo + Wpo bff2
  float ln_eps;
};

__global__ __launch_bounds__(256) void ff_fused_kernel(NrFFParams p) {
  constexpr int C = FF_C, KS = C / 32, NT2 = C / 16;       // 10 k-steps of the panel, 20 16-column groups of the output
  extern __shared__ __attribute__((aligned(16))) bf16 smem[];   // FF_NS stages, then c1 | b1 (2 x 8C floats)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;

  // ---- weight stream: every workgroup walks the same 65 stages; wave w copies bytes [10 w KiB, 10 (w+1) KiB) of each ----
  const char* wsrc = reinterpret_cast<const char*>(p.stream) + (size_t)wave * (FF_DMA * 1024) + (size_t)lane * 16;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(lptr_t)smem) + (unsigned)wave * (FF_DMA * 1024);
  auto issue = [&](int stage) {
    const unsigned dst = lds0 + (unsigned)((stage % FF_NS) * FF_STAGE * 2);
    const char* src = wsrc + (size_t)stage * (FF_STAGE * 2);
#pragma unroll
    for (int i = 0; i < FF_DMA; ++i) glds16(src + i * 1024, dst + (unsigned)(i * 1024));
  };
  issue(0);
  issue(1);
  // the GEGLU epilogue constants (LayerNorm-fold column sums and folded bias, 2 x 8C floats) live in LDS behind the ring: a global load
  // inside the stage loop would make hipcc wait for it with a vmcnt that also drains the LDS-DMA issued before it
  float* sC1 = reinterpret_cast<float*>(smem + FF_NS * FF_STAGE);
  float* sB1 = sC1 + 8 * C;
  for (int i = tid * 4; i < 8 * C; i += 256 * 4) {
    *(f32x4*)(sC1 + i) = *(const f32x4*)(p.c1 + i);
    *(f32x4*)(sB1 + i) = *(const f32x4*)(p.b1 + i);
  }
  __syncthreads();      // the constants are visible to every wave before the first epilogue (the stage barriers are raw s_barrier)

  // ---- the row panel: lane holds row (16 mt + fr) of its wave's 32 rows, k = 32 ks + 8 fg .. +7 ----
  const int mrow0 = blockIdx.x * FF_ROWS + wave * 32;
  bf16x8 xb[2][KS];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    int m = mrow0 + 16 * mt + fr;
    m = m < p.M ? m : p.M - 1;
    const bf16* ap = p.t + (size_t)m * p.ldt + 8 * fg;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) xb[mt][ks] = *(const bf16x8*)(ap + 32 * ks);
  }
  // LayerNorm statistics from the panel (fp32 sums of the raw bf16 pairs; var = E[x^2] - mean^2 as in gemm.hip LNF / rowpanel.hip)
  float rs[2], mr[2];      // rstd_m and mean_m * rstd_m
  {
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

  f32x4 oacc[NT2][2];
#pragma unroll
  for (int nt = 0; nt < NT2; ++nt)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) oacc[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 gB[2][2];            // [k-step of the g-piece = W1 chunk parity][mt]

  // A fragment of a [320 n][64 k] stage (t-part / g-piece): 16 n rows nt, k-step ks2 (32 deep)
  auto frag_n320 = [&](const bf16* sW, int nt, int ks2) {
    const int row = (nt & 3) * 16 + fr;
    return *(const bf16x8*)(sW + (nt >> 2) * FF_SUB + row * 64 + (((ks2 * 4 + fg) ^ (row & 7)) << 3));
  };
  // out += A(stage [320 n][64 k]) x B(b0 | b1 per mt): 40 fragment reads, 80 MFMAs; reads run one 4-group ahead of the MFMAs
  auto gemm_n320 = [&](const bf16* sW, const bf16x8 (&b0)[2], const bf16x8 (&b1)[2]) {
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
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int nt = 4 * q + i;
        oacc[nt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur[i], ks2 ? b1[0] : b0[0], oacc[nt][0], 0, 0, 0);
        oacc[nt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur[i], ks2 ? b1[1] : b0[1], oacc[nt][1], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- stage sequencing: ring index and waits are runtime values, everything that indexes registers is unrolled ----
  int s = 0, ring = 0;
  auto stage_begin = [&]() -> const bf16* {
    // stage s must have landed: this wave's DMA of stage s + 1 (issued one stage ago) may stay outstanding.  No other vector-memory
    // operation is issued inside the loop (the GEGLU constants sit in LDS), so the count is exact.
    if (s + 1 < FF_NSTAGES) wait_vmcnt<FF_DMA>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();            // everyone's pieces of stage s landed; everyone is done reading stage s - 1
    if (s + 2 < FF_NSTAGES) issue(s + 2);    // into the buffer stage s - 1 occupied
    const bf16* sW = smem + ring * FF_STAGE;
    ++s;
    ring = ring + 1 == FF_NS ? 0 : ring + 1;
    return sW;
  };

  // ---- t-part: out += t[:, 64 ts .. 64 ts + 63] . Wpo[:, same]^T ----
#pragma unroll
  for (int ts = 0; ts < FF_TSTAGES; ++ts) {
    const bf16* sW = stage_begin();
    const bf16x8 b0[2] = {xb[0][2 * ts], xb[1][2 * ts]};
    const bf16x8 b1[2] = {xb[0][2 * ts + 1], xb[1][2 * ts + 1]};
    gemm_n320(sW, b0, b1);
  }

  for (int pair = 0; pair < FF_PAIRS; ++pair) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      // ---- W1 chunk: 64 interleaved rows = 32 hidden units (16 value | 16 gate | 16 value | 16 gate) ----
      const bf16* sW = stage_begin();
      const int nw0 = (2 * pair + r) * 64;
      f32x4 acc[4][2];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) acc[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
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
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[nt], xb[mt][ks], acc[nt][mt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      // GEGLU epilogue in fragment layout: lane holds rows (16 mt + fr), interleaved columns 16 nt + 4 fg + e
      f32x4 bv[4], cv[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        bv[nt] = *(const f32x4*)(sB1 + nw0 + 16 * nt + 4 * fg);
        cv[nt] = *(const f32x4*)(sC1 + nw0 + 16 * nt + 4 * fg);
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        bf16x8 gb;
#pragma unroll
        for (int h = 0; h < 2; ++h) {              // value tile 2 h, gate tile 2 h + 1
          const f32x4 v = acc[2 * h][mt] * rs[mt] + (bv[2 * h] - cv[2 * h] * mr[mt]);
          const f32x4 g = acc[2 * h + 1][mt] * rs[mt] + (bv[2 * h + 1] - cv[2 * h + 1] * mr[mt]);
#pragma unroll
          for (int e = 0; e < 4; ++e) gb[4 * h + e] = (bf16)(v[e] * gelu_erf_fast(g[e]));
        }
        gB[r][mt] = gb;
      }
    }
    // ---- g-piece: out += g[:, 64 pair .. +63] . (Wpo Wff2)[:, same]^T ----
    const bf16* sW = stage_begin();
    gemm_n320(sW, gB[0], gB[1]);
  }

  // ---- epilogue: out = x + bc + acc, lane holds rows (16 mt + fr), columns 16 nt + 4 fg .. +3 ----
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
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
      *(bf16x4*)(orow + 16 * nt) = o;
    }
  }
}

// Builds the 65-stage weight stream from W1' ([8C][C] bf16: LayerNorm-folded, value/gate-interleaved rows, engine w_ln_linear) and
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

}  // namespace

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
                                  const float* c1, const float* b1, const float* bc, float ln_eps, hipStream_t s) {
  if (M <= 0 || ldt % 8 != 0 || ldx % 4 != 0 || ldo % 4 != 0) return 1;
  NrFFParams p;
  p.t = t; p.ldt = ldt; p.x = x; p.ldx = ldx; p.out = out; p.ldo = ldo; p.M = M; p.stream = stream; p.c1 = c1; p.b1 = b1; p.bc = bc;
  p.ln_eps = ln_eps;
  constexpr size_t shm = (size_t)FF_NS * FF_STAGE * sizeof(bf16) + (size_t)2 * 8 * FF_C * sizeof(float);
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!(g_ff_attr >> (dev & 63) & 1ull)) {
    if (hipFuncSetAttribute((const void*)ff_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess) return 2;
    g_ff_attr |= 1ull << (dev & 63);
  }
  hipLaunchKernelGGL(ff_fused_kernel, dim3((unsigned)((M + FF_ROWS - 1) / FF_ROWS)), dim3(256), shm, s, p);
  return 0;
}
