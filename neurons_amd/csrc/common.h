// Shared device/host helpers for the neurons_amd HIP library (gfx950 / CDNA4 only).
//
// Physical activation layout everywhere in this library: channels-last frame-images,
//   act[n][y][x][c]  bf16,  n = b*F + f   (b = CFG-batch index, f = frame)
// which is the reference's "(b f) (h w) c" token layout (animatediff/models/attention.py:99,109)
// and also what its per-frame convolutions see after "b c f h w -> (b f) c h w"
// (animatediff/models/resnet.py:14-16).  NCFHW fp32 exists only at the 4-channel latent boundary.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define NR_WAVE 64

__device__ __forceinline__ float bf2f(bf16 v) { return (float)v; }
__device__ __forceinline__ bf16 f2bf(float v) { return (bf16)v; }

__device__ __forceinline__ bf16x8 bf16x8_zero() {
  bf16x8 z;
#pragma unroll
  for (int i = 0; i < 8; ++i) z[i] = (bf16)0.0f;
  return z;
}

// x * sigmoid(x); v_rcp_f32 (1 ulp) instead of the IEEE division sequence: the result is rounded to bf16 anyway
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// exact (erf) GELU, as torch F.gelu default (reference: animatediff/models/motion_module_new.py:508-518)
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
// Same function with erf from Abramowitz-Stegun 7.1.25 (3-term, |erf error| <= 2.5e-5: two orders below the bf16 rounding of the
// product), sign handled through |x|: 11 plain VALU ops + v_rcp + v_exp per element.  The GEGLU epilogues are VALU-bound (K is short:
// 5-20 MFMA k-tiles per output tile), so the gate is kept as short as the output precision allows.
__device__ __forceinline__ float gelu_erf_fast(float x) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(z, 0.47047f, 1.0f));
  float poly = __builtin_fmaf(t, 0.7478556f, -0.0958798f);
  poly = __builtin_fmaf(poly, t, 0.3480242f) * t;
  const float e = __builtin_amdgcn_exp2f(z * z * -1.4426950408889634f);
  const float q = __builtin_fmaf(-poly, e, 1.0f);          // erf(|x| / sqrt 2)
  const float h = 0.5f * x;
  return __builtin_fmaf(fabsf(h), q, h);                   // 0.5 x (1 + sign(x) erf(|x| / sqrt 2))
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float quick_gelu_f(float x) { return x / (1.f + __expf(-1.702f * x)); }

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ----------------------------------------------------------------------------------------------
// Output stores.  The eight XCD L2s are not coherent with each other, so the release at the end of every kernel writes the L2's dirty lines
// back before the next kernel of the stream may start: with plain stores that write-back sits on the critical path between two launches
// (MI355X_MICROARCH.md price list, row "boundary": + B / 6 TB/s for B dirty bytes, 1.7 us behind a 10 MB activation).  NR_STORE_WT builds
// store activations write-through (`sc1`: agent scope, the bytes leave the L2 while the kernel still computes; same instruction rate as a
// plain 16-byte store), so the end-of-kernel release finds nothing to write.  The consumer kernel reads them from the Infinity Cache, where
// 7 of its 8 XCDs had to look anyway.  Data registers: the hardware needs them for two wait states only (s_nop 1), as in rowpanel.hip.
#ifndef NR_STORE_WT
#define NR_STORE_WT 0
#endif
__device__ __forceinline__ void nr_store16(bf16* ptr, const bf16x8& v) {
#if NR_STORE_WT
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(ptr), "v"(v) : "memory");
#else
  *(bf16x8*)ptr = v;
#endif
}
__device__ __forceinline__ void nr_store16f(float* ptr, const f32x4& v) {      // split-K slabs: written by one kernel, read by the reduce
#if NR_STORE_WT
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(ptr), "v"(v) : "memory");
#else
  *(f32x4*)ptr = v;
#endif
}
__device__ __forceinline__ void nr_store8(bf16* ptr, const bf16x4& v) {
#if NR_STORE_WT
  asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" : : "v"(ptr), "v"(v) : "memory");
#else
  *(bf16x4*)ptr = v;
#endif
}

// Kernel arguments pinned in SGPRs at entry.  Left alone, hipcc loads the argument block lazily: one s_load + s_waitcnt lgkmcnt(0) in front of
// every first use (the tiled igemm had 16 such round trips, scalar-cache misses among them, between its entry and its first LDS-DMA: 3,500 of the
// ~20,000 cycles of a short-K workgroup, profiles/r05_igemm_timeline.txt).  An empty asm that takes the value as an "s" operand makes it opaque
// (no rematerialisation from the argument segment), so every pinned field is fetched by the loads the compiler batches at the top of the kernel.
#ifndef NR_PIN_ARGS
#define NR_PIN_ARGS 1     // 0: the lazy loads again (A/B arm: make variant NAME=nopin VFLAGS=-DNR_PIN_ARGS=0)
#endif
template <class T> __device__ __forceinline__ T nr_pin(T v) {
#if NR_PIN_ARGS
  asm volatile("" : "+s"(v));
#endif
  return v;
}

// ----------------------------------------------------------------------------------------------
// Launch-parameter structs shared between the kernels (*.hip) and the engine (engine.hip).
// ----------------------------------------------------------------------------------------------

// C[M][N] = epilogue( A[M][K] * W[N][K]^T ), A given as an implicit im2col view of an NHWC tensor
// (ksize=1: plain row-major GEMM).  See gemm.hip.
struct NrGemmParams {
  const bf16* a0;      // first channel-concat source, pixel-major [pix][lda0]
  const bf16* a1;      // second source (may be null); channels [c0, c0+c1)
  int c0, c1;          // channels taken from a0 / a1 (c1 = 0 when a1 == null)
  int lda0, lda1;      // pixel stride of a0 / a1 in elements
  int H, W;            // source spatial size (before optional nearest-2x upsample)
  int OH, OW;          // output spatial size
  int ksize;           // 1 or 3 (3 => pad 1)
  int stride;          // 1 or 2
  int ups;             // 1: source is nearest-2x upsampled before the 3x3 conv
  const bf16* w;       // [N][K] bf16, K = ksize*ksize*(c0+c1), tap-major (ky,kx,c)
  int M, N, K;
  const float* bias;   // [N] fp32 or null
  const float* rowvec; // [M/rowvec_div][N] fp32 or null (time-embedding add)
  int rowvec_div;      // row m uses rowvec[((m / rowvec_div) % rowvec_mod) * rowvec_ld + n]
  int rowvec_mod;      // 0 = no modulo (time embedding per image batch); F = frame index (temporal PE term)
  int rowvec_ld;
  const bf16* res;     // residual [M][ldr] or null
  int ldr;
  bf16* out;           // [M][ldo]
  int ldo;
  float out_scale;     // (acc + bias + rowvec) * out_scale + res
  int geglu;           // 1: W rows are (value16|gate16)-interleaved; out has N/2 columns
  // LayerNorm folded into the GEMM (1x1 only): w holds gamma-scaled rows W'[n][k] = gamma[k] W[n][k], ln_c[n] = sum_k W'[n][k],
  // bias already contains beta . W[n][:]; the kernel accumulates the row sums of the raw activations it streams anyway and
  // applies out = rstd_m * (acc - mean_m * ln_c[n]) + bias[n] in the epilogue.  null = plain GEMM.
  const float* ln_c;
  float ln_eps;
  int act;             // 0 none; 1 quick_gelu x*sigmoid(1.702x) (CLIP MLP), applied after bias/scale, before the residual
  int pad_tl0;         // 3x3 only: 1 = no top/left padding (bottom/right zero) — the VAE Downsample's F.pad (0,1,0,1)
  float* out_f32;      // non-null: write the raw fp32 accumulators to out_f32[M][N] and skip the epilogue (attention scores)
  int plan_m;          // > 0: every choice that can change the arithmetic of a row (tile -> split-K depth, row-panel eligibility) is made as if the
                       // launch had plan_m rows (the rows of ONE clip's CFG pair), so a clip's result does not depend on its neighbours in the
                       // batch (NR_DETERMINISTIC_BATCH / nr_net_set_deterministic_batch); 0: use M
  int tap_inner;       // 3x3, stride 1, single source: K walks (64-channel chunk, tap) with the TAP fastest; weights [N][Cin/64][9][64].
                       // The 9 re-reads of an activation row segment then fall into 9 consecutive k-tiles (L2 hits) instead of being
                       // spread over the whole K loop (tap-major order: the working set of the tiles in flight exceeds the 4 MiB L2)
  int* sk_ctr;         // non-null: a split-K launch of this description reduces IN the launch (gemm.hip, l2red): one int per output tile, zero before
                       // the first launch and left zero by every launch; owned by this launch description (concurrent streams never share one).
                       // null: fp32 slabs + splitk_reduce_kernel.  Ask nr_igemm_splitk_l2_tiles() how many ints a description needs (0 = not eligible)
  const bf16* w_fm;    // non-null: the same weights in FRAGMENT-MAJOR order for smallm.hip: [N/16][K/32][64 lanes][8], lane (fr, fg) of block
                       // (T, ks) holds W[16 T + fr][32 ks + 8 fg .. + 7], so a wave's MFMA A-operand load is one contiguous KiB (nr_launch_smallm_w_pack)
};

__device__ __forceinline__ NrGemmParams nr_pin_params(NrGemmParams p) {
  p.a0 = nr_pin(p.a0); p.a1 = nr_pin(p.a1); p.c0 = nr_pin(p.c0); p.c1 = nr_pin(p.c1); p.lda0 = nr_pin(p.lda0); p.lda1 = nr_pin(p.lda1);
  p.H = nr_pin(p.H); p.W = nr_pin(p.W); p.OH = nr_pin(p.OH); p.OW = nr_pin(p.OW); p.ksize = nr_pin(p.ksize); p.stride = nr_pin(p.stride);
  p.ups = nr_pin(p.ups); p.w = nr_pin(p.w); p.M = nr_pin(p.M); p.N = nr_pin(p.N); p.K = nr_pin(p.K); p.bias = nr_pin(p.bias);
  p.rowvec = nr_pin(p.rowvec); p.rowvec_div = nr_pin(p.rowvec_div); p.rowvec_mod = nr_pin(p.rowvec_mod); p.rowvec_ld = nr_pin(p.rowvec_ld);
  p.res = nr_pin(p.res); p.ldr = nr_pin(p.ldr); p.out = nr_pin(p.out); p.ldo = nr_pin(p.ldo); p.out_scale = nr_pin(p.out_scale);
  p.geglu = nr_pin(p.geglu); p.ln_c = nr_pin(p.ln_c); p.ln_eps = nr_pin(p.ln_eps); p.act = nr_pin(p.act); p.pad_tl0 = nr_pin(p.pad_tl0);
  p.out_f32 = nr_pin(p.out_f32); p.tap_inner = nr_pin(p.tap_inner); p.w_fm = nr_pin(p.w_fm); p.sk_ctr = nr_pin(p.sk_ctr);
  return p;
}

// Row-panel GEMM (rowpanel.hip): out = epilogue(a . w^T), K = 320, the 256-row panel held in registers
struct NrRowPanelParams {
  const bf16* a; int lda;            // [M][K]
  const bf16* w;                     // [N][K] (GEGLU: value/gate-interleaved rows)
  int M, N;
  const float* bias;                 // [N] or null
  const float* ln_c; float ln_eps;   // LayerNorm folded as in NrGemmParams (ln_c null = plain GEMM)
  const float* rowvec; int rowvec_div, rowvec_mod, rowvec_ld;   // fp32 row-vector term, indexing as NrGemmParams
  const bf16* res; int ldr;          // residual [M][ldr] or null
  bf16* out; int ldo;
  float out_scale; int geglu; int act;
  int nsplit;                        // workgroups per 256-row panel (each takes a contiguous range of 64-column chunks)
  int dbg;                           // timing experiments only (NR_RP_DBG): 1 no stores, 2 no DMA after the prologue, 4 no barrier wait
};

struct NrAttnParams {
  const bf16* q; const bf16* k; const bf16* v; bf16* out;
  // element offset of (batch nb, seq s, head h, dim x):
  //   base(nb) + s*seq + h*d + x,   base(nb) = (nb / inner)*outer + (nb % inner)*inner_stride
  long long q_outer, q_inner_stride, q_seq;
  long long kv_outer, kv_inner_stride, kv_seq;
  long long o_outer, o_inner_stride, o_seq;
  int inner;       // inner batch extent for q/out (spatial: 1, temporal: h*w)
  int kv_inner;    // inner batch extent for k/v
  int kv_div;      // kv batch index = nb / kv_div (cross-attention: frames share one text context)
  int nbatch, heads, d, Lq, Lk;
  float scale;
  int causal;      // 1: key j is visible to query i only if j <= i (CLIP text encoder)
  int fp8;         // 1: OCP e4m3 MFMA operands in the block-shared (spatial / cross) kernel; bf16 is the default
};

// GroupNorm launch parameters (norm.hip)
struct NrGnParams {
  const bf16* x0; const bf16* x1;   // sources: channels [0,c0) from x0, [c0,c0+c1) from x1
  int c0, c1, ld0, ld1;             // pixel strides in elements
  int nimg, hw;                     // images, pixels per image
  int plan_nimg;                    // > 0: kernel variant / slab width / chunking chosen as for plan_nimg images (batch-independent results); 0: nimg
  int groups;
  int pix_per_blk, nchunk;          // pixel chunking: nchunk = ceil(hw / pix_per_blk)
  float* partial;                   // [nimg][nchunk][groups][2], then (large images) [nimg][groups][2] mean/rstd
  int finalized;                    // 1: gn_apply reads mean/rstd written by gn_finalize instead of re-reducing
  const float* gamma; const float* beta;  // [C]
  float eps;
  int silu;
  bf16* out; int ldo;               // [nimg*hw][ldo]
};

// in-place residual adds dst[i] += src[i] of up to 16 tensors in ONE launch (the ControlNet residuals, unet.py:422-439)
struct NrAddMulti {
  bf16* dst[16];
  const bf16* src[16];
  long long n8_end[16];    // exclusive prefix sums of the tensors' sizes in 16-byte units
  int count;
};
