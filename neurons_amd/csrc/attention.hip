// Flash-style softmax(scale * Q K^T) V on bf16 MFMA for every attention on the denoiser path:
//   spatial self-attention  (L = h*w tokens per frame-image, heads 8, d = C/8 in {40,80,160})
//   text cross-attention    (Lk = 77 keys shared by all frames of a clip)
//   temporal self-attention (L = F frames per pixel; the "(b f) d c -> (b d) f c" regroup of
//                            animatediff/models/motion_module.py:274-278,327 is done purely by the
//                            q/k/v/out strides below - nothing is transposed in memory)
// Reference arithmetic: diffusers-0.11.1 CrossAttention._attention, vendored at
// animatediff/models/motion_module_new.py:258-287 (baddbmm(alpha=scale) -> softmax -> bmm) with the
// head split of :181-193.  The score matrix is never materialised; softmax statistics are fp32.
//
// Work unit: one wave = 16 queries of one (batch, head).  The scores are computed TRANSPOSED,
// S^T = K Q^T (keys on MFMA rows, queries on lanes), so a lane's accumulator registers are the
// probabilities of ITS query for 8 keys - exactly the B-operand fragment the second product
// O^T = V^T P^T needs (k-slot permutation applied identically to both operands), i.e. P never leaves
// registers.  Row max / row sum are two xor-shuffles across the 4 lane groups.  V tiles are staged
// transposed in a per-wave LDS region.
#include "common.h"

namespace {

constexpr int KT = 32;        // keys per iteration
// LDS row stride (elements) of a wave's V tile [32 keys][d]: 16 (mod 32) so the 8 rows a half-wave touches in one
// ds_read_b64_tr_b16 start 8 banks apart (as KSV of the block-shared kernel below)
__host__ __device__ constexpr int vt_stride(int DT) { return (DT * 16) % 32 == 16 ? DT * 16 : DT * 16 + 16; }

// DK = ceil(d/32) k-steps for QK^T;  DT = ceil(d/16) row tiles of O^T
template <int DK, int DT>
__global__ __launch_bounds__(256) void attn_fwd_kernel(NrAttnParams p) {
  extern __shared__ __attribute__((aligned(16))) bf16 vt_all[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  constexpr int VS = vt_stride(DT);
  bf16* vt = vt_all + (size_t)wave * KT * VS;             // this wave's private V tile, row-major [key][dim]
  const int c = lane & 15, g = lane >> 4;
  const int tq = c >> 2, tp = c & 3;                      // transpose-read role inside the 16-lane group

  const int qtiles = (p.Lq + 15) >> 4;
  const long long total = (long long)p.nbatch * p.heads * qtiles;
  long long wid = (long long)blockIdx.x * 4 + wave;
  const bool active = wid < total;
  if (!active) wid = total - 1;   // keep the wave in step with the block barriers; results discarded
  const int qt = (int)(wid % qtiles);
  const int h = (int)((wid / qtiles) % p.heads);
  const int nb = (int)(wid / ((long long)qtiles * p.heads));
  const int nbkv = nb / p.kv_div;

  const long long qbase = (long long)(nb / p.inner) * p.q_outer + (long long)(nb % p.inner) * p.q_inner_stride + (long long)h * p.d;
  const long long obase = (long long)(nb / p.inner) * p.o_outer + (long long)(nb % p.inner) * p.o_inner_stride + (long long)h * p.d;
  const long long kbase = (long long)(nbkv / p.kv_inner) * p.kv_outer + (long long)(nbkv % p.kv_inner) * p.kv_inner_stride + (long long)h * p.d;

  const bf16x8 zero8 = bf16x8_zero();
  const int qrow = qt * 16 + c;
  const int qrow_c = min(qrow, p.Lq - 1);
  const int klim = p.causal ? min(p.Lk, qrow + 1) : p.Lk;   // keys visible to this lane's query

  // Q fragments (B operand of S^T = K Q^T): lane holds Q[query c][dim 32*ks + 8*g .. +7]
  bf16x8 qf[DK];
#pragma unroll
  for (int ks = 0; ks < DK; ++ks) {
    const int dim0 = ks * 32 + g * 8;
    qf[ks] = dim0 < p.d ? *(const bf16x8*)(p.q + qbase + (long long)qrow_c * p.q_seq + dim0) : zero8;
  }

  f32x4 acc[DT];
#pragma unroll
  for (int i = 0; i < DT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_i = -1e30f, l_i = 0.f;

  const int chunks_per_key = p.d >> 3;
  const int vchunks = KT * chunks_per_key;
  // the tile is wave-private and a wave's LDS operations execute in order, so no workgroup barrier is needed around it; the pad
  // columns [d, 16 DT) are cleared once (they only feed output rows that are never stored, but must be finite)
  for (int id = lane; id < KT * (VS >> 3); id += 64) *(bf16x8*)(vt + id * 8) = zero8;

  for (int k0 = 0; k0 < p.Lk; k0 += KT) {
    // ---- stage the V tile row-major with 16-byte stores: vt[key - k0][dim]; the MFMA A operand (V^T) is read transposed ----
    __builtin_amdgcn_wave_barrier();
    for (int id = lane; id < vchunks; id += 64) {
      const int kk = id / chunks_per_key;
      const int x0 = (id - kk * chunks_per_key) << 3;
      const int key = k0 + kk;
      bf16x8 vv = zero8;
      if (key < p.Lk) vv = *(const bf16x8*)(p.v + kbase + (long long)key * p.kv_seq + x0);
      *(bf16x8*)(vt + kk * VS + x0) = vv;
    }

    // ---- S^T = K Q^T for two 16-key tiles ----
    f32x4 s[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int key = k0 + 16 * t + c;
      const int keyc = min(key, p.Lk - 1);
      const bf16* kp = p.k + kbase + (long long)keyc * p.kv_seq;
#pragma unroll
      for (int ks = 0; ks < DK; ++ks) {
        const int dim0 = ks * 32 + g * 8;
        const bf16x8 kf = dim0 < p.d ? *(const bf16x8*)(kp + dim0) : zero8;
        s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], s[t], 0, 0, 0);
      }
    }
    // s[t][r]: key = k0 + 16t + 4g + r, query = c
    float mx = -1e30f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = k0 + 16 * t + 4 * g + r;
        const float v = key < klim ? s[t][r] * p.scale : -1e30f;
        s[t][r] = v;
        mx = fmaxf(mx, v);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_i, mx);
    const float alpha = __expf(m_i - m_new);
    float rs = 0.f;
    bf16x8 pf;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = k0 + 16 * t + 4 * g + r;
        const float e = key < klim ? __expf(s[t][r] - m_new) : 0.f;
        rs += e;
        pf[t * 4 + r] = (bf16)e;
      }
    rs += __shfl_xor(rs, 16, 64);
    rs += __shfl_xor(rs, 32, 64);
    l_i = l_i * alpha + rs;
    m_i = m_new;
#pragma unroll
    for (int i = 0; i < DT; ++i) acc[i] *= alpha;

    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();   // the wave's own stores are ordered before its transposed reads
    // ---- O^T += V^T P^T : A = V^T rows dv = 16*i + c, k-slot j -> key (j<4: 4g+j ; j>=4: 16+4g+j-4) ----
#pragma unroll
    for (int i = 0; i < DT; ++i) {
      const bf16* a0 = vt + (4 * g + tq) * VS + 16 * i + 4 * tp;
      const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)a0);
      const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(a0 + 16 * VS));
      bf16x8 vf;
#pragma unroll
      for (int e = 0; e < 4; ++e) { vf[e] = lo[e]; vf[4 + e] = hi[e]; }
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, acc[i], 0, 0, 0);
    }
  }

  // ---- epilogue: acc[i][r] = O[query c][dv = 16i + 4g + r] ----
  if (active && qrow < p.Lq) {
    const float inv = 1.0f / l_i;
    bf16* op = p.out + obase + (long long)qrow * p.o_seq;
#pragma unroll
    for (int i = 0; i < DT; ++i) {
      const int dv0 = i * 16 + 4 * g;
      if (dv0 < p.d) {
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16)(acc[i][e] * inv);
        nr_store8(op + dv0, o);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Block-shared variant for long sequences (spatial self-attention, text cross-attention): a 256-thread
// workgroup = 4 waves = 64 queries of one (batch, head); K and V tiles of 64 keys are staged ONCE per
// workgroup into LDS (row-major, 16-byte loads -> ds_write_b128, register-prefetched one tile ahead,
// two LDS buffers, one barrier per tile).  K rows are the A operand of S^T = K Q^T (ds_read_b128, row
// stride an odd number of 16-B units => conflict-free); V is consumed through the CDNA4 transpose read
// ds_read_b64_tr_b16, which hands each lane V[4 keys][its dv column] = the A fragment of O^T = V^T P^T.
// ------------------------------------------------------------------------------------------------
constexpr int KT2 = 64;

// OCP e4m3 operands for the fp8 variant (BASELINE config 5): 8 values -> one 64-bit MFMA operand (v_cvt_pk_fp8_f32; gfx950 = OCP)
__device__ __forceinline__ long to_fp8x8(const bf16x8& v) {
  int lo = 0, hi = 0;
  lo = __builtin_amdgcn_cvt_pk_fp8_f32((float)v[0], (float)v[1], lo, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32((float)v[2], (float)v[3], lo, true);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32((float)v[4], (float)v[5], hi, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32((float)v[6], (float)v[7], hi, true);
  return (long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ long to_fp8x8(const float (&v)[8]) {
  int lo = 0, hi = 0;
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], lo, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], lo, true);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], hi, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], hi, true);
  return (long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}

// FP8: both products (S^T = K Q^T and O^T = V^T P^T) on v_mfma_f32_16x16x32_fp8_fp8 with OCP e4m3 operands converted in
// registers (Q, K, V from their bf16 tiles; P straight from its fp32 exponentials, scaled by 256 so the probabilities use the
// e4m3 normal range, and the sum divided back in fp32); softmax statistics and accumulators stay fp32.  The non-scaled fp8 MFMA
// issues at the bf16 rate (MI355X_MICROARCH.md), so this variant is about the numerics of config 5, not about speed.
// max / sum over the four 16-lane rows of a wave (lanes l, l^16, l^32, l^48) on the VALU: v_permlane16_swap exchanges the odd rows of
// its first operand with the even rows of its second, v_permlane32_swap the upper half of the first with the lower half of the second;
// with both operands = v every lane gets {own, partner} in the two results (a ds_bpermute round trip through the LDS otherwise).
__device__ __forceinline__ float rows_max(float v) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float rows_sum(float v) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// acc += A B with the accumulator TIED (vDst = SrcC).  In the ONES instantiation hipcc 7.2 otherwise allocates the third accumulator as a
// v[2:5] -> v[0:3] -> v[2:5] chain (destination partially overlapping the SrcC the previous MFMA wrote) with no wait states between the
// dependent MFMAs, and the kernel returns wrong sums on gfx950 (tools/check_mfma_overlap.py scans the shipped ISA for that pattern).
// The compiler keeps no hazard bookkeeping for an MFMA it cannot see, so the asm carries its own:
//  * VALU write -> MFMA read of that register needs wait states (measured: a v_cvt_pk_bf16_f32 of the P operand directly in front of the
//    MFMA, with only an already-satisfied s_waitcnt between them, gave NaN; one wait state cures it): the leading s_nop 1 gives two,
//    the figure LLVM uses for its own MFMAs;
//  * MFMA result -> VALU / LDS read: the accumulators are next touched by the rescale (after the next tile's barrier and its eight K Q^T
//    MFMAs) or after the loop's final barrier -- far beyond the 18 wait states of an 8-pass MFMA.
__device__ __forceinline__ void mfma_bf16_tied(f32x4& acc, const bf16x8& a, const bf16x8& b) {
  asm("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}

// LDS images of a 64-key tile.  K (d <= 64): [64][64] bf16, 128-byte rows with the 16-byte chunk index XOR-swizzled by (key & 7): the
// b128 fragment reads (lane = key 16t + c, chunk 4ks + g) are conflict-free, the layout gemm.hip uses for its operand tiles; larger
// heads keep padded rows of 32 DK + 8 elements.  V: rows of KSV elements with KSV = 16 (mod 32), so the 8 rows a half-wave touches in
// one ds_read_b64_tr_b16 start 8 banks apart (conflict-free transpose reads).  Pad columns are zero (written once).
// ONES (d = 40 in its 48-wide V image only): pad column d of the V image holds 1.0 for every key, so row d of O^T = V^T P^T IS the softmax
// denominator: the sum of the probabilities exactly as the matrix pipe sees them (bf16 / e4m3, fp32 accumulation), rescaled with the
// accumulator for free.  It replaces 17 dependent v_add_f32 per key tile in a loop whose bound is the vector issue port.
template <int DK, int DT, bool FP8 = false, bool ONES = false>
__global__ __launch_bounds__(256) void attn_fwd_shared_kernel(NrAttnParams p) {
  static_assert(!ONES || (DK == 2 && DT == 3), "the ones column lives in the pad of the 48-wide image of d = 40");
  constexpr bool KSWZ = DK <= 2;
  constexpr int KSK = KSWZ ? 64 : DK * 32 + 8;
  constexpr int KSV = (DT * 16) % 32 == 16 ? DT * 16 : DT * 16 + 16;
  constexpr int CH = (DT + 1) / 2;            // 16-B chunks per thread per matrix per tile
  extern __shared__ __attribute__((aligned(16))) bf16 lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  constexpr int TILEK = KT2 * KSK, TILEV = KT2 * KSV, TILE2 = TILEK + TILEV;   // elements per buffer
  const int qblocks = (p.Lq + 63) >> 6;
  // XCD-aware remap (bijective, as in gemm.hip): workgroups are dealt round-robin over the 8 XCDs, so give each XCD a CONTIGUOUS range
  // of (image, head, query-block) ids: the 16 query blocks of a head, and the heads that share its 128-byte lines of the fused q|k|v
  // rows, then read their K/V through ONE L2 instead of fetching them into up to eight
  int bid;
  {
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7, local = orig >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
  }
  const int qb = bid % qblocks;
  const int h = (bid / qblocks) % p.heads;
  const int nb = bid / (qblocks * p.heads);
  const int nbkv = nb / p.kv_div;
  const long long qbase = (long long)(nb / p.inner) * p.q_outer + (long long)(nb % p.inner) * p.q_inner_stride + (long long)h * p.d;
  const long long obase = (long long)(nb / p.inner) * p.o_outer + (long long)(nb % p.inner) * p.o_inner_stride + (long long)h * p.d;
  const long long kbase = (long long)(nbkv / p.kv_inner) * p.kv_outer + (long long)(nbkv % p.kv_inner) * p.kv_inner_stride + (long long)h * p.d;

  const bf16x8 zero8 = bf16x8_zero();
  const int qrow = qb * 64 + wave * 16 + c;
  const int qrow_c = min(qrow, p.Lq - 1);
  bf16x8 qf[DK];
#pragma unroll
  for (int ks = 0; ks < DK; ++ks) {
    const int dim0 = ks * 32 + g * 8;
    qf[ks] = dim0 < p.d ? *(const bf16x8*)(p.q + qbase + (long long)qrow_c * p.q_seq + dim0) : zero8;
  }

  // d = 40 / 48 (DK = 2, DT = 3): the second k-step of K Q^T covers dims 32..47 only, so it runs on v_mfma_f32_16x16x16_bf16 (K = 16: half the
  // matrix-pipe cycles of the 32-deep form, whose upper 16 dims would multiply zeros): lane group g supplies dims 32 + 4g .. 32 + 4g + 3
  constexpr bool K48 = !FP8 && DK == 2 && DT == 3;
  typedef __attribute__((ext_vector_type(4))) short s16x4;
  s16x4 qf4 = {0, 0, 0, 0};
  if constexpr (K48) {
    const int dim0 = 32 + 4 * g;
    if (dim0 < p.d) qf4 = *(const s16x4*)(p.q + qbase + (long long)qrow_c * p.q_seq + dim0);
  }
  long qf8[DK];
  if constexpr (FP8) {
#pragma unroll
    for (int ks = 0; ks < DK; ++ks) qf8[ks] = to_fp8x8(qf[ks]);
  }
  const int cpk = p.d >> 3;                   // chunks per key row
  const int nchunk = KT2 * cpk;
  bf16x8 rk[CH], rv[CH];
  // staging addresses: each thread owns CH fixed (key-in-tile, 16-byte chunk) slots, so its K / V source pointers just advance by one
  // tile per iteration; only the last, partial tile clamps the key (the clamped rows are finite data that the softmax masks out)
  int s_kk[CH];
  int s_off[CH];                              // element offset of the slot inside one (batch, head) K / V sequence (fits 32 bits)
  int s_lok[CH], s_lov[CH];                   // LDS element offsets of the slot in the K / V image
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int id = min(tid + 256 * i, nchunk - 1);
    const int kk = id / cpk, x0 = (id - kk * cpk) << 3;
    s_kk[i] = kk;
    s_lok[i] = KSWZ ? kk * KSK + ((((x0 >> 3) ^ (kk & 7))) << 3) : kk * KSK + x0;
    s_lov[i] = kk * KSV + x0;
    s_off[i] = kk * p.kv_seq + x0;
  }
  // workgroup-uniform bases (scalar registers) + 32-bit lane offsets: two VGPRs per slot instead of two 64-bit pointers
  const bf16* kb = p.k + kbase;
  const bf16* vb = p.v + kbase;
  const long long tile_step = (long long)KT2 * p.kv_seq;
  auto load_regs = [&](int k0) {
    const bf16* kt = kb + (long long)(k0 / KT2) * tile_step;
    const bf16* vtp = vb + (long long)(k0 / KT2) * tile_step;
    if (k0 + KT2 <= p.Lk) {
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        rk[i] = *(const bf16x8*)(kt + s_off[i]);
        rv[i] = *(const bf16x8*)(vtp + s_off[i]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const int back = max(k0 + s_kk[i] - (p.Lk - 1), 0) * p.kv_seq;
        rk[i] = *(const bf16x8*)(kt + (s_off[i] - back));
        rv[i] = *(const bf16x8*)(vtp + (s_off[i] - back));
      }
    }
  };
  auto write_lds = [&](int buf) {
    bf16* sk = lds + buf * TILE2;
    bf16* sv = sk + TILEK;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      if (tid + 256 * i < nchunk) {
        *(bf16x8*)(sk + s_lok[i]) = rk[i];
        *(bf16x8*)(sv + s_lov[i]) = rv[i];
      }
    }
  };

  f32x4 acc[DT];
#pragma unroll
  for (int i = 0; i < DT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_i = -1e30f, l_i = 0.f;

  // every pad column of the K / V images is zero in both buffers (whole images cleared once, the staging only writes the d data
  // columns): the fragment reads below need no head-dim predicate (a predicated 16-byte LDS read costs an exec-mask branch each)
  for (int id = tid; id < 2 * TILE2 / 8; id += 256) *(bf16x8*)(lds + id * 8) = zero8;
  __syncthreads();
  if constexpr (ONES) {                        // column d of both V images (the staging below never touches pad columns)
    if (tid < 2 * KT2) lds[(tid / KT2) * TILE2 + TILEK + (tid % KT2) * KSV + p.d] = (bf16)1.0f;
  }
  const int ntile = (p.Lk + KT2 - 1) / KT2;
  load_regs(0);
  write_lds(0);
  __syncthreads();
  // a zero accumulator the compiler cannot rematerialise: it stays in 4 registers instead of 16 v_mov per key tile
  f32x4 zacc = f32x4{0.f, 0.f, 0.f, 0.f};
  asm volatile("" : "+v"(zacc));
  const int tq = c >> 2, tp = c & 3;          // transpose-read role inside the 16-lane group
  for (int it = 0; it < ntile; ++it) {
    const int k0 = it * KT2;
    if (it + 1 < ntile) load_regs(k0 + KT2);
    const bf16* sk = lds + (it & 1) * TILE2;
    const bf16* sv = sk + TILEK;
    // ---- S^T = K Q^T for four 16-key tiles ----
    f32x4 s[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      s[t] = zacc;
#pragma unroll
      for (int ks = 0; ks < DK; ++ks) {
        if constexpr (K48) {
          if (ks == 1) {       // dims 32 + 4g ..: 8 bytes of chunk 4 + (g >> 1) of the swizzled K row (pad dims >= d are zero in the image)
            const s16x4 kf4 = *(const s16x4*)(sk + (16 * t + c) * KSK + ((((4 + (g >> 1)) ^ (c & 7)) << 3) + ((g & 1) << 2)));
            s[t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(kf4, qf4, s[t], 0, 0, 0);
            continue;
          }
        }
        const bf16x8 kf = KSWZ ? *(const bf16x8*)(sk + (16 * t + c) * KSK + (((ks * 4 + g) ^ (c & 7)) << 3))
                               : *(const bf16x8*)(sk + (16 * t + c) * KSK + ks * 32 + g * 8);
        if constexpr (FP8) s[t] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(to_fp8x8(kf), qf8[ks], s[t], 0, 0, 0);
        else s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], s[t], 0, 0, 0);
      }
    }
    // softmax in the log2 domain: exp(scale*s - m) = exp2(fma(s, scale*log2(e), -m2)): one fma + one v_exp per score.
    // Full tiles (all 64 keys valid) skip the key-mask selects.
    const float sl2 = p.scale * 1.4426950408889634f;
    const bool full = k0 + KT2 <= p.Lk;
    float mx = -1e30f;
    if (full) {
#pragma unroll
      for (int t = 0; t < 4; ++t) mx = fmaxf(fmaxf(fmaxf(fmaxf(mx, s[t][0]), s[t][1]), s[t][2]), s[t][3]);      // 8 x v_max3_f32
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = k0 + 16 * t + 4 * g + r;
          const float v = key < p.Lk ? s[t][r] : -1e30f;
          s[t][r] = v;
          mx = fmaxf(mx, v);
        }
    }
    // Lazy running maximum: the reference maximum of a query only moves when the tile's maximum exceeds it by more than 2^8, so the
    // probabilities stay <= 256 (exact in the fp32 sums, in range for bf16 / scaled e4m3) and the accumulator rescale, which costs a
    // register round trip of the whole O tile, runs on the first tile and then almost never (wave-uniform skip).
    constexpr float LAZY = FP8 ? 0.0f : 8.0f;  // e4m3 probabilities keep the exact running maximum (P <= 1, scaled by 256 below)
    // The cross-lane maximum (two permlane swaps + selects: ~16 issue slots of a loop bound by its issue port) is only needed when the
    // reference moves.  If NO lane's own 16 scores exceed its query's reference by the margin, no query's tile maximum does (it is the
    // maximum over that query's four lanes), so m_new = m_i for every lane exactly as the full computation would find: skip it.
    mx *= sl2;                                 // sl2 > 0: the max of this lane's scaled scores
    float m_new = m_i;
    if (__builtin_amdgcn_ballot_w64(mx > m_i + LAZY) != 0ull) {
      mx = rows_max(mx);
      m_new = mx > m_i + LAZY ? mx : m_i;
      // l_i stays a per-lane partial sum (this lane's 16 keys of every tile); the four rows are added once after the loop
      const float alpha = __builtin_amdgcn_exp2f(m_i - m_new);      // 1 for the queries whose reference stays
      if constexpr (!ONES) l_i *= alpha;
#pragma unroll
      for (int i = 0; i < DT; ++i) acc[i] *= alpha;
      m_i = m_new;
    }
    float rs = 0.f;
    bf16x8 pf[2];
    float pe[2][8];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s[t][r], sl2, -m_new));   // masked keys: exp2(-huge) = 0
        if constexpr (!ONES) rs += e;
        if constexpr (FP8) pe[t >> 1][(t & 1) * 4 + r] = e * 256.0f;
        else pf[t >> 1][(t & 1) * 4 + r] = (bf16)e;
      }
    long pf8[2];
    if constexpr (FP8) { pf8[0] = to_fp8x8(pe[0]); pf8[1] = to_fp8x8(pe[1]); }
    if constexpr (!ONES) l_i += rs;
    // ---- O^T += V^T P^T ; k-slot j of step u -> key 32u + (j<4 ? 4g+j : 16+4g+j-4) ----
#pragma unroll
    for (int i = 0; i < DT; ++i) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const bf16* a0 = sv + (32 * u + 4 * g + tq) * KSV + 16 * i + 4 * tp;
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)a0);
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(a0 + 16 * KSV));
        bf16x8 vf;
#pragma unroll
        for (int e = 0; e < 4; ++e) { vf[e] = lo[e]; vf[4 + e] = hi[e]; }
        if constexpr (FP8) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(to_fp8x8(vf), pf8[u], acc[i], 0, 0, 0);
        else if constexpr (ONES) mfma_bf16_tied(acc[i], vf, pf[u]);
        else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[u], acc[i], 0, 0, 0);
      }
    }
    if (it + 1 < ntile) write_lds((it + 1) & 1);
    __syncthreads();
  }

  // all lanes active here.  ONES: dv = 40 = 16 * 2 + 4 * 2 + 0 is element 0 of acc[2] in lane group g = 2 of the query's column c
  float l_all;
  // ONES: the last P.V MFMAs are inline asm (mfma_bf16_tied), so hipcc does not know acc[] is an in-flight MFMA result and inserts no wait
  // states in front of its first reader (the ds_bpermute of the shuffle below).  An 8-pass XDL write needs ~11 issue slots on gfx950; the
  // compiled stream happened to leave ~13 (two branches, a wait, a barrier, four VALU ops): state the distance instead of relying on it.
  if constexpr (ONES) asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
  if constexpr (ONES) l_all = __shfl(acc[2][0], 32 + c, 64);
  else l_all = rows_sum(l_i);
  if (qrow < p.Lq) {
    const float inv = (FP8 && !ONES ? 1.0f / 256.0f : 1.0f) / l_all;      // ONES + FP8: the 256 of the scaled probabilities is in l_all too
    bf16* op = p.out + obase + (long long)qrow * p.o_seq;
#pragma unroll
    for (int i = 0; i < DT; ++i) {
      const int dv0 = i * 16 + 4 * g;
      if (dv0 < p.d) {
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16)(acc[i][e] * inv);
        nr_store8(op + dv0, o);
      }
    }
  }
}

template <int DK, int DT>
int launch_attn(const NrAttnParams& p, hipStream_t stream) {
  if (p.Lq >= 48 && p.inner == 1 && !p.causal) {   // causal masking lives in the per-wave kernel only
    // long sequences: block-shared K/V tiles
    constexpr int KSK = DK <= 2 ? 64 : DK * 32 + 8;                              // as in the kernel
    constexpr int KSV = (DT * 16) % 32 == 16 ? DT * 16 : DT * 16 + 16;
    const size_t shm = (size_t)2 * KT2 * (KSK + KSV) * sizeof(bf16);
    static unsigned long long attr_mask = 0;       // per device
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!(attr_mask >> (dev & 63) & 1ull)) {
      (void)hipFuncSetAttribute((const void*)attn_fwd_shared_kernel<DK, DT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
      (void)hipFuncSetAttribute((const void*)attn_fwd_shared_kernel<DK, DT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
      if constexpr (DK == 2 && DT == 3) {
        (void)hipFuncSetAttribute((const void*)attn_fwd_shared_kernel<DK, DT, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
        (void)hipFuncSetAttribute((const void*)attn_fwd_shared_kernel<DK, DT, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
      }
      attr_mask |= 1ull << (dev & 63);
    }
    const int qblocks = (p.Lq + 63) / 64;
    const unsigned blocks = (unsigned)((long long)p.nbatch * p.heads * qblocks);
    if constexpr (DK == 2 && DT == 3) {
      // d = 40: the denominator rides on the pad column of V (NR_ATTN_ROWSUM=adds keeps the VALU sums: A/B only)
      static const bool ones = !(getenv("NR_ATTN_ROWSUM") && getenv("NR_ATTN_ROWSUM")[0] == 'a');
      if (ones && p.d == 40) {
        if (p.fp8) hipLaunchKernelGGL((attn_fwd_shared_kernel<DK, DT, true, true>), dim3(blocks), dim3(256), shm, stream, p);
        else hipLaunchKernelGGL((attn_fwd_shared_kernel<DK, DT, false, true>), dim3(blocks), dim3(256), shm, stream, p);
        return 0;
      }
    }
    if (p.fp8) hipLaunchKernelGGL((attn_fwd_shared_kernel<DK, DT, true>), dim3(blocks), dim3(256), shm, stream, p);
    else hipLaunchKernelGGL((attn_fwd_shared_kernel<DK, DT, false>), dim3(blocks), dim3(256), shm, stream, p);
    return 0;
  }
  const int qtiles = (p.Lq + 15) / 16;
  const long long total = (long long)p.nbatch * p.heads * qtiles;
  const unsigned blocks = (unsigned)((total + 3) / 4);
  const size_t shm = (size_t)4 * KT * vt_stride(DT) * sizeof(bf16);
  hipLaunchKernelGGL((attn_fwd_kernel<DK, DT>), dim3(blocks), dim3(256), shm, stream, p);
  return 0;
}

}  // namespace

extern "C" int nr_launch_attention(const NrAttnParams* pp, hipStream_t stream) {
  const NrAttnParams& p = *pp;
  if (p.d % 8 != 0 || p.d > 160 || p.d <= 0) return 1;
  if (p.Lq <= 0 || p.Lk <= 0 || p.nbatch <= 0) return 2;
  const int DK = (p.d + 31) / 32, DT = (p.d + 15) / 16;
  // instantiate the head dims on the path (40, 80, 160) plus small ones used by reduced-width tests
  if (DK == 1 && DT == 1) return launch_attn<1, 1>(p, stream);       // d = 8, 16
  if (DK == 1 && DT == 2) return launch_attn<1, 2>(p, stream);       // d = 24, 32
  if (DK == 2 && DT == 3) return launch_attn<2, 3>(p, stream);       // d = 40, 48
  if (DK == 2 && DT == 4) return launch_attn<2, 4>(p, stream);       // d = 56, 64
  if (DK == 3 && DT == 5) return launch_attn<3, 5>(p, stream);       // d = 72, 80
  if (DK == 3 && DT == 6) return launch_attn<3, 6>(p, stream);       // d = 88, 96
  if (DK == 4 && DT == 8) return launch_attn<4, 8>(p, stream);       // d = 120, 128
  if (DK == 5 && DT == 10) return launch_attn<5, 10>(p, stream);     // d = 152, 160
  return 3;
}
