// Micro-benchmark: what one LDS-DMA piece (1 KiB per wave instruction: 8 rows x 128 B) costs the issuing wave and the CU on MI355X, by
// address form: (0) global_load_lds_dwordx4 with a 64-bit VGPR address, (2) buffer_load_dwordx4 ... offen lds (SRD + 32-bit offset).  Every wave issues P pieces, waits for them (vmcnt(0)), repeats; the source
// is an L2-resident matrix walked like a GEMM operand tile (row stride S bytes).  Reported: shader cycles per piece per wave and bytes per
// clock per CU, for 4 / 8 / 16 waves per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int MODE>
__device__ __forceinline__ void piece(const char* base, unsigned off, const u32x4& srd, unsigned lds) {
  if constexpr (MODE == 0) {
    const char* p = base + off;
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(p), "s"(lds) : "memory", "m0");
  } else if constexpr (MODE == 1) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(off), "s"(base), "s"(lds) : "memory", "m0");
  } else {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" : : "v"(off), "s"(srd), "s"(lds) : "memory", "m0");
  }
}

template <int MODE, int P>
__global__ __launch_bounds__(1024) void dma_kernel(const char* src, size_t src_bytes, int stride, int iters, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  const int lr = lane >> 3, lp = lane & 7;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(lptr_t)smem) + (unsigned)(wave * P * 1024);
  const char* base = (const char*)(((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)src >> 32)) << 32) |
                                   __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)src));
  u32x4 srd;
  srd[0] = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)src);
  srd[1] = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)src >> 32) & 0xffffu);
  srd[2] = __builtin_amdgcn_readfirstlane((unsigned)src_bytes);
  srd[3] = 0x00020000u;
  // rows: workgroup b, wave w, piece j -> rows ((b * nw + w) * P + j) * 8 + lr of an [R][stride] matrix (wrapping inside the buffer)
  const unsigned rows_total = (unsigned)(src_bytes / (size_t)stride);
  unsigned off[P];
#pragma unroll
  for (int j = 0; j < P; ++j) {
    const unsigned row = (((blockIdx.x * nw + wave) * P + j) * 8 + lr) % (rows_total - 8);
    off[j] = row * (unsigned)stride + lp * 16;
  }
  const int ksteps = stride / 128;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  int k = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < P; ++j) piece<MODE>(base, off[j] + (unsigned)k * 128u, srd, lds0 + (unsigned)(j * 1024));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    k = k + 1 == ksteps ? 0 : k + 1;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[blockIdx.x * nw + wave] = t1 - t0;
}

// Round 5 (first step): the weight operand of a small-M GEMM straight into REGISTERS, no LDS: P x global_load_dwordx4 per wave in
// flight, then vmcnt(0).  SHAPE 0 = MFMA-fragment-shaped loads from a row-major [N][K] matrix (lane = row fr, 16-byte k-group fg: one wave instruction
// touches 16 rows x 64 contiguous bytes); SHAPE 1 = the same bytes stored FRAGMENT-MAJOR (one wave instruction = 1 KiB contiguous, lane l at +16 l).
template <int SHAPE, int P>
__global__ __launch_bounds__(1024) void reg_kernel(const char* src, size_t src_bytes, int stride, int iters, unsigned long long* cyc, unsigned* sink) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const unsigned rows_total = (unsigned)(src_bytes / (size_t)stride);
  const char* ptr[P];
#pragma unroll
  for (int j = 0; j < P; ++j) {
    if constexpr (SHAPE == 0) {
      const unsigned row = (((blockIdx.x * nw + wave) * P + j) * 16 + fr) % (rows_total - 16);
      ptr[j] = src + (size_t)row * stride + fg * 16;
    } else {
      const size_t piece = ((size_t)(blockIdx.x * nw + wave) * P + j) * 1024 % (src_bytes - 65536);
      ptr[j] = src + piece + lane * 16;
    }
  }
  const int ksteps = SHAPE == 0 ? stride / 64 : 32;
  u32x4 acc = {0u, 0u, 0u, 0u};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  int k = 0;
  for (int it = 0; it < iters; ++it) {
    u32x4 v[P];
#pragma unroll
    for (int j = 0; j < P; ++j) v[j] = *(const u32x4*)(ptr[j] + (SHAPE == 0 ? k * 64 : k * (int)(P * 1024 * 4)) % (SHAPE == 0 ? stride : 32768));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < P; ++j) acc ^= v[j];
    k = k + 1 == ksteps ? 0 : k + 1;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[blockIdx.x * nw + wave] = t1 - t0;
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) sink[0] = 1;
}

template <int SHAPE, int P>
static void run_reg(const char* name, const char* src, size_t bytes, int stride, int waves, int wgs_per_cu, unsigned long long* dcyc, unsigned* sink) {
  const int iters = 200, nwg = 256 * wgs_per_cu;
  for (int rep = 0; rep < 2; ++rep)
    hipLaunchKernelGGL((reg_kernel<SHAPE, P>), dim3(nwg), dim3(64 * waves), 0, 0, src, bytes, stride, iters, dcyc, sink);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((reg_kernel<SHAPE, P>), dim3(nwg), dim3(64 * waves), 0, 0, src, bytes, stride, iters, dcyc, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h((size_t)nwg * waves);
  hipMemcpy(h.data(), dcyc, h.size() * 8, hipMemcpyDeviceToHost);
  double sum = 0; for (auto v : h) sum += (double)v;
  const double per_piece = sum / h.size() / (iters * P);
  printf("%-44s stride %5d  P=%2d  waves/WG %2d  WG/CU %d: %7.1f cycles per 1-KiB load per wave, %5.1f B/clk/CU, %5.2f TB/s chip (wall %.1f us)\n", name, stride, P,
         waves, wgs_per_cu, per_piece, (double)waves * wgs_per_cu * 1024.0 / per_piece, (double)nwg * waves * iters * P * 1024.0 / (ms * 1e-3) / 1e12, ms * 1e3);
}

template <int MODE, int P>
static void run(const char* name, const char* src, size_t bytes, int stride, int waves, int wgs_per_cu, unsigned long long* dcyc) {
  const int iters = 200, nwg = 256 * wgs_per_cu;
  const size_t shm = (size_t)waves * P * 1024;
  hipFuncSetAttribute((const void*)dma_kernel<MODE, P>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
  for (int rep = 0; rep < 2; ++rep)
    hipLaunchKernelGGL((dma_kernel<MODE, P>), dim3(nwg), dim3(64 * waves), shm, 0, src, bytes, stride, iters, dcyc);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((dma_kernel<MODE, P>), dim3(nwg), dim3(64 * waves), shm, 0, src, bytes, stride, iters, dcyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h((size_t)nwg * waves);
  hipMemcpy(h.data(), dcyc, h.size() * 8, hipMemcpyDeviceToHost);
  double sum = 0; for (auto v : h) sum += (double)v;
  const double cyc_wave = sum / h.size();
  const double per_piece = cyc_wave / (iters * P);
  const double bpc_cu = (double)waves * wgs_per_cu * 1024.0 / per_piece;      // bytes per shader clock per CU (all its waves issuing)
  const double tbs = (double)nwg * waves * iters * P * 1024.0 / (ms * 1e-3) / 1e12;
  fflush(stdout);
  printf("%-34s stride %5d  P=%d  waves/WG %2d  WG/CU %d: %7.1f cycles per piece per wave, %5.1f B/clk/CU, %5.2f TB/s chip (wall %.1f us)\n", name,
         stride, P, waves, wgs_per_cu, per_piece, bpc_cu, tbs, ms * 1e3);
}

int main(int argc, char** argv) {
  const int only = argc > 1 ? atoi(argv[1]) : -1;       // run one address form only
  const size_t bytes = 3u << 20;     // 3 MiB: stays in every XCD's 4 MiB L2
  char* src; unsigned long long* dcyc;
  hipMalloc(&src, bytes); hipMemset(src, 1, bytes);
  hipMalloc(&dcyc, 8 * 512 * 16);
  for (int stride : {640, 2560, 5760}) {
    for (int waves : {4, 8}) {
      for (int wpc : {1, 2}) {
        if (only < 0 || only == 0) run<0, 9>("global_load_lds 64-bit vaddr", src, bytes, stride, waves, wpc, dcyc);
        if (only < 0 || only == 2) run<2, 9>("buffer_load offen lds", src, bytes, stride, waves, wpc, dcyc);
      }
    }
  }
  if (only < 0 || only == 3) {
    unsigned* sink; hipMalloc(&sink, 64);
    for (int waves : {4, 8})
      for (int wpc : {1, 2}) {
        run_reg<0, 8>("global_load_dwordx4 -> VGPR, fragment-shaped", src, bytes, 2560, waves, wpc, dcyc, sink);
        run_reg<1, 8>("global_load_dwordx4 -> VGPR, 1 KiB contiguous", src, bytes, 2560, waves, wpc, dcyc, sink);
        run_reg<0, 16>("global_load_dwordx4 -> VGPR, fragment-shaped", src, bytes, 2560, waves, wpc, dcyc, sink);
        run_reg<1, 16>("global_load_dwordx4 -> VGPR, 1 KiB contiguous", src, bytes, 2560, waves, wpc, dcyc, sink);
      }
  }
  if (only < 0 || only == 0) run<0, 4>("global_load_lds 64-bit vaddr", src, bytes, 2560, 16, 1, dcyc);
  if (only < 0 || only == 2) run<2, 4>("buffer_load offen lds", src, bytes, 2560, 16, 1, dcyc);
  return 0;
}
