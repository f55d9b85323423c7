// Micro-benchmark behind DESIGN 3d ("where does the matrix pipe go before a single byte moves"): the k-loop skeleton of the tiled igemm
// (gemm.hip, 128x160 instantiation: 4 waves per workgroup, 2 workgroups per CU, 40 x v_mfma_f32_16x16x32_bf16 per wave and k-tile, ONE
// s_barrier per k-tile, 18 ds_read_b128 fragment reads per wave and k-tile) with the pieces switched on one at a time, on ZERO and on
// full-range RANDOM operands, each arm run for >= 1.5 s back to back so that the chip settles on the clock it holds under that load.
// Per arm: wall TFLOP/s, the in-kernel shader clock (delta s_memtime / delta s_memrealtime x 100 MHz, median over workgroups, stamped once
// around the loop: MI355X_MICROARCH.md "DVFS give-back" item 6) and cycles per MFMA per SIMD (16 = back-to-back issue).
//   build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form -o mfma_clock mfma_clock.hip      run: ./mfma_clock
//   (without -amdgpu-mfma-vgpr-form hipcc rotates the 20 accumulators of this loop through AGPRs: ~95 v_accvgpr_* copies per k-tile, 21.6
//   instead of 16.x cycles per MFMA -- not what gemm.hip compiles to: its hot loops carry no such copies; check with -save-temps)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// MODE bit 0: s_barrier per k-tile; bit 1: fragment reads from LDS per k-tile (else operands stay in registers)
template <int MODE>
__global__ __launch_bounds__(256) void kloop(const bf16x8* __restrict__ in, float* __restrict__ out, unsigned long long* __restrict__ stamps, int ktiles) {
  extern __shared__ __attribute__((aligned(16))) bf16x8 lds[];   // (128 + 160) rows x 64 k x 2 B = 36 KiB = 2304 x 16 B
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 2304; i += 256) lds[i] = in[(i * 7 + blockIdx.x) & 4095];
  __syncthreads();
  bf16x8 wf[2][5], xf[2][4];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
    for (int i = 0; i < 5; ++i) wf[ks][i] = in[(lane + 64 * (i + 5 * ks) + wave * 17) & 4095];
#pragma unroll
    for (int j = 0; j < 4; ++j) xf[ks][j] = in[(lane + 64 * (j + 4 * ks) + 1000 + wave * 29) & 4095];
  }
  f32x4 acc[5][4];
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fg = lane >> 4;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  for (int kt = 0; kt < ktiles; ++kt) {
    if constexpr (MODE & 1) __builtin_amdgcn_s_barrier();
    if constexpr (!(MODE & 2)) {             // operands stay in registers but are opaque per k-tile (same code shape as the loop that re-reads them)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int i = 0; i < 5; ++i) asm volatile("" : "+v"(wf[ks][i]));
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(xf[ks][j]));
      }
    }
    if constexpr (MODE & 2) {
      asm volatile("" ::: "memory");         // the fragment reads are re-issued every k-tile (not hoisted out of the loop)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int i = 0; i < 5; ++i) { const int row = (wave & 1) * 80 + i * 16 + fr; wf[ks][i] = lds[128 * 8 + row * 8 + ((4 * ks + fg) ^ (row & 7))]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int row = (wave >> 1) * 64 + j * 16 + fr; xf[ks][j] = lds[row * 8 + ((4 * ks + fg) ^ (row & 7))]; }
      }
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][i], xf[ks][j], acc[i][j], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
  out[(size_t)blockIdx.x * 256 + tid] = s;
  if (tid == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

typedef void (*kern_t)(const bf16x8*, float*, unsigned long long*, int);

int main() {
  const int nwg = 512, ktiles = 20000;                        // 2 workgroups per CU; 40 MFMAs per wave and k-tile
  bf16x8* in; float* out; unsigned long long* st;
  hipMalloc(&in, 4096 * 16); hipMalloc(&out, (size_t)nwg * 256 * 4); hipMalloc(&st, nwg * 16);
  std::vector<unsigned short> h(4096 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[4] = {"MFMA only (operands in registers)", "+ one s_barrier per k-tile", "+ 18 ds_read_b128 per k-tile, no barrier", "+ barrier + fragment reads"};
  kern_t ks[4] = {kloop<0>, kloop<1>, kloop<2>, kloop<3>};
  for (int k = 0; k < 4; ++k) hipFuncSetAttribute((const void*)ks[k], hipFuncAttributeMaxDynamicSharedMemorySize, 36864);
  printf("%-44s %-7s %9s %9s %10s %12s\n", "arm", "data", "TFLOP/s", "clock GHz", "cyc/MFMA", "frac 2.5 PF");
  for (int data = 0; data < 2; ++data) {
    srand(1);
    for (auto& v : h) {
      if (!data) v = 0;
      else { const float f = (rand() / (float)RAND_MAX) * 2.f - 1.f; unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
    }
    hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    for (int k = 0; k < 4; ++k) {
      float ms = 0.f; double total_ms = 0.0; int reps = 0;
      while (total_ms < 1500.0) {                             // settle the clock: >= 1.5 s of back-to-back launches
        hipEventRecord(e0);
        hipLaunchKernelGGL(ks[k], dim3(nwg), dim3(256), 36864, 0, in, out, st, ktiles);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); total_ms += ms; ++reps;
      }
      std::vector<unsigned long long> hs(nwg * 2);
      hipMemcpy(hs.data(), st, nwg * 16, hipMemcpyDeviceToHost);
      std::vector<double> clk, cyc;
      for (int b = 0; b < nwg; ++b) { clk.push_back((double)hs[2 * b] / (double)hs[2 * b + 1] * 0.1); cyc.push_back((double)hs[2 * b]); }
      std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
      const double flop = (double)nwg * 4 * 40.0 * ktiles * 16384.0;
      const double tf = flop / ms / 1e9;
      // each SIMD hosts 2 waves (one of each co-resident workgroup): MFMAs per SIMD = 2 x 40 x ktiles
      printf("%-44s %-7s %9.1f %9.3f %10.2f %12.3f\n", names[k], data ? "random" : "zero", tf, clk[nwg / 2], cyc[nwg / 2] / (2.0 * 40.0 * ktiles), tf / 2500.0);
    }
  }
  return 0;
}
