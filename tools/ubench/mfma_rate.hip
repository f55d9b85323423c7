// Micro-benchmark: issue rate of v_mfma_f32_16x16x32_bf16 / 32x32x16 on MI355X, one or two waves per SIMD, 8 independent accumulators.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NACC>
__global__ void k16(float* out, int iters, const bf16x8* in) {
  bf16x8 a = in[threadIdx.x & 63], b = in[64 + (threadIdx.x & 63)];
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 10; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k32(float* out, int iters, const bf16x8* in) {
  bf16x8 a = in[threadIdx.x & 63], b = in[64 + (threadIdx.x & 63)];
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 10; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  float* out; bf16x8* in;
  hipMalloc(&out, 256 * 1024 * 4 * 8); hipMalloc(&in, 4096);
  unsigned short h[1024]; for (int i = 0; i < 1024; ++i) h[i] = 0x3f80 + (i * 37 % 64);   // random-ish bf16 around 1
  hipMemcpy(in, h, 2048, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  for (int threads : {256, 512, 1024}) {
    for (int kind = 0; kind < 3; ++kind) {
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (kind == 0) hipLaunchKernelGGL(k16<8>, dim3(256), dim3(threads), 0, 0, out, iters, in);
        else if (kind == 1) hipLaunchKernelGGL(k16<1>, dim3(256), dim3(threads), 0, 0, out, iters * 8, in);
        else hipLaunchKernelGGL(k32, dim3(256), dim3(threads), 0, 0, out, iters, in);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double nm = (kind == 2 ? 40.0 : 80.0) * iters;          // MFMAs per wave
      const double flop = nm * (threads / 64) * 256 * 16384.0 * (kind == 2 ? 2 : 1);
      printf("threads/CU %4d %s: %.3f ms, %.1f ns per MFMA per wave, %.1f TFLOP/s\n", threads,
             kind == 0 ? "16x16x32 8acc" : (kind == 1 ? "16x16x32 1acc" : "32x32x16 4acc"), ms, ms * 1e6 / nm, flop / ms / 1e9);
    }
  }
  return 0;
}
