// What a device-scope barrier costs on MI355X (VERDICT r5 next #2c: "the persistent per-block kernel with a device-scope barrier -- do it or record
// the number that kills it").  A persistent kernel that runs the ~8 dependent Linears / attentions of one transformer block of the keyframe model
// (M = 512 rows: every op's output columns are the next op's K, spread over all CUs) has to separate the ops by a grid-wide barrier: every
// workgroup's stores must be visible to every other workgroup on all 8 XCDs (whose L2s are not coherent with each other) before the next op reads.
// The alternative it competes with is a kernel boundary inside a replayed hipGraph: ~1.5-2.0 us between dependent launches (profiles/r05_launch_floor.txt).
//
// Arms (G workgroups of 256 threads, one per CU; NB barriers back to back; time per barrier = kernel time / NB):
//   0  arrive (atomicAdd, agent scope, release) + spin on an agent-scope acquire load of the counter + __syncthreads            -- the barrier alone
//   1  arm 0 with 16 KiB of payload per workgroup written before and the NEIGHBOUR's payload read (and checked) after the barrier   -- + visibility
//      (stores written back / reads invalidated: release / acquire fences at agent scope, i.e. buffer_wbl2 + buffer_inv around the barrier)
//   2  empty-kernel launches in a captured hipGraph of NB dependent nodes                                                     -- the boundary it replaces
//   build: hipcc --offload-arch=gfx950 -O3 -o grid_barrier grid_barrier.hip        run: ./grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
}

template <int PAYLOAD>
__global__ __launch_bounds__(256) void barrier_loop(unsigned* counter, unsigned* payload, int nb, unsigned* errors) {
  const unsigned G = gridDim.x;
  unsigned bad = 0;
  for (int it = 0; it < nb; ++it) {
    if (PAYLOAD) {            // 16 KiB per workgroup: 256 threads x 16 words
      unsigned* mine = payload + (size_t)blockIdx.x * 4096;
#pragma unroll
      for (int j = 0; j < 16; ++j) mine[j * 256 + threadIdx.x] = (unsigned)(it * 131071u + blockIdx.x * 17u + j);
      __atomic_thread_fence(__ATOMIC_RELEASE);   // hipcc: agent-scope release of this thread's stores (the arrive below is the workgroup's)
    }
    grid_barrier(counter, (unsigned)(it + 1) * G);
    if (PAYLOAD) {
      const unsigned nb_id = (blockIdx.x + 37u) % G;      // a workgroup on another XCD (ids are dealt round-robin over the 8 XCDs)
      const unsigned* theirs = payload + (size_t)nb_id * 4096;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const unsigned v = __hip_atomic_load(theirs + j * 256 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bad += v != (unsigned)(it * 131071u + nb_id * 17u + j);
      }
      // the next iteration overwrites the payload: nobody may still be reading it
      grid_barrier(counter + 32, (unsigned)(it + 1) * G);
    }
  }
  if (bad) atomicAdd(errors, bad);
}

__global__ void empty_kernel(unsigned* p) { if (p == nullptr) __builtin_trap(); }

int main() {
  int dev = 0;
  CK(hipSetDevice(dev));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, dev));
  const int cus = prop.multiProcessorCount;
  unsigned *counter, *payload, *errors;
  CK(hipMalloc(&counter, 4096));
  CK(hipMalloc(&payload, (size_t)1024 * 4096 * 4));
  CK(hipMalloc(&errors, 4));
  hipStream_t s;
  CK(hipStreamCreate(&s));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int NB = 2000;
  printf("MI355X device-scope barrier, %d CUs, %d barriers per launch, 256 threads per workgroup\n", cus, NB);
  for (int G : {64, 128, cus}) {
    for (int arm = 0; arm < 2; ++arm) {
      float best = 1e30f;
      unsigned herr = 0;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemsetAsync(counter, 0, 4096, s));
        CK(hipMemsetAsync(errors, 0, 4, s));
        CK(hipEventRecord(e0, s));
        if (arm == 0) hipLaunchKernelGGL(barrier_loop<0>, dim3(G), dim3(256), 0, s, counter, payload, NB, errors);
        else hipLaunchKernelGGL(barrier_loop<1>, dim3(G), dim3(256), 0, s, counter, payload, NB, errors);
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
        CK(hipMemcpy(&herr, errors, 4, hipMemcpyDeviceToHost));
      }
      const int per_it = arm == 0 ? 1 : 2;
      printf("  G = %3d workgroups, arm %d (%s): %.2f us per iteration = %.2f us per barrier%s\n", G, arm,
             arm == 0 ? "barrier only" : "16 KiB written / neighbour's 16 KiB read, 2 barriers", best * 1000.f / NB, best * 1000.f / NB / per_it,
             arm == 1 ? (herr ? "  STALE READS SEEN" : "  (all neighbour payloads correct)") : "");
    }
  }
  // arm 2: the kernel boundary it would replace, NB dependent empty launches in one graph
  {
    const int NL = 500;
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < NL; ++i) hipLaunchKernelGGL(empty_kernel, dim3(cus), dim3(256), 0, s, counter);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipEventRecord(e0, s));
      CK(hipGraphLaunch(ge, s));
      CK(hipEventRecord(e1, s));
      CK(hipStreamSynchronize(s));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      best = ms < best ? ms : best;
    }
    printf("  arm 2: %d dependent empty launches (%d workgroups) in a replayed hipGraph: %.2f us per launch boundary\n", NL, cus, best * 1000.f / NL);
  }
  return 0;
}
