// Host stand-in for the HIP runtime, for the sanitizer dry-run of the engine's HOST logic (tests/test_sanitize_host.py): device memory is
// host heap memory (so AddressSanitizer sees every weight-conversion upload, manifest offset and arena copy), kernel launches are no-ops
// (nothing on this path reads kernel results on the host), streams / events / graphs are counted dummy objects.  Test infrastructure
// only: never linked into libneurons_amd.so.
#include <hip/hip_runtime_api.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

static long g_live_allocs = 0, g_graphs = 0, g_captures = 0, g_launches = 0;
static bool g_capturing = false;

extern "C" {
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); ++g_live_allocs; return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void* p) { if (p) { free(p); --g_live_allocs; } return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "hip stub"; }
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = -1; return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (hipStream_t)malloc(8); return hipSuccess; }
hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = (hipStream_t)malloc(8); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = (hipEvent_t)malloc(8); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (hipEvent_t)malloc(8); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.001f; return hipSuccess; }
hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { g_capturing = true; ++g_captures; return hipSuccess; }
hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { g_capturing = false; *g = (hipGraph_t)malloc(8); return hipSuccess; }
hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t, hipGraphNode_t*, char*, size_t) { *e = (hipGraphExec_t)malloc(8); ++g_graphs; return hipSuccess; }
hipError_t hipGraphDestroy(hipGraph_t g) { free(g); return hipSuccess; }
hipError_t hipGraphExecDestroy(hipGraphExec_t e) { free(e); --g_graphs; return hipSuccess; }
hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipSuccess; }
hipError_t hipLaunchKernel(const void*, dim3, dim3, void**, size_t, hipStream_t) { ++g_launches; return hipSuccess; }
// registration / launch-configuration hooks the host side of every __global__ function references
struct CallCfg { dim3 g, b; size_t shm; hipStream_t s; };
static thread_local CallCfg g_cfg;
void** __hipRegisterFatBinary(const void*) { static void* h = nullptr; return &h; }
void __hipRegisterFunction(void**, const void*, char*, const char*, unsigned, void*, void*, void*, void*, int*) {}
void __hipRegisterVar(void**, void*, char*, const char*, int, size_t, int, int) {}
void __hipUnregisterFatBinary(void**) {}
hipError_t __hipPushCallConfiguration(dim3 g, dim3 b, size_t shm, hipStream_t s) { g_cfg = CallCfg{g, b, shm, s}; return hipSuccess; }
hipError_t __hipPopCallConfiguration(dim3* g, dim3* b, size_t* shm, hipStream_t* s) { *g = g_cfg.g; *b = g_cfg.b; *shm = g_cfg.shm; *s = g_cfg.s; return hipSuccess; }
// counters for the driver
long nr_stub_live_allocs(void) { return g_live_allocs; }
long nr_stub_live_graphs(void) { return g_graphs; }
long nr_stub_captures(void) { return g_captures; }
long nr_stub_launches(void) { return g_launches; }
}
