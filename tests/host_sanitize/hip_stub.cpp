// CPU stand-in for the handful of HIP runtime entry points csrc/ calls, for the HOST-SIDE sanitizer build only
// (tests/host_sanitize/Makefile: every .hip translation unit compiled with --cuda-host-only
// -fsanitize=address,undefined and linked against THIS file instead of libamdhip64).  "Device" memory is malloc'd
// host memory, so every upload / clear / copy the plan builders issue is bounds-checked by AddressSanitizer
// against the allocation it targets; kernel launches are accepted and dropped (no device code exists in this build).
// Test infrastructure: never linked into libjarvis_hip.so.
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace {
struct CallCfg { dim3 grid, block; size_t shmem; hipStream_t stream; };
thread_local CallCfg g_cfg;
long g_launches = 0, g_rejected = 0;
thread_local hipError_t g_last = hipSuccess;
}  // namespace

extern "C" {
long jh_stub_launches() { return g_launches; }
long jh_stub_rejected() { return g_rejected; }

hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t) {
  for (size_t r = 0; r < h; ++r) memcpy((char*)d + r * dp, (const char*)s + r * sp, w);
  return hipSuccess;
}
hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipGetLastError() { const hipError_t e = g_last; g_last = hipSuccess; return e; }
const char* hipGetErrorString(hipError_t) { return "hip stub"; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_tR0600* p, int) {
  memset(p, 0, sizeof *p);
  p->multiProcessorCount = 256;
  strcpy(p->gcnArchName, "gfx950");
  return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (hipStream_t)malloc(8); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus* st) { *st = hipStreamCaptureStatusNone; return hipSuccess; }
hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return hipSuccess; }
hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { *g = (hipGraph_t)malloc(8); return hipSuccess; }
hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t, hipGraphNode_t*, char*, size_t) { *e = (hipGraphExec_t)malloc(8); return hipSuccess; }
hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipSuccess; }
hipError_t hipGraphExecDestroy(hipGraphExec_t e) { free(e); return hipSuccess; }
hipError_t hipGraphDestroy(hipGraph_t g) { free(g); return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = (hipEvent_t)malloc(8); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (hipEvent_t)malloc(8); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }

// kernel launch plumbing of the host stubs clang emits for `kernel<<<...>>>`: the configuration is checked where it is
// pushed (what the hardware would refuse is reported through hipGetLastError, which every launch site checks); the
// launch itself is accepted and dropped
hipError_t __hipPushCallConfiguration(dim3 grid, dim3 block, size_t shmem, hipStream_t stream) {
  g_cfg = CallCfg{grid, block, shmem, stream};
  const size_t threads = (size_t)block.x * block.y * block.z;
  if (!grid.x || !grid.y || !grid.z || !threads || threads > 1024 || shmem > 160 * 1024 || grid.y > 65535 ||
      grid.z > 65535 || (size_t)grid.x * grid.y * grid.z > ((size_t)1 << 32)) {
    ++g_rejected;
    fprintf(stderr, "hip stub: invalid launch configuration grid %u %u %u block %u %u %u lds %zu\n", grid.x, grid.y,
            grid.z, block.x, block.y, block.z, shmem);
    g_last = hipErrorInvalidConfiguration;
    return hipErrorInvalidConfiguration;          // (non-zero: the host stub skips the launch)
  }
  ++g_launches;
  if (getenv("JH_STUB_TRACE")) fprintf(stderr, "launch %u %u %u / %u lds %zu\n", grid.x, grid.y, grid.z, block.x, shmem);
  return hipSuccess;
}
hipError_t __hipPopCallConfiguration(dim3* grid, dim3* block, size_t* shmem, hipStream_t* stream) {
  *grid = g_cfg.grid; *block = g_cfg.block; *shmem = g_cfg.shmem; *stream = g_cfg.stream;
  return hipSuccess;
}
hipError_t hipLaunchKernel(const void*, dim3, dim3, void**, size_t, hipStream_t) { return hipSuccess; }
void** __hipRegisterFatBinary(const void*) { static void* h; return &h; }
void __hipUnregisterFatBinary(void**) {}
void __hipRegisterFunction(void**, const void*, char*, const char*, unsigned, void*, void*, void*, void*, int*) {}
void __hipRegisterVar(void**, void*, char*, const char*, int, size_t, int, int) {}
}
