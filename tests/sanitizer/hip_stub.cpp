// A stand-in for the HIP runtime, for the sanitizer job of the CPU suite only (tests/test_sanitizers.py): the HOST code of
// libpmc_hip.so (pmc_api.hip, pmc_ctx.hip: argument checks, pack building, workspace layout, scratch slots, stream and
// event bookkeeping, host-side conversions) is compiled with -fsanitize=address,undefined and linked against this file
// instead of libamdhip64.  "Device" memory is host heap (so every copy the host code issues is bounds-checked by
// AddressSanitizer), kernel launches do nothing, events measure nothing.  No numbers are checked here -- the GPU suite
// does that -- only that the host side runs clean.
#include <hip/hip_runtime_api.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <set>

namespace {
std::mutex g_mu;
std::set<void *> g_streams, g_events;
thread_local int t_device = 0;
}  // namespace

extern "C" {

hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int d) { if (d != 0) return hipErrorInvalidDevice; t_device = d; return hipSuccess; }
hipError_t hipGetDevice(int *d) { *d = t_device; return hipSuccess; }
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_tR0600 *p, int d)
{
    if (d != 0) return hipErrorInvalidDevice;
    std::memset(p, 0, sizeof(*p));
    std::strcpy(p->gcnArchName, "gfx950:stub");
    return hipSuccess;
}
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "stub HIP error"; }
hipError_t hipGetLastError(void) { return hipSuccess; }

hipError_t hipMalloc(void **p, size_t bytes) { *p = std::calloc(bytes ? bytes : 1, 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void *p) { std::free(p); return hipSuccess; }
hipError_t hipExtMallocWithFlags(void **p, size_t bytes, unsigned) { return hipMalloc(p, bytes); }
hipError_t hipHostMalloc(void **p, size_t bytes, unsigned) { return hipMalloc(p, bytes); }
hipError_t hipHostFree(void *p) { return hipFree(p); }
hipError_t hipHostGetDevicePointer(void **d, void *h, unsigned) { *d = h; return hipSuccess; }
hipError_t hipMallocAsync(void **p, size_t bytes, hipStream_t) { return hipMalloc(p, bytes); }
hipError_t hipFreeAsync(void *p, hipStream_t) { return hipFree(p); }
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t bytes, hipMemcpyKind, hipStream_t)
{
    std::memmove(dst, src, bytes);
    return hipSuccess;
}
hipError_t hipMemcpyPeerAsync(void *dst, int, const void *src, int, size_t bytes, hipStream_t)
{
    std::memmove(dst, src, bytes);
    return hipSuccess;
}
hipError_t hipDeviceCanAccessPeer(int *can, int a, int b) { *can = (a == 0 && b == 0); return hipSuccess; }
hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }
hipError_t hipDeviceGetPCIBusId(char *buf, int len, int d)
{
    if (d != 0 || len < 13) return hipErrorInvalidDevice;
    std::strcpy(buf, "0000:00:00.0");
    return hipSuccess;
}
hipError_t hipDeviceGetByPCIBusId(int *d, const char *id)
{
    if (std::strcmp(id, "0000:00:00.0") != 0) return hipErrorInvalidDevice;
    *d = 0;
    return hipSuccess;
}
hipError_t hipMemset(void *dst, int v, size_t bytes) { std::memset(dst, v, bytes); return hipSuccess; }
hipError_t hipMemsetAsync(void *dst, int v, size_t bytes, hipStream_t) { std::memset(dst, v, bytes); return hipSuccess; }

hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned)
{
    void *h = std::malloc(8);
    std::lock_guard<std::mutex> lock(g_mu);
    g_streams.insert(h);
    *s = (hipStream_t)h;
    return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t s)
{
    std::lock_guard<std::mutex> lock(g_mu);
    if (!g_streams.erase((void *)s)) return hipErrorInvalidHandle;       // double destroy / never created
    std::free((void *)s);
    return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
// IPC within one process: the handle carries the pointer
hipError_t hipIpcGetMemHandle(hipIpcMemHandle_t *h, void *p) { std::memset(h, 0, sizeof(*h)); std::memcpy(h, &p, sizeof(p)); return hipSuccess; }
hipError_t hipIpcOpenMemHandle(void **p, hipIpcMemHandle_t h, unsigned) { std::memcpy(p, &h, sizeof(*p)); return hipSuccess; }
hipError_t hipIpcCloseMemHandle(void *) { return hipSuccess; }

hipError_t hipEventCreate(hipEvent_t *e)
{
    void *h = std::malloc(8);
    std::lock_guard<std::mutex> lock(g_mu);
    g_events.insert(h);
    *e = (hipEvent_t)h;
    return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t)
{
    std::lock_guard<std::mutex> lock(g_mu);
    return g_events.count((void *)e) ? hipSuccess : hipErrorInvalidHandle;
}
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }

// kernel launches: nothing runs
hipError_t hipLaunchKernel(const void *, dim3, dim3, void **, size_t, hipStream_t) { return hipSuccess; }
hipError_t __hipPushCallConfiguration(dim3, dim3, size_t, hipStream_t) { return hipSuccess; }
hipError_t __hipPopCallConfiguration(dim3 *g, dim3 *b, size_t *shmem, hipStream_t *s)
{
    *g = dim3(1); *b = dim3(1); *shmem = 0; *s = nullptr;
    return hipSuccess;
}
void **__hipRegisterFatBinary(const void *) { static void *h = nullptr; return &h; }
void __hipRegisterFunction(void **, const void *, char *, const char *, unsigned, void *, void *, void *, void *, int *) {}
void __hipRegisterVar(void **, void *, char *, const char *, int, size_t, int, int) {}
void __hipUnregisterFatBinary(void **) {}

// (called when the process ends: handles the host code never gave back)
int pmc_stub_leaked_streams(void) { std::lock_guard<std::mutex> lock(g_mu); return (int)g_streams.size(); }

}  // extern "C"
