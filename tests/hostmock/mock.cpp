// Implementation of the host-memory HIP / RCCL stand-ins (hip/hip_runtime.h, rccl/rccl.h in this directory).
#include <cstdlib>
#include <cstring>
#include <set>
#include <rccl/rccl.h>

struct mockStream { int id; };
struct mockEvent { long tick; bool recorded; };
struct mockComm { int rank, nranks; };
namespace {
int g_devices = 1, g_fail_mallocs = 0;
long g_tick = 0, g_streams = 0, g_events = 0, g_comms = 0;
std::set<void *> g_allocs;
hipError_t g_last = hipSuccess;
}
extern "C" {
void mock_hip_set_devices(int n) { g_devices = n; }
void mock_hip_fail_next_mallocs(int n) { g_fail_mallocs = n; }
long mock_hip_live_allocations(void) { return (long)g_allocs.size(); }
long mock_hip_live_events(void) { return g_events; }
long mock_hip_live_streams(void) { return g_streams; }
long mock_nccl_live_comms(void) { return g_comms; }
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : e == hipErrorOutOfMemory ? "out of memory" : e == hipErrorNoDevice ? "no HIP device" : "mock error"; }
hipError_t hipGetLastError(void) { hipError_t e = g_last; g_last = hipSuccess; return e; }
hipError_t hipGetDeviceCount(int *n) { *n = g_devices; return g_devices ? hipSuccess : hipErrorNoDevice; }
hipError_t hipSetDevice(int d) { return d >= 0 && d < g_devices ? hipSuccess : hipErrorInvalidValue; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) { std::memset(p, 0, sizeof(*p)); std::strcpy(p->name, "mock"); std::strcpy(p->gcnArchName, "host"); p->multiProcessorCount = 4; return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = new mockStream{(int)++g_streams}; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { delete s; --g_streams; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t e, unsigned) { return e ? hipSuccess : hipErrorInvalidValue; }
hipError_t hipMalloc(void **p, size_t n) {
    if (g_fail_mallocs > 0) { --g_fail_mallocs; *p = nullptr; return g_last = hipErrorOutOfMemory; }
    *p = std::malloc(n ? n : 1);
    g_allocs.insert(*p);
    return hipSuccess;
}
hipError_t hipFree(void *p) { if (!p) return hipSuccess; if (!g_allocs.erase(p)) return hipErrorInvalidValue; std::free(p); return hipSuccess; }
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { std::memmove(d, s, n); return hipSuccess; }
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { std::memset(d, v, n); return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t *e) { *e = new mockEvent{0, false}; ++g_events; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; --g_events; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->tick = ++g_tick; e->recorded = true; return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { if (!a->recorded || !b->recorded) return hipErrorInvalidValue; *ms = 0.001f * (float)(b->tick - a->tick); return hipSuccess; }
const char *ncclGetErrorString(ncclResult_t) { return "mock nccl error"; }
ncclResult_t ncclGetUniqueId(ncclUniqueId *id) { for (int i = 0; i < 128; ++i) id->internal[i] = (char)(37 * i + 11); return ncclSuccess; }
ncclResult_t ncclCommInitRank(ncclComm_t *c, int n, ncclUniqueId, int r) { if (r < 0 || r >= n) return ncclInvalidArgument; *c = new mockComm{r, n}; ++g_comms; return ncclSuccess; }
ncclResult_t ncclCommDestroy(ncclComm_t c) { delete c; --g_comms; return ncclSuccess; }
ncclResult_t ncclAllReduce(const void *s, void *d, size_t n, ncclDataType_t t, ncclRedOp_t, ncclComm_t c, hipStream_t) {
    if (!c) return ncclInvalidArgument;
    if (s != d) std::memmove(d, s, n * (t == ncclFloat64 ? 8 : 4));
    return ncclSuccess;
}
ncclResult_t ncclBroadcast(const void *s, void *d, size_t n, ncclDataType_t t, int, ncclComm_t c, hipStream_t) { return ncclAllReduce(s, d, n, t, ncclSum, c, nullptr); }
}
