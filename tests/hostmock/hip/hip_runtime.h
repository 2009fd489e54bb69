// Host-memory stand-in for the HIP runtime API that the HOST halves of libnpm_hip.so / libnpm_rccl.so use
// (csrc/npm_runtime.hip, csrc/npm_comm.cpp).  Test infrastructure only (tests/test_host_sanitizers.py builds those two
// translation units against it with g++ -fsanitize=address,undefined and runs tests/hostmock/sanitize_main.cpp): streams
// execute immediately, "device" memory is malloc'd, events carry a tick counter.  Never part of the product.
#pragma once
#include <cstddef>
#include <cstdint>

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1, hipErrorNotReady = 600, hipErrorNoDevice = 100 };
typedef struct mockStream *hipStream_t;
typedef struct mockEvent *hipEvent_t;
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
struct hipDeviceProp_t { char name[256]; char gcnArchName[256]; int multiProcessorCount; };

extern "C" {
const char *hipGetErrorString(hipError_t);
hipError_t hipGetLastError(void);
hipError_t hipGetDeviceCount(int *);
hipError_t hipSetDevice(int);
hipError_t hipGetDeviceProperties(hipDeviceProp_t *, int);
hipError_t hipStreamCreateWithFlags(hipStream_t *, unsigned);
hipError_t hipStreamDestroy(hipStream_t);
hipError_t hipStreamSynchronize(hipStream_t);
hipError_t hipStreamQuery(hipStream_t);
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned);
hipError_t hipMalloc(void **, size_t);
hipError_t hipFree(void *);
hipError_t hipMemcpyAsync(void *, const void *, size_t, hipMemcpyKind, hipStream_t);
hipError_t hipMemsetAsync(void *, int, size_t, hipStream_t);
hipError_t hipEventCreate(hipEvent_t *);
hipError_t hipEventCreateWithFlags(hipEvent_t *, unsigned);
hipError_t hipEventDestroy(hipEvent_t);
hipError_t hipEventRecord(hipEvent_t, hipStream_t);
hipError_t hipEventSynchronize(hipEvent_t);
hipError_t hipEventElapsedTime(float *, hipEvent_t, hipEvent_t);
// test hooks
void mock_hip_set_devices(int n);
void mock_hip_fail_next_mallocs(int n);      // the next n hipMalloc calls report out of memory
long mock_hip_live_allocations(void);
long mock_hip_live_events(void);
long mock_hip_live_streams(void);
}
