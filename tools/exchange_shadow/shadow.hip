// Stand-in for one rank's share of an 8-rank ring all-reduce, for tools/exchange_shadow.py: what the gradient exchange
// of np_modeling_amd/parallel.py costs the COMPUTE of a step when it has a real peer, measured on one GPU.
//
// With one rank ncclAllReduce moves nothing and holds no CU (profiles/r04_exchange_path_ab.log: +0.3 % of the step).  With
// eight, RCCL's kernel holds `channels` workgroups for as long as the wire needs, and every one of its 2 (R - 1) steps
// moves count / R floats through the local HBM: reduce-scatter steps read the chunk a neighbour wrote into the local
// receive buffer and the local chunk and write the sum, all-gather steps read a received chunk and write it into place.
// This kernel does exactly that local work -- on zero-filled "received" data, so the gradients keep their values --
// with `channels` workgroups of 256 threads that stay resident for the whole collective, each step not starting before
// the time a link of `busbw` would have delivered the previous ones (s_memrealtime, 100 MHz).  No xGMI traffic: the
// fabric side of the CUs is not loaded, only their issue slots, registers, LDS allocation and the local HBM.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

namespace {

struct Shadow {
    bool ready = false;
    hipStream_t compute = nullptr, stream = nullptr;
    hipEvent_t produced = nullptr, reduced = nullptr;
    float *recv = nullptr, *gathered = nullptr;       // zero-filled receive buffer, all-gather target
    size_t capacity = 0;                              // floats
    int channels = 16, ranks = 8, lds_bytes = 0;
    double busbw_gbs = 0;                             // 0: unpaced (as fast as the local HBM allows)
    // HIP-event spans: collectives on the shadow stream, waits on the compute stream
    hipEvent_t ev[4096];
    int nev = 0, ncoll = 0, nwait = 0;
    int coll_idx[1024], wait_idx[1024];
} g;

char g_err[256] = "";
int fail(const char *what, hipError_t e) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return -1;
}
#define HIPC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(#x, e_); } while (0)

__global__ void __launch_bounds__(256)
shadow_ring_kernel(float *__restrict__ buf, const float *__restrict__ recv, float *__restrict__ gathered, size_t count, int ranks,
                   unsigned long long step_ticks) {
    extern __shared__ float lds[];                    // only ALLOCATED (what RCCL's kernel would keep other blocks from using)
    if (threadIdx.x == 1023) lds[0] = 0.f;          // (never: 256 threads) keeps the allocation alive
    const size_t chunk = (count / ranks + 3) / 4 * 4;                 // floats per ring step
    const size_t per = (chunk / gridDim.x + 3) / 4 * 4;               // ... per channel
    const size_t lo = (size_t)blockIdx.x * per, hi = lo + per < chunk ? lo + per : chunk;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const int steps = 2 * (ranks - 1);
    for (int s = 0; s < steps; ++s) {
        if (step_ticks) {
            while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)s * step_ticks) __builtin_amdgcn_s_sleep(32);
        }
        const size_t c = (size_t)((blockIdx.x + s) % ranks) * chunk;  // the chunk this step works on
        for (size_t i = lo + 4 * threadIdx.x; i < hi; i += 4 * blockDim.x) {
            const size_t at = c + i;
            if (at + 4 > count) break;
            const float4 r = *reinterpret_cast<const float4 *>(recv + at);
            float4 v = *reinterpret_cast<const float4 *>(buf + at);
            if (s < ranks - 1) {                      // reduce-scatter: local chunk += received chunk
                v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
                *reinterpret_cast<float4 *>(buf + at) = v;
            } else {                                  // all-gather: a received chunk goes into place
                v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
                *reinterpret_cast<float4 *>(gathered + at) = v;
            }
        }
    }
}

}  // namespace

extern "C" {

const char *shadow_last_error(void) { return g_err; }

int shadow_init(void *compute_stream, size_t max_floats) {
    if (g.ready) return 0;
    g.compute = (hipStream_t)compute_stream;
    HIPC(hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking));
    HIPC(hipEventCreateWithFlags(&g.produced, hipEventDisableTiming));
    HIPC(hipEventCreateWithFlags(&g.reduced, hipEventDisableTiming));
    g.capacity = max_floats + 1024;
    HIPC(hipMalloc((void **)&g.recv, g.capacity * sizeof(float)));
    HIPC(hipMalloc((void **)&g.gathered, g.capacity * sizeof(float)));
    HIPC(hipMemset(g.recv, 0, g.capacity * sizeof(float)));
    for (auto &e : g.ev) HIPC(hipEventCreate(&e));
    HIPC(hipFuncSetAttribute((const void *)shadow_ring_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    g.ready = true;
    return 0;
}

int shadow_configure(int channels, double busbw_gbs, int lds_bytes, int ranks) {
    g.channels = channels > 0 ? channels : 16;
    g.busbw_gbs = busbw_gbs;
    g.lds_bytes = lds_bytes;
    g.ranks = ranks > 1 ? ranks : 8;
    return 0;
}

int shadow_allreduce(float *buf, size_t count) {
    if (!g.ready) { snprintf(g_err, sizeof(g_err), "shadow_init() has not been called"); return -1; }
    if (count == 0) return 0;
    if (count > g.capacity) { snprintf(g_err, sizeof(g_err), "shadow_allreduce: %zu floats > capacity %zu", count, g.capacity); return -1; }
    HIPC(hipEventRecord(g.produced, g.compute));
    HIPC(hipStreamWaitEvent(g.stream, g.produced, 0));
    const bool timed = g.nev + 2 <= 4096 && g.ncoll < 1024;
    if (timed) HIPC(hipEventRecord(g.ev[g.nev], g.stream));
    // wire time of the whole collective at `busbw`: 2 (R - 1) / R * bytes / busbw; one step is 1 / (2 (R - 1)) of it
    unsigned long long step_ticks = 0;
    if (g.busbw_gbs > 0) {
        const double total_s = 2.0 * (g.ranks - 1) / g.ranks * (double)count * 4.0 / (g.busbw_gbs * 1e9);
        step_ticks = (unsigned long long)(total_s / (2.0 * (g.ranks - 1)) * 1e8);       // s_memrealtime: 100 MHz
    }
    hipLaunchKernelGGL(shadow_ring_kernel, dim3(g.channels), dim3(256), (size_t)g.lds_bytes, g.stream, buf, (const float *)g.recv,
                       g.gathered, count, g.ranks, step_ticks);
    HIPC(hipGetLastError());
    if (timed) {
        HIPC(hipEventRecord(g.ev[g.nev + 1], g.stream));
        g.coll_idx[g.ncoll++] = g.nev;
        g.nev += 2;
    }
    return 0;
}

int shadow_wait(void) {
    if (!g.ready) return -1;
    HIPC(hipEventRecord(g.reduced, g.stream));
    const bool timed = g.nev + 2 <= 4096 && g.nwait < 1024;
    if (timed) HIPC(hipEventRecord(g.ev[g.nev], g.compute));
    HIPC(hipStreamWaitEvent(g.compute, g.reduced, 0));
    if (timed) {
        HIPC(hipEventRecord(g.ev[g.nev + 1], g.compute));
        g.wait_idx[g.nwait++] = g.nev;
        g.nev += 2;
    }
    return 0;
}

// milliseconds the collectives held the shadow stream / the compute stream stood waiting for them since the last call
int shadow_stats(double *collective_ms, double *exposed_ms, int *collectives) {
    if (!g.ready) return -1;
    HIPC(hipStreamSynchronize(g.stream));
    HIPC(hipStreamSynchronize(g.compute));
    double c = 0, w = 0;
    for (int i = 0; i < g.ncoll; ++i) { float ms = 0; HIPC(hipEventElapsedTime(&ms, g.ev[g.coll_idx[i]], g.ev[g.coll_idx[i] + 1])); c += ms; }
    for (int i = 0; i < g.nwait; ++i) { float ms = 0; HIPC(hipEventElapsedTime(&ms, g.ev[g.wait_idx[i]], g.ev[g.wait_idx[i] + 1])); w += ms; }
    if (collective_ms) *collective_ms = c;
    if (exposed_ms) *exposed_ms = w;
    if (collectives) *collectives = g.ncoll;
    g.nev = g.ncoll = g.nwait = 0;
    return 0;
}

}  // extern "C"
