// libnpm_rccl.so: the gradient all-reduce of the data-parallel hot path, on RCCL.
// xGMI is point-to-point (7 links per GPU), so the exchange is ONE flat bucket per
// sub-layer group rather than one call per parameter: 16 latency-bound calls would cost
// more than the 50 MB payload.  See include/npm_comm.h for the contract.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <dlfcn.h>
#include <unistd.h>
#include <utility>
#include <vector>

#include "npm_comm.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code ? code : -1;
}

// RCCL prints a version banner on STDOUT when it initialises.  The host program's stdout is a data channel
// (bench.py prints exactly one JSON line), so stdout points at stderr while RCCL initialises.
struct StdoutToStderr {
    int saved;
    StdoutToStderr() {
        fflush(stdout);
        saved = dup(STDOUT_FILENO);
        if (saved >= 0) dup2(STDERR_FILENO, STDOUT_FILENO);
    }
    ~StdoutToStderr() {
        fflush(stdout);
        if (saved >= 0) { dup2(saved, STDOUT_FILENO); close(saved); }
    }
};

struct Comm {
    bool ready = false;
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1;
    hipStream_t compute = nullptr;
    hipStream_t stream = nullptr;      // communication stream
    hipEvent_t produced = nullptr;     // compute -> comm
    hipEvent_t reduced = nullptr;      // comm -> compute
    double *scalar = nullptr;          // device scratch for host scalar reductions
    // exchange statistics (npm_comm_stats_*): timing-enabled event pairs, read back and recycled by npm_comm_stats
    bool stats = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> reduce_spans;   // on the communication stream, around each all-reduce
    std::vector<std::pair<hipEvent_t, hipEvent_t>> wait_spans;     // on the compute stream, around each wait
    std::vector<size_t> last_before_wait;                          // index into reduce_spans of the newest all-reduce at each wait
    std::vector<hipEvent_t> spare;
    unsigned long long bytes = 0;
    int calls = 0, waits = 0, dropped = 0;
} g;

// Spans wait for npm_comm_stats to read them; a caller that enables the statistics and never reads them must not grow
// the event pool without bound: beyond this many pending spans further ones are counted (`dropped`), not recorded.
constexpr size_t kMaxPendingSpans = 8192;

// every pending span back to the spare list (the streams are idle: their events have completed)
void recycle_spans() {
    for (auto &span : g.reduce_spans) { g.spare.push_back(span.first); g.spare.push_back(span.second); }
    for (auto &span : g.wait_spans) { g.spare.push_back(span.first); g.spare.push_back(span.second); }
    g.reduce_spans.clear();
    g.wait_spans.clear();
    g.last_before_wait.clear();
}

int timed_event(hipEvent_t *ev) {
    if (!g.spare.empty()) {
        *ev = g.spare.back();
        g.spare.pop_back();
        return 0;
    }
    return (int)hipEventCreate(ev);
}

#define HIPC(expr)                                                                  \
    do {                                                                            \
        hipError_t _e = (expr);                                                     \
        if (_e != hipSuccess) return fail((int)_e, "%s: %s -> %s", __func__, #expr, hipGetErrorString(_e)); \
    } while (0)

#define NCCLC(expr)                                                                 \
    do {                                                                            \
        ncclResult_t _r = (expr);                                                   \
        if (_r != ncclSuccess) return fail(20000 + (int)_r, "%s: %s -> %s", __func__, #expr, ncclGetErrorString(_r)); \
    } while (0)

#define REQUIRE_READY()                                                             \
    do {                                                                            \
        if (!g.ready) return fail(-2, "%s: npm_comm_init() has not been called", __func__); \
    } while (0)

ncclRedOp_t to_op(int op) {
    switch (op) {
        case NPM_REDUCE_AVG: return ncclAvg;
        case NPM_REDUCE_MAX: return ncclMax;
        default: return ncclSum;
    }
}

}  // namespace

extern "C" {

const char *npm_comm_last_error(void) { return g_err; }

int npm_comm_library_path(char *buf, int len) {
    if (!buf || len < 2) return fail(-1, "npm_comm_library_path: bad buffer");
    Dl_info info;
    if (!dladdr((void *)&ncclAllReduce, &info) || !info.dli_fname) return fail(-1, "npm_comm_library_path: dladdr failed");
    snprintf(buf, (size_t)len, "%s", info.dli_fname);
    return 0;
}

int npm_comm_unique_id(char *id) {
    if (!id) return fail(-1, "npm_comm_unique_id: null id");
    static_assert(sizeof(ncclUniqueId) == NPM_COMM_ID_BYTES, "id size");
    ncclUniqueId uid;
    {
        StdoutToStderr quiet;
        NCCLC(ncclGetUniqueId(&uid));
    }
    memcpy(id, &uid, sizeof(uid));
    return 0;
}

int npm_comm_init(const char *id, int rank, int nranks, void *compute_stream) {
    if (g.ready) return fail(-1, "npm_comm_init: already initialised");
    if (!id || nranks < 1 || rank < 0 || rank >= nranks) return fail(-1, "npm_comm_init: bad arguments");
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    {
        StdoutToStderr quiet;
        NCCLC(ncclCommInitRank(&g.comm, nranks, uid, rank));
    }
    g.rank = rank;
    g.nranks = nranks;
    g.compute = (hipStream_t)compute_stream;
    HIPC(hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking));
    HIPC(hipEventCreateWithFlags(&g.produced, hipEventDisableTiming));
    HIPC(hipEventCreateWithFlags(&g.reduced, hipEventDisableTiming));
    HIPC(hipMalloc((void **)&g.scalar, 64));
    g.ready = true;
    return 0;
}

int npm_comm_rank(int *rank, int *nranks) {
    REQUIRE_READY();
    if (rank) *rank = g.rank;
    if (nranks) *nranks = g.nranks;
    return 0;
}

int npm_comm_allreduce_f32(float *buf, size_t count, int op) {
    REQUIRE_READY();
    if (count == 0) return 0;
    if (!buf) return fail(-1, "npm_comm_allreduce_f32: null buffer");
    HIPC(hipEventRecord(g.produced, g.compute));          // gradients written so far ...
    HIPC(hipStreamWaitEvent(g.stream, g.produced, 0));    // ... are visible to the collective
    hipEvent_t t0 = nullptr, t1 = nullptr;
    const bool timed = g.stats && g.reduce_spans.size() < kMaxPendingSpans;
    if (g.stats && !timed) g.dropped += 1;
    if (timed) {
        HIPC((hipError_t)timed_event(&t0));
        HIPC((hipError_t)timed_event(&t1));
        HIPC(hipEventRecord(t0, g.stream));
    }
    NCCLC(ncclAllReduce(buf, buf, count, ncclFloat32, to_op(op), g.comm, g.stream));
    if (timed) {
        HIPC(hipEventRecord(t1, g.stream));
        g.reduce_spans.emplace_back(t0, t1);
        g.bytes += (unsigned long long)count * sizeof(float);
        g.calls += 1;
    }
    return 0;
}

int npm_comm_broadcast_f32(float *buf, size_t count, int root) {
    REQUIRE_READY();
    if (count == 0) return 0;
    if (!buf) return fail(-1, "npm_comm_broadcast_f32: null buffer");
    HIPC(hipEventRecord(g.produced, g.compute));
    HIPC(hipStreamWaitEvent(g.stream, g.produced, 0));
    NCCLC(ncclBroadcast(buf, buf, count, ncclFloat32, root, g.comm, g.stream));
    return 0;
}

int npm_comm_wait(void) {
    REQUIRE_READY();
    HIPC(hipEventRecord(g.reduced, g.stream));
    hipEvent_t t0 = nullptr, t1 = nullptr;
    const bool timed = g.stats && g.wait_spans.size() < kMaxPendingSpans;
    if (g.stats && !timed) g.dropped += 1;
    if (timed) {
        HIPC((hipError_t)timed_event(&t0));
        HIPC((hipError_t)timed_event(&t1));
        HIPC(hipEventRecord(t0, g.compute));              // the compute stream arrives here ...
    }
    HIPC(hipStreamWaitEvent(g.compute, g.reduced, 0));
    if (timed) {
        HIPC(hipEventRecord(t1, g.compute));              // ... and goes on here: the difference is EXPOSED exchange time
        g.wait_spans.emplace_back(t0, t1);
        g.waits += 1;
        // the newest all-reduce at this wait is the one nothing can overlap (a backward issues it after its last
        // gradient): its duration is reported separately (last_allreduce_ms)
        if (!g.reduce_spans.empty() && (g.last_before_wait.empty() || g.last_before_wait.back() != g.reduce_spans.size() - 1))
            g.last_before_wait.push_back(g.reduce_spans.size() - 1);
    }
    return 0;
}

int npm_comm_stats_enable(int on) {
    REQUIRE_READY();
    if (!on && g.stats) {                                 // nobody will read what is pending: give the events back
        HIPC(hipStreamSynchronize(g.stream));
        HIPC(hipStreamSynchronize(g.compute));
        recycle_spans();
        g.bytes = 0;
        g.calls = g.waits = g.dropped = 0;
    }
    g.stats = on != 0;
    return 0;
}

int npm_comm_stats(npm_comm_exchange_stats *out) {
    REQUIRE_READY();
    if (!out) return fail(-1, "npm_comm_stats: null result");
    HIPC(hipStreamSynchronize(g.stream));
    HIPC(hipStreamSynchronize(g.compute));
    double reduce_ms = 0, exposed_ms = 0, last_ms = 0;
    for (auto &span : g.reduce_spans) {
        float ms = 0;
        HIPC(hipEventElapsedTime(&ms, span.first, span.second));
        reduce_ms += ms;
    }
    for (size_t idx : g.last_before_wait) {
        float ms = 0;
        HIPC(hipEventElapsedTime(&ms, g.reduce_spans[idx].first, g.reduce_spans[idx].second));
        last_ms += ms;
    }
    for (auto &span : g.wait_spans) {
        float ms = 0;
        HIPC(hipEventElapsedTime(&ms, span.first, span.second));
        exposed_ms += ms;
    }
    out->bytes = g.bytes;
    out->allreduce_calls = g.calls;
    out->waits = g.waits;
    out->allreduce_ms = reduce_ms;
    out->exposed_ms = exposed_ms;
    out->last_allreduce_ms = last_ms;
    out->dropped = g.dropped;
    recycle_spans();
    g.bytes = 0;
    g.calls = g.waits = g.dropped = 0;
    return 0;
}

int npm_comm_barrier(void) {
    REQUIRE_READY();
    HIPC(hipStreamSynchronize(g.compute));
    HIPC(hipMemsetAsync(g.scalar, 0, 8, g.stream));
    NCCLC(ncclAllReduce(g.scalar, g.scalar, 1, ncclFloat64, ncclSum, g.comm, g.stream));
    // Poll with short sleeps instead of hipStreamSynchronize: a rank may wait here for a long time (bench.py: the other
    // ranks wait while rank 0 times the CPU baseline on the host cores) and must not spin on a core meanwhile.
    for (long polls = 0;; ++polls) {
        const hipError_t q = hipStreamQuery(g.stream);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) return fail((int)q, "npm_comm_barrier: hipStreamQuery -> %s", hipGetErrorString(q));
        if (polls > 4000) usleep(200);                    // the first millisecond or so spins: a timed region ends in this call
    }
    return 0;
}

int npm_comm_allreduce_host_f64(double *value, int op) {
    REQUIRE_READY();
    if (!value) return fail(-1, "npm_comm_allreduce_host_f64: null value");
    HIPC(hipMemcpyAsync(g.scalar, value, sizeof(double), hipMemcpyHostToDevice, g.stream));
    NCCLC(ncclAllReduce(g.scalar, g.scalar, 1, ncclFloat64, to_op(op), g.comm, g.stream));
    HIPC(hipMemcpyAsync(value, g.scalar, sizeof(double), hipMemcpyDeviceToHost, g.stream));
    HIPC(hipStreamSynchronize(g.stream));
    return 0;
}

int npm_comm_destroy(void) {
    if (!g.ready) return 0;
    (void)hipStreamSynchronize(g.stream);
    (void)ncclCommDestroy(g.comm);
    (void)hipFree(g.scalar);
    (void)hipEventDestroy(g.produced);
    (void)hipEventDestroy(g.reduced);
    for (auto &span : g.reduce_spans) { (void)hipEventDestroy(span.first); (void)hipEventDestroy(span.second); }
    for (auto &span : g.wait_spans) { (void)hipEventDestroy(span.first); (void)hipEventDestroy(span.second); }
    for (hipEvent_t ev : g.spare) (void)hipEventDestroy(ev);
    (void)hipStreamDestroy(g.stream);
    g = Comm();
    return 0;
}

}  // extern "C"
