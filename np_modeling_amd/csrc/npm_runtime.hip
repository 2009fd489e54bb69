// Runtime of libnpm_hip.so: device binding, the compute stream, a stream-ordered
// caching allocator over hipMalloc, copies and events.  One process drives one GPU.
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "npm_internal.h"

namespace npm {

static thread_local char g_error[512] = "";

static int g_last_math = 0;
void note_math(int mode) { g_last_math = mode; }
int last_math() { return g_last_math; }

// ---- K rendezvous counters (npm_mfma_tile.h ksync_wait) -------------------------------------------------------------
namespace {
constexpr int KSYNC_SLICES = 1024, KSYNC_SLICE_WORDS = 256;      // 1 KiB per launch: 8 groups x 32 words
unsigned *g_ksync_ring = nullptr;
int g_ksync_next = 0;
int g_ksync_every = 128;
}  // namespace

int ksync_every() { return g_ksync_every; }
void set_ksync_every(int every) { g_ksync_every = (every > 0 && (every & (every - 1)) == 0) ? every : 0; }

unsigned *ksync_slice() {
    if (g_ksync_every <= 0 || !ctx().ready) return nullptr;
    const size_t bytes = (size_t)KSYNC_SLICES * KSYNC_SLICE_WORDS * sizeof(unsigned);
    if (!g_ksync_ring) {
        if (hipMalloc((void **)&g_ksync_ring, bytes) != hipSuccess) { g_ksync_ring = nullptr; (void)hipGetLastError(); return nullptr; }
        g_ksync_next = KSYNC_SLICES;                             // zero the ring before its first use
    }
    if (g_ksync_next >= KSYNC_SLICES) {                          // every launch that used the ring is ahead of this memset in the stream
        if (hipMemsetAsync(g_ksync_ring, 0, bytes, ctx().stream) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        g_ksync_next = 0;
    }
    return g_ksync_ring + (size_t)(g_ksync_next++) * KSYNC_SLICE_WORDS;
}

void ksync_release() {
    if (g_ksync_ring) (void)hipFree(g_ksync_ring);
    g_ksync_ring = nullptr;
    g_ksync_next = 0;
}

Context &ctx() {
    static Context c;
    return c;
}

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
    return code ? code : NPM_E_BAD_ARGUMENT;
}

// ---- caching pool --------------------------------------------------------------
// Blocks are rounded to a size class and recycled by exact class.  Activations of a
// training step recur with identical sizes, so after the first step no hipMalloc /
// hipFree (which synchronises the device) is issued.  HBM3E is 288 GB: the pool never
// returns memory on its own (npm_pool_trim does).
struct Pool {
    std::mutex mu;
    std::unordered_map<void *, size_t> live;               // ptr -> class bytes
    std::map<size_t, std::vector<void *>> free_blocks;     // class bytes -> cached ptrs
    size_t in_use = 0, reserved = 0;

    static size_t size_class(size_t bytes) {
        if (bytes < 512) return 512;
        if (bytes <= (1u << 20)) {                         // next power of two up to 1 MiB
            size_t c = 512;
            while (c < bytes) c <<= 1;
            return c;
        }
        const size_t gran = 2u << 20;                      // 2 MiB granules above
        return (bytes + gran - 1) / gran * gran;
    }
};

static Pool &pool() {
    static Pool p;
    return p;
}

}  // namespace npm

using npm::ctx;

extern "C" {

int npm_abi_version(void) { return NPM_ABI_VERSION; }

const char *npm_last_error(void) { return npm::g_error; }

int npm_device_count(int *count) {
    NPM_ARG(count != nullptr);
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return npm::fail((int)e, "npm_device_count: %s", hipGetErrorString(e));
    }
    *count = n;
    return NPM_OK;
}

int npm_init(int device) {
    auto &c = ctx();
    if (c.ready) {
        if (c.device == device) return NPM_OK;
        return npm::fail(NPM_E_BAD_ARGUMENT, "npm_init: already bound to device %d", c.device);
    }
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0)
        return npm::fail(NPM_E_NO_DEVICE, "npm_init: no HIP device visible (%s)", hipGetErrorString(e));
    NPM_ARG(device >= 0 && device < n);
    NPM_HIP(hipSetDevice(device));
    NPM_HIP(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
    hipDeviceProp_t prop;
    NPM_HIP(hipGetDeviceProperties(&prop, device));
    c.num_cus = prop.multiProcessorCount;
    c.device = device;
    c.ready = true;
    return NPM_OK;
}

int npm_shutdown(void) {
    auto &c = ctx();
    if (!c.ready) return NPM_OK;
    NPM_HIP(hipStreamSynchronize(c.stream));
    npm_pool_trim();
    npm::ksync_release();
    NPM_HIP(hipStreamDestroy(c.stream));
    c = npm::Context();
    return NPM_OK;
}

int npm_device_name(char *buf, int len) {
    NPM_REQUIRE_INIT();
    NPM_ARG(buf != nullptr && len > 0);
    hipDeviceProp_t prop;
    NPM_HIP(hipGetDeviceProperties(&prop, ctx().device));
    snprintf(buf, len, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return NPM_OK;
}

void *npm_stream(void) { return (void *)ctx().stream; }

int npm_sync(void) {
    NPM_REQUIRE_INIT();
    NPM_HIP(hipStreamSynchronize(ctx().stream));
    return NPM_OK;
}

int npm_malloc(void **ptr, size_t bytes) {
    NPM_REQUIRE_INIT();
    NPM_ARG(ptr != nullptr);
    auto &p = npm::pool();
    const size_t cls = npm::Pool::size_class(bytes);
    std::lock_guard<std::mutex> lock(p.mu);
    auto it = p.free_blocks.find(cls);
    void *d = nullptr;
    if (it != p.free_blocks.end() && !it->second.empty()) {
        d = it->second.back();
        it->second.pop_back();
    } else {
        hipError_t e = hipMalloc(&d, cls);
        if (e != hipSuccess) {
            // Out of memory: drop the cache once and retry.
            for (auto &kv : p.free_blocks) {
                for (void *q : kv.second) { (void)hipFree(q); p.reserved -= kv.first; }
                kv.second.clear();
            }
            e = hipMalloc(&d, cls);
            if (e != hipSuccess)
                return npm::fail((int)e, "npm_malloc(%zu): %s", bytes, hipGetErrorString(e));
        }
        p.reserved += cls;
    }
    p.live[d] = cls;
    p.in_use += cls;
    *ptr = d;
    return NPM_OK;
}

int npm_free(void *ptr) {
    if (ptr == nullptr) return NPM_OK;
    auto &p = npm::pool();
    std::lock_guard<std::mutex> lock(p.mu);
    auto it = p.live.find(ptr);
    if (it == p.live.end()) return npm::fail(NPM_E_BAD_ARGUMENT, "npm_free: %p is not a live pool block", ptr);
    p.free_blocks[it->second].push_back(ptr);
    p.in_use -= it->second;
    p.live.erase(it);
    return NPM_OK;
}

int npm_pool_stats(size_t *bytes_in_use, size_t *bytes_reserved) {
    auto &p = npm::pool();
    std::lock_guard<std::mutex> lock(p.mu);
    if (bytes_in_use) *bytes_in_use = p.in_use;
    if (bytes_reserved) *bytes_reserved = p.reserved;
    return NPM_OK;
}

int npm_pool_trim(void) {
    auto &p = npm::pool();
    std::lock_guard<std::mutex> lock(p.mu);
    if (ctx().ready) (void)hipStreamSynchronize(ctx().stream);
    for (auto &kv : p.free_blocks) {
        for (void *q : kv.second) { (void)hipFree(q); p.reserved -= kv.first; }
        kv.second.clear();
    }
    return NPM_OK;
}

int npm_h2d(void *dst, const void *src, size_t bytes) {
    NPM_REQUIRE_INIT();
    if (bytes == 0) return NPM_OK;
    NPM_ARG(dst != nullptr && src != nullptr);
    NPM_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx().stream));
    NPM_HIP(hipStreamSynchronize(ctx().stream));
    return NPM_OK;
}

int npm_d2h(void *dst, const void *src, size_t bytes) {
    NPM_REQUIRE_INIT();
    if (bytes == 0) return NPM_OK;
    NPM_ARG(dst != nullptr && src != nullptr);
    NPM_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx().stream));
    NPM_HIP(hipStreamSynchronize(ctx().stream));
    return NPM_OK;
}

int npm_d2d(void *dst, const void *src, size_t bytes) {
    NPM_REQUIRE_INIT();
    if (bytes == 0) return NPM_OK;
    NPM_ARG(dst != nullptr && src != nullptr);
    NPM_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx().stream));
    return NPM_OK;
}

int npm_event_create(void **event) {
    NPM_REQUIRE_INIT();
    NPM_ARG(event != nullptr);
    hipEvent_t ev;
    NPM_HIP(hipEventCreate(&ev));
    *event = (void *)ev;
    return NPM_OK;
}

int npm_event_destroy(void *event) {
    if (event) NPM_HIP(hipEventDestroy((hipEvent_t)event));
    return NPM_OK;
}

int npm_event_record(void *event) {
    NPM_REQUIRE_INIT();
    NPM_ARG(event != nullptr);
    NPM_HIP(hipEventRecord((hipEvent_t)event, ctx().stream));
    return NPM_OK;
}

int npm_event_sync(void *event) {
    NPM_ARG(event != nullptr);
    NPM_HIP(hipEventSynchronize((hipEvent_t)event));
    return NPM_OK;
}

int npm_event_elapsed_ms(void *start, void *stop, float *ms) {
    NPM_ARG(start != nullptr && stop != nullptr && ms != nullptr);
    NPM_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return NPM_OK;
}

}  // extern "C"

namespace npm { int last_math(); }
extern "C" int npm_last_math(void) { return npm::last_math(); }
