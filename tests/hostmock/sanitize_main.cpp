// Host-side sanitizer run of the C++ of the product that is not device code: the runtime of libnpm_hip.so (device binding,
// the caching pool, copies, events: csrc/npm_runtime.hip) and the whole exchange shim (csrc/npm_comm.cpp), compiled against
// the host-memory HIP / RCCL stand-ins of this directory with -fsanitize=address,undefined (tests/test_host_sanitizers.py).
// SURVEY.md section 5 "race detection / sanitizers: new, small".  Every check aborts with a message; the leak checker and the
// stand-ins' live-object counters catch what is not given back.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "npm_comm.h"
#include "npm_hip.h"
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#define CHECK(cond) do { if (!(cond)) { std::fprintf(stderr, "FAILED %s:%d: %s (last error: %s / %s)\n", __FILE__, __LINE__, #cond, \
                                                     npm_last_error(), npm_comm_last_error()); std::abort(); } } while (0)

static void runtime_and_pool() {
    void *p = nullptr;
    CHECK(npm_malloc(&p, 64) == NPM_E_NOT_INITIALIZED);                    // nothing before npm_init
    CHECK(npm_sync() == NPM_E_NOT_INITIALIZED);
    mock_hip_set_devices(0);
    CHECK(npm_init(0) == NPM_E_NO_DEVICE && std::strstr(npm_last_error(), "no HIP device"));
    mock_hip_set_devices(2);
    CHECK(npm_init(5) == NPM_E_BAD_ARGUMENT);
    CHECK(npm_init(1) == NPM_OK && npm_init(1) == NPM_OK);                 // idempotent for the same device
    CHECK(npm_init(0) == NPM_E_BAD_ARGUMENT);
    char name[64];
    CHECK(npm_device_name(name, sizeof(name)) == NPM_OK && std::strstr(name, "mock"));
    CHECK(npm_device_name(name, 0) == NPM_E_BAD_ARGUMENT);

    // size classes and recycling: a freed block of a class is what the next request of that class gets
    void *a = nullptr, *b = nullptr, *c = nullptr;
    CHECK(npm_malloc(&a, 1) == NPM_OK && npm_malloc(&b, 600) == NPM_OK && npm_malloc(&c, (3u << 20) + 1) == NPM_OK);
    size_t used = 0, reserved = 0;
    CHECK(npm_pool_stats(&used, &reserved) == NPM_OK && used == 512 + 1024 + (4u << 20) && reserved == used);
    std::memset(c, 0xab, (3u << 20) + 1);                                  // the whole request is addressable
    CHECK(npm_free(b) == NPM_OK && npm_free(b) == NPM_E_BAD_ARGUMENT);     // double free is reported, not executed
    void *b2 = nullptr;
    CHECK(npm_malloc(&b2, 1000) == NPM_OK && b2 == b);                     // same class (1024): recycled
    int on_stack;
    CHECK(npm_free(&on_stack) == NPM_E_BAD_ARGUMENT && npm_free(nullptr) == NPM_OK);

    // out of memory: the cache is dropped once and the allocation retried
    CHECK(npm_free(a) == NPM_OK);                                          // one cached block
    mock_hip_fail_next_mallocs(1);
    void *d = nullptr;
    CHECK(npm_malloc(&d, 5000) == NPM_OK);
    CHECK(npm_pool_stats(&used, &reserved) == NPM_OK && reserved == 1024 + (4u << 20) + 8192);    // the cached 512 went back
    mock_hip_fail_next_mallocs(2);
    void *e = nullptr;
    CHECK(npm_malloc(&e, 70000) != NPM_OK && std::strstr(npm_last_error(), "out of memory"));

    // copies and events
    std::vector<float> host(256, 1.5f), back(256, 0.f);
    CHECK(npm_h2d(b2, host.data(), 1000) == NPM_OK && npm_d2d(d, b2, 1000) == NPM_OK && npm_d2h(back.data(), d, 1000) == NPM_OK);
    CHECK(back[0] == 1.5f && back[249] == 1.5f);
    CHECK(npm_h2d(nullptr, host.data(), 4) == NPM_E_BAD_ARGUMENT && npm_h2d(nullptr, nullptr, 0) == NPM_OK);
    void *e0 = nullptr, *e1 = nullptr;
    float ms = -1.f;
    CHECK(npm_event_create(&e0) == NPM_OK && npm_event_create(&e1) == NPM_OK);
    CHECK(npm_event_elapsed_ms(e0, e1, &ms) != NPM_OK);                    // not recorded yet
    CHECK(npm_event_record(e0) == NPM_OK && npm_event_record(e1) == NPM_OK && npm_event_sync(e1) == NPM_OK);
    CHECK(npm_event_elapsed_ms(e0, e1, &ms) == NPM_OK && ms > 0.f);
    CHECK(npm_event_destroy(e0) == NPM_OK && npm_event_destroy(e1) == NPM_OK && npm_event_destroy(nullptr) == NPM_OK);

    // the pool is mutex-guarded: two host threads allocating and freeing (the library is driven by one thread per process;
    // Python's garbage collector may still free from another)
    std::vector<std::thread> workers;
    for (int t = 0; t < 4; ++t)
        workers.emplace_back([t] {
            for (int i = 0; i < 2000; ++i) {
                void *q = nullptr;
                if (npm_malloc(&q, 256u << (i % 6)) != NPM_OK) std::abort();
                std::memset(q, t, 256u << (i % 6));
                if (npm_free(q) != NPM_OK) std::abort();
            }
        });
    for (auto &w : workers) w.join();

    CHECK(npm_free(b2) == NPM_OK && npm_free(c) == NPM_OK && npm_free(d) == NPM_OK);
    CHECK(npm_pool_stats(&used, &reserved) == NPM_OK && used == 0 && reserved > 0);
    CHECK(npm_pool_trim() == NPM_OK && npm_pool_stats(&used, &reserved) == NPM_OK && reserved == 0);
    CHECK(mock_hip_live_allocations() == 0);
}

static void exchange() {
    char id[NPM_COMM_ID_BYTES];
    CHECK(npm_comm_allreduce_f32(nullptr, 4, NPM_REDUCE_SUM) != 0 && std::strstr(npm_comm_last_error(), "npm_comm_init"));
    CHECK(npm_comm_unique_id(id) == 0 && npm_comm_unique_id(nullptr) != 0);
    CHECK(npm_comm_init(id, 3, 2, npm_stream()) != 0);                     // rank out of range
    CHECK(npm_comm_init(id, 0, 1, npm_stream()) == 0 && npm_comm_init(id, 0, 1, npm_stream()) != 0);
    int rank = -1, n = -1;
    CHECK(npm_comm_rank(&rank, &n) == 0 && rank == 0 && n == 1);
    std::vector<float> bucket(4096, 2.f);
    CHECK(npm_comm_allreduce_f32(bucket.data(), 0, NPM_REDUCE_AVG) == 0 && npm_comm_allreduce_f32(nullptr, 8, NPM_REDUCE_AVG) != 0);
    CHECK(npm_comm_stats_enable(1) == 0);
    npm_comm_exchange_stats st;
    // more pending spans than the cap: the surplus is counted, not recorded (nobody reads the statistics meanwhile)
    for (int i = 0; i < 8200; ++i) CHECK(npm_comm_allreduce_f32(bucket.data(), bucket.size(), NPM_REDUCE_AVG) == 0);
    CHECK(npm_comm_wait() == 0);
    CHECK(npm_comm_stats(&st) == 0 && st.allreduce_calls == 8192 && st.dropped == 8 && st.waits == 1 && st.bytes == 8192ull * 4096 * 4);
    CHECK(st.allreduce_ms > 0 && st.last_allreduce_ms > 0 && st.exposed_ms > 0);
    // a step's pattern: four flushes, one wait; the statistics reset after each read and the events are recycled
    const long events_before = mock_hip_live_events();
    for (int step = 0; step < 3; ++step) {
        for (int f = 0; f < 4; ++f) CHECK(npm_comm_allreduce_f32(bucket.data() + 1024 * f, 1024, NPM_REDUCE_SUM) == 0);
        CHECK(npm_comm_wait() == 0);
        CHECK(npm_comm_stats(&st) == 0 && st.allreduce_calls == 4 && st.waits == 1 && st.dropped == 0 && st.bytes == 4ull * 1024 * 4);
    }
    CHECK(mock_hip_live_events() == events_before);                        // no event is created once the spare list is warm
    CHECK(npm_comm_stats(nullptr) != 0);
    CHECK(npm_comm_broadcast_f32(bucket.data(), 16, 0) == 0 && npm_comm_barrier() == 0);
    double v = 3.25;
    CHECK(npm_comm_allreduce_host_f64(&v, NPM_REDUCE_MAX) == 0 && v == 3.25 && npm_comm_allreduce_host_f64(nullptr, 0) != 0);
    CHECK(npm_comm_stats_enable(0) == 0);
    CHECK(npm_comm_destroy() == 0 && npm_comm_destroy() == 0);
    CHECK(mock_nccl_live_comms() == 0 && mock_hip_live_events() == 0);
}

int main() {
    runtime_and_pool();
    exchange();
    CHECK(npm_shutdown() == NPM_OK && npm_shutdown() == NPM_OK);
    CHECK(mock_hip_live_streams() == 0 && mock_hip_live_allocations() == 0);
    std::puts("host sanitizers: ok");
    return 0;
}
