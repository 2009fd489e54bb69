// Internal helpers shared by the translation units of libnpm_hip.so.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "npm_hip.h"

namespace npm {

struct Context {
    bool ready = false;
    int device = -1;
    hipStream_t stream = nullptr;
    int num_cus = 256;
};

Context &ctx();
void set_error(const char *fmt, ...);
int fail(int code, const char *fmt, ...);

// out[c] = sum_r x[r * ld + c], two-stage, fixed order (npm_rowops.hip)
int colsum_launch(const float *x, float *out, long rows, long cols, long ld);
void set_ln_bwd_blocks(int blocks_per_cu);
void set_ln_nt_split(int mode);
void set_ew_grid_cap(int blocks);
void set_stream_nt(int on);
// streaming tensors of at least 32 MB move with the nontemporal cache hint (npm_rowops.hip; NPM_TUNE_STREAM_NT)
bool stream_nt_enabled(size_t bytes);
// the arithmetic the most recent matrix-product launch actually ran (npm_last_math): NPM_MATH_*
void note_math(int mode);
// K rendezvous of co-resident split-K blocks (npm_mfma_tile.h ksync_wait): a zeroed 1 KiB slice of counters for ONE launch on
// the compute stream (a ring of slices, re-zeroed in stream order when it wraps), or null when the feature is off / unavailable
unsigned *ksync_slice();
int ksync_every();             // K tiles between rendezvous (NPM_TUNE_KSYNC; 0 = off)
void set_ksync_every(int every);
void ksync_release();

// Pool-backed scratch for split-K slabs and reduction partials; released on scope exit.
// Safe because every launch goes to the single compute stream (stream-ordered reuse).
struct Scratch {
    void *ptr = nullptr;
    int alloc(size_t bytes) { return npm_malloc(&ptr, bytes); }
    ~Scratch() { if (ptr) npm_free(ptr); }
};

}  // namespace npm

#define NPM_REQUIRE_INIT()                                                         \
    do {                                                                           \
        if (!npm::ctx().ready)                                                     \
            return npm::fail(NPM_E_NOT_INITIALIZED, "%s: npm_init() has not been called", __func__); \
    } while (0)

#define NPM_HIP(expr)                                                              \
    do {                                                                           \
        hipError_t _e = (expr);                                                    \
        if (_e != hipSuccess)                                                      \
            return npm::fail((int)_e, "%s: %s -> %s", __func__, #expr, hipGetErrorString(_e)); \
    } while (0)

#define NPM_CHECK_LAUNCH() NPM_HIP(hipGetLastError())

#define NPM_ARG(cond)                                                              \
    do {                                                                           \
        if (!(cond)) return npm::fail(NPM_E_BAD_ARGUMENT, "%s: bad argument: %s", __func__, #cond); \
    } while (0)
