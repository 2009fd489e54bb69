// Shared device code of the fp32-MFMA tile kernels (GEMM and implicit-GEMM convolution):
// 128x128 block tile, K step 32, 4 wavefronts as 2x2, 64x64 per wave = 2x2 MFMA 32x32 tiles.
// See npm_gemm.hip for the design notes.
#pragma once

#include "npm_internal.h"

namespace npm_tile {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128;
constexpr int BN = 128;
constexpr int BK = 32;
constexpr int KPITCH = 36;                  // floats; K-major LDS row pitch (conflict-free ds_read_b128)
constexpr int NTHREADS = 256;
constexpr int TILE_FLOATS = BM * KPITCH;    // >= BK * BM: one operand tile in either layout

// Output side of a launch: where C goes and what the epilogue does.
struct Epilogue {
    float *C;
    long ldc;
    float alpha;
    int flags;               // NPM_EPI_*
    const float *bias;
    const float *R;          // residual, same indexing as C
    long ldr;
    float *aux;              // pre-activation out (RELU_SAVE) / mask in (RELU_MASK)
    long ldaux;
    const float *rowvec;     // per-row term of NPM_EPI_SOFTMAX_BWD (already offset to this batch)
    float *ws;               // split-K slabs (raw accumulators), pitch N
    int buf_ok;              // every extent (C, residual, aux, slab) < 2^31 bytes: buffer-instruction epilogue
    float *cs;               // optional column-sum partials: one row of N per (tile row, wave row)
    long cs_wm;              // elements between the partial rows of wave rows wm = 0 and 1
    int prio;                // NPM_TUNE_GEMM_WAVE_PRIO: bit 0 raise the wave's issue priority in the prologue, bit 1 in the epilogue
    float *rowdot;           // NPM_EPI_ROWDOT: [N / 128][M] row dots of C with aux per 128-column block (zero on entry)
    float rowdot_scale;
    long rowdot_m;           // M: elements between the rows of two column blocks
};

// Wave issue priority (s_setprio).  Three of the four blocks of a CU are always inside their MFMA loops; the
// scalar/vector ALU work of a block's prologue and epilogue competes with their back-to-back MFMAs for issue.
__device__ __forceinline__ void prio_high(int on) { if (on) __builtin_amdgcn_s_setprio(3); }
__device__ __forceinline__ void prio_low(int on) { if (on) __builtin_amdgcn_s_setprio(0); }

// Bijective XCD-contiguous remap: blocks b and b+8 share an XCD (and its L2), so give each
// XCD one contiguous run of logical tiles.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// tile index -> (tm, tn), column-major inside groups of `group_m` tile rows.
__device__ __forceinline__ void tile_coords(int t, int tiles_m, int tiles_n, int group_m, int &tm, int &tn) {
    const int per_group = group_m * tiles_n;
    const int gid = t / per_group;
    const int first = gid * group_m;
    const int gsz = min(tiles_m - first, group_m);
    const int in_group = t - gid * per_group;
    tm = first + in_group % gsz;
    tn = in_group / gsz;
}

// Registers -> LDS for one 128 x 32 operand tile held as 4 float4 per thread.
//   K-major : thread holds rows (tid>>3) + 32 i, k = 4 (tid&7) .. +3      -> s[row][k], pitch 36
//   MN-major: thread holds k rows (tid>>5) + 8 i, mn = 4 (tid&31) .. +3   -> s[k][mn], pitch 128
template <bool KMAJ>
__device__ __forceinline__ void store_tile(float *__restrict__ s, int tid, const float4 (&r)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (KMAJ) {
            const int row = (tid >> 3) + 32 * i;
            *reinterpret_cast<float4 *>(s + row * KPITCH + (tid & 7) * 4) = r[i];
        } else {
            const int k = (tid >> 5) + 8 * i;
            *reinterpret_cast<float4 *>(s + k * BM + (tid & 31) * 4) = r[i];
        }
    }
}

// Fragment for k-group g (8 k values): element s is k = 8g + 4*half + s.
template <bool KMAJ>
__device__ __forceinline__ float4 read_frag(const float *__restrict__ s, int row, int g, int half) {
    if (KMAJ) {
        return *reinterpret_cast<const float4 *>(s + row * KPITCH + 8 * g + 4 * half);
    } else {
        const float *p = s + (8 * g + 4 * half) * BM + row;
        return make_float4(p[0], p[BM], p[2 * BM], p[3 * BM]);
    }
}

__device__ __forceinline__ void zero_acc(f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
}

// MFMAs of k-groups [G0, G1) of one K tile for one wave (16 per group).
template <bool A_KMAJ, bool B_KMAJ, int G0, int G1>
__device__ __forceinline__ void mma_groups(const float *__restrict__ sA, const float *__restrict__ sB,
                                           int arow, int brow, int half, f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int g = G0; g < G1; ++g) {
        const float4 a0 = read_frag<A_KMAJ>(sA, arow, g, half);
        const float4 a1 = read_frag<A_KMAJ>(sA, arow + 32, g, half);
        const float4 b0 = read_frag<B_KMAJ>(sB, brow, g, half);
        const float4 b1 = read_frag<B_KMAJ>(sB, brow + 32, g, half);
        const float av[2][4] = {{a0.x, a0.y, a0.z, a0.w}, {a1.x, a1.y, a1.z, a1.w}};
        const float bv[2][4] = {{b0.x, b0.y, b0.z, b0.w}, {b1.x, b1.y, b1.z, b1.w}};
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][s], bv[j][s], acc[i][j], 0, 0, 0);
    }
}

// The 64 MFMAs of one K tile for one wave.
template <bool A_KMAJ, bool B_KMAJ>
__device__ __forceinline__ void mma_tile(const float *__restrict__ sA, const float *__restrict__ sB,
                                         int arow, int brow, int half, f32x16 (&acc)[2][2]) {
    mma_groups<A_KMAJ, B_KMAJ, 0, BK / 8>(sA, sB, arow, brow, half, acc);
}

// C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 half.
// `raw` stores the accumulators into a split-K slab (pitch N); otherwise the full epilogue.
__device__ __forceinline__ void write_tile(const f32x16 (&acc)[2][2], const Epilogue &e, bool raw,
                                           int m0, int n0, int M, int N, int wm, int wn, int l32, int half) {
    if (raw) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = n0 + wn * 64 + j * 32 + l32;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (row < M && col < N) e.ws[(long)row * N + col] = acc[i][j][r];
                }
            }
        return;
    }
    const bool has_bias = (e.flags & NPM_EPI_BIAS) != 0;
    const bool has_res = (e.flags & NPM_EPI_RESIDUAL) != 0;
    const bool relu_save = (e.flags & NPM_EPI_RELU_SAVE) != 0;
    const bool relu_mask = (e.flags & NPM_EPI_RELU_MASK) != 0;
    const bool relu = (e.flags & NPM_EPI_RELU) != 0;
    const bool sm_bwd = (e.flags & NPM_EPI_SOFTMAX_BWD) != 0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + l32;
            const float bias = (has_bias && col < N) ? e.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (row < M && col < N) {
                    float v = e.alpha * acc[i][j][r] + bias;
                    if (sm_bwd) v = e.alpha * e.aux[(long)row * e.ldaux + col] * (acc[i][j][r] - e.rowvec[row]);
                    if (has_res) v += e.R[(long)row * e.ldr + col];
                    if (relu_save) {
                        e.aux[(long)row * e.ldaux + col] = v;
                        v = fmaxf(v, 0.f);
                    }
                    if (relu_mask) v = (e.aux[(long)row * e.ldaux + col] >= 0.f) ? v : 0.f;
                    if (relu) v = fmaxf(v, 0.f);
                    e.C[(long)row * e.ldc + col] = v;
                }
            }
        }
}

// Per-wave state handed from write_tile_buf to the read-modify-write epilogue below.
struct RmwArgs {
    __amdgpu_buffer_rsrc_t rc;
    int vc[2];
    float bias[2];
    int ldc, row0, rows_here, flags, m0, n0, N, wn, l32, half;
    float alpha;
    float *cptr;
};

// F >= 0: the NPM_EPI_* flags (bit 6 = alpha is 1) are compile-time constants; F < 0: read them at run time.
// Per 32x32 MFMA tile: issue its 16 (or 32) loads together, wait once, then compute and store.
template <int F, bool CS>
__device__ __forceinline__ void rmw_epilogue(const f32x16 (&acc)[2][2], const Epilogue &e, const RmwArgs &a, float (&csum)[2]) {
    constexpr int OOB = 0x7FFFFFFF;
    const int flags = F >= 0 ? F : a.flags;
    const bool has_bias = F >= 0 ? (F & NPM_EPI_BIAS) != 0 : true;        // run time: bias[] is 0 when absent
    const bool has_res = (flags & NPM_EPI_RESIDUAL) != 0;
    const bool relu_save = (flags & NPM_EPI_RELU_SAVE) != 0;
    const bool relu_mask = (flags & NPM_EPI_RELU_MASK) != 0;
    const bool relu = (flags & NPM_EPI_RELU) != 0;
    const bool sm_bwd = (flags & NPM_EPI_SOFTMAX_BWD) != 0;
    const bool alpha_one = F >= 0 && (F & 64) != 0;
    const float alpha = a.alpha;
    const int ldc = a.ldc, row0 = a.row0, half = a.half, rows_here = a.rows_here;
    const int ldr = has_res ? (int)e.ldr : 0;
    const bool use_aux = relu_save || relu_mask || sm_bwd;
    const int ldx = use_aux ? (int)e.ldaux : 0;
    const auto rv = __builtin_amdgcn_make_buffer_rsrc((void *)(sm_bwd ? e.rowvec + a.m0 : a.cptr), 0, sm_bwd ? rows_here * 4 : 0, 0x00020000);
    const auto rr = __builtin_amdgcn_make_buffer_rsrc((void *)(has_res ? e.R + (long)a.m0 * ldr : a.cptr), 0,
                                                      has_res ? (int)(((long)(rows_here - 1) * ldr + a.N) * 4) : 0, 0x00020000);
    const auto rx = __builtin_amdgcn_make_buffer_rsrc((void *)(use_aux ? e.aux + (long)a.m0 * ldx : a.cptr), 0,
                                                      use_aux ? (int)(((long)(rows_here - 1) * ldx + a.N) * 4) : 0, 0x00020000);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = a.n0 + a.wn * 64 + j * 32 + a.l32;
        const int vr = col < a.N ? (4 * half * ldr + col) * 4 : OOB;
        const int vx = col < a.N ? (4 * half * ldx + col) * 4 : OOB;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float res[16], msk[16];
            if (has_res) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row0 + i * 32 + (r & 3) + 8 * (r >> 2);
                    res[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rr, vr, row * ldr * 4, 0));
                }
            }
            if (relu_mask || sm_bwd) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row0 + i * 32 + (r & 3) + 8 * (r >> 2);
                    msk[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, vx, row * ldx * 4, 0));
                }
            }
            if (sm_bwd) {   // res[] doubles as the per-row term (a residual is not combined with this mode)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row0 + i * 32 + (r & 3) + 8 * (r >> 2);
                    res[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rv, 16 * half, row * 4, 0));
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + i * 32 + (r & 3) + 8 * (r >> 2);
                float v = alpha_one ? acc[i][j][r] : alpha * acc[i][j][r];
                if (has_bias) v += a.bias[j];
                if (sm_bwd) v = alpha * msk[r] * (acc[i][j][r] - res[r]);
                else if (has_res) v += res[r];
                if (relu_save) {
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rx, vx, row * ldx * 4, 0);
                    v = fmaxf(v, 0.f);
                }
                if (relu_mask) v = msk[r] >= 0.f ? v : 0.f;
                if (relu) v = fmaxf(v, 0.f);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), a.rc, a.vc[j], row * ldc * 4, 0);
                if (CS && row + 4 * half < rows_here) csum[j] += v;
            }
        }
    }
}

// NPM_EPI_ROWDOT: store the accumulators as they are and add, per row, scale * sum_j acc[row, j] * aux[row, j] over this wave's 64
// columns to rowdot[column block][row] (the other wave of the tile row adds its 64: two commutative float atomics onto zero).
// Lane layout of a 32 x 32 MFMA result: column l32 on the lane, rows (r & 3) + 8 (r >> 2) + 4 half in registers r.  Per row
// tile i the 16 per-lane products (both column tiles added) are summed over the 32 lanes of a half: one exchange with lane ^ 16
// that adds all 16, then a transposing butterfly (step m = 8, 4, 2, 1: a lane keeps the half of its values whose index has bit
// m equal to its own lane bit and sends the other half to lane ^ m), after which lane l32 holds the sum of register
// r = l32 & 15.  31 exchanges per row tile, all ds_bpermute (LDS pipe): beside three co-resident blocks of back-to-back MFMAs
// it is VECTOR-ALU instructions that cost.  One row tile at a time: 16 sums + 16 aux values live beside the 64 accumulators
// (the kernel keeps its 128 registers = four blocks per CU).
__device__ __forceinline__ void rowdot_epilogue(const f32x16 (&acc)[2][2], const Epilogue &e, __amdgpu_buffer_rsrc_t rc, const int (&vc)[2],
                                                int ldc, int row0, int rows_here, int m0, int n0, int N, int wn, int l32, int half) {
    constexpr int OOB = 0x7FFFFFFF;
    const int ldx = (int)e.ldaux;
    const auto rx = __builtin_amdgcn_make_buffer_rsrc((void *)(e.aux + (long)m0 * ldx), 0, (int)(((long)(rows_here - 1) * ldx + N) * 4), 0x00020000);
    const int lane = 32 * half + l32;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float v[16];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + l32;
            const int vx = col < N ? (4 * half * ldx + col) * 4 : OOB;
#pragma unroll
            for (int q = 0; q < 2; ++q) {            // eight aux values at a time: the register file is full
                float x[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const int r = 8 * q + t;
                    const int row = row0 + i * 32 + (r & 3) + 8 * (r >> 2);
                    x[t] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, vx, row * ldx * 4, 0));
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[i][j][r]), rc, vc[j], row * ldc * 4, 0);
                }
#pragma unroll
                for (int t = 0; t < 8; ++t) v[8 * q + t] = j == 0 ? acc[i][0][8 * q + t] * x[t] : fmaf(acc[i][1][8 * q + t], x[t], v[8 * q + t]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r)
            v[r] += __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lane ^ 16) << 2, __builtin_bit_cast(int, v[r])));
#pragma unroll
        for (int m = 8, n = 16; m >= 1; m >>= 1, n >>= 1) {
            const bool upper = (l32 & m) != 0;
#pragma unroll
            for (int k = 0; k < n / 2; ++k) {
                const float keep = upper ? v[k + n / 2] : v[k], send = upper ? v[k] : v[k + n / 2];
                v[k] = keep + __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lane ^ m) << 2, __builtin_bit_cast(int, send)));
            }
        }
        const int r = l32 & 15;
        const int row = row0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (l32 < 16 && row < rows_here) atomicAdd(e.rowdot + (long)(n0 / 128) * e.rowdot_m + m0 + row, e.rowdot_scale * v[0]);
    }
}

// Branch-free epilogue through buffer instructions.  Each store is ONE instruction:
// the per-lane column offset sits in the VGPR offset (computed once), the row offset is a
// scalar (soffset) and rows >= M / columns >= N are dropped by the descriptor's range check
// (out-of-range lanes carry an offset beyond num_records) -- no per-element address VALU,
// no exec-mask juggling.  Requires every extent below 2^31 bytes (the host checks).
template <bool WITH_COLSUM = false, bool ROWDOT = false>
__device__ __forceinline__ void write_tile_buf(const f32x16 (&acc)[2][2], const Epilogue &e, bool raw,
                                               int m0, int n0, int M, int N, int wm, int wn, int l32, int half) {
    constexpr int OOB = 0x7FFFFFFF;
    // Descriptors are rebased to this block's first row, so offsets stay below 256 rows x pitch whatever the
    // size of C (a [131072, 4096] fp32 output is exactly 2^31 bytes); rows_here bounds the range check.
    const int ldc = raw ? N : (int)e.ldc;
    const int rows_here = min(M - m0, 256);
    float *cptr = (raw ? e.ws : e.C) + (long)m0 * ldc;
    const int flags = raw ? 0 : e.flags;
    const float alpha = raw ? 1.f : e.alpha;
    const auto rc = __builtin_amdgcn_make_buffer_rsrc((void *)cptr, 0, (int)(((long)(rows_here - 1) * ldc + N) * 4), 0x00020000);
    const bool has_bias = (flags & NPM_EPI_BIAS) != 0;
    const bool has_res = (flags & NPM_EPI_RESIDUAL) != 0;
    const bool relu_save = (flags & NPM_EPI_RELU_SAVE) != 0;
    const bool relu_mask = (flags & NPM_EPI_RELU_MASK) != 0;
    const bool relu = (flags & NPM_EPI_RELU) != 0;
    const bool sm_bwd = (flags & NPM_EPI_SOFTMAX_BWD) != 0;
    const bool want_cs = WITH_COLSUM && !raw && e.cs != nullptr;
    int vc[2];
    float bias[2], csum[2] = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + l32;
        vc[j] = col < N ? (4 * half * ldc + col) * 4 : OOB;
        bias[j] = (has_bias && col < N) ? e.bias[col] : 0.f;
    }
    const int row0 = wm * 64;                          // wave-uniform, relative to the block's first row

    if constexpr (ROWDOT) {        // its own kernel instantiation: the other epilogues carry none of this
        rowdot_epilogue(acc, e, rc, vc, ldc, row0, rows_here, m0, n0, N, wn, l32, half);
        return;
    }
    if (!has_res && !relu_save && !relu_mask && !sm_bwd) {
        // Store-only epilogues (plain, bias, relu): 64 stores back to back, nothing to wait for.  Vector-ALU
        // instructions of an epilogue issue slowly beside three blocks of back-to-back MFMAs (measured: 64 raw
        // stores 5 us, with 3 VALU each 25 us), so the common cases run specialised code with 0 or 1 per store.
        const bool simple = !relu && !(WITH_COLSUM && want_cs);
#define NPM_STORE_ONLY(EXPR)                                                                           \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                  \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                               \
            const int sc = (row0 + i * 32 + (r & 3) + 8 * (r >> 2)) * ldc * 4;                         \
            _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                            \
                const float a = acc[i][j][r];                                                          \
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(EXPR), rc, vc[j], sc, 0);        \
            }                                                                                          \
        }
        if (simple && !has_bias && alpha == 1.f) {
            NPM_STORE_ONLY(a)
        } else if (relu && !has_bias && alpha == 1.f && !(WITH_COLSUM && want_cs)) {      // (Conv2D: the bias rode in the accumulators)
            NPM_STORE_ONLY(fmaxf(a, 0.f))
        } else if (simple && !has_bias) {
            NPM_STORE_ONLY(alpha * a)
        } else if (simple && alpha == 1.f) {
            NPM_STORE_ONLY(a + bias[j])
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row0 + i * 32 + (r & 3) + 8 * (r >> 2);
                    const int sc = row * ldc * 4;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        float v = alpha * acc[i][j][r] + bias[j];
                        if (relu) v = fmaxf(v, 0.f);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rc, vc[j], sc, 0);
                        if (WITH_COLSUM && want_cs && row + 4 * half < rows_here) csum[j] += v;
                    }
                }
        }
#undef NPM_STORE_ONLY
    } else {
        // Epilogues that READ (residual / ReLU mask / softmax-backward operand) or write twice (ReLU with saved
        // pre-activation).  The flag combinations the layers use run code specialised at compile time (no
        // per-element flag tests, the minimum of vector-ALU work); anything else takes the generic instance.
        RmwArgs a;
        a.rc = rc; a.vc[0] = vc[0]; a.vc[1] = vc[1]; a.bias[0] = bias[0]; a.bias[1] = bias[1];
        a.ldc = ldc; a.row0 = row0; a.rows_here = rows_here; a.alpha = alpha; a.flags = flags;
        a.m0 = m0; a.n0 = n0; a.N = N; a.wn = wn; a.l32 = l32; a.half = half; a.cptr = cptr;
        const bool cs = WITH_COLSUM && want_cs;
        const int key = flags | (alpha == 1.f ? 64 : 0);
        if (key == (NPM_EPI_BIAS | NPM_EPI_RESIDUAL | 64)) rmw_epilogue<NPM_EPI_BIAS | NPM_EPI_RESIDUAL | 64, false>(acc, e, a, csum);
        else if (key == (NPM_EPI_RESIDUAL | 64) && !cs) rmw_epilogue<NPM_EPI_RESIDUAL | 64, false>(acc, e, a, csum);
        else if (key == (NPM_EPI_BIAS | NPM_EPI_RELU_SAVE | 64)) rmw_epilogue<NPM_EPI_BIAS | NPM_EPI_RELU_SAVE | 64, false>(acc, e, a, csum);
        else if (key == (NPM_EPI_RELU_SAVE | 64) && !cs) rmw_epilogue<NPM_EPI_RELU_SAVE | 64, false>(acc, e, a, csum);
        else if (key == (NPM_EPI_RELU_MASK | 64)) { if (cs) rmw_epilogue<NPM_EPI_RELU_MASK | 64, WITH_COLSUM>(acc, e, a, csum);
                                                    else rmw_epilogue<NPM_EPI_RELU_MASK | 64, false>(acc, e, a, csum); }
        else if (key == (NPM_EPI_RELU_MASK | NPM_EPI_RESIDUAL | 64) && !cs) rmw_epilogue<NPM_EPI_RELU_MASK | NPM_EPI_RESIDUAL | 64, false>(acc, e, a, csum);
        else if ((key & ~64) == NPM_EPI_SOFTMAX_BWD && !cs) rmw_epilogue<NPM_EPI_SOFTMAX_BWD, false>(acc, e, a, csum);
        else { if (cs) rmw_epilogue<-1, WITH_COLSUM>(acc, e, a, csum); else rmw_epilogue<-1, false>(acc, e, a, csum); }
    }
    if (WITH_COLSUM && want_cs) {   // this wave's 64 rows summed per column: rows live in the registers and the two lane halves
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float total = csum[j] + __shfl_xor(csum[j], 32, 64);
            const int col = n0 + wn * 64 + j * 32 + l32;
            if (half == 0 && col < N) e.cs[wm * e.cs_wm + col] = total;
        }
    }
}

// Dense operand tile: global -> registers (4 float4 per thread), zero-filled outside
// [0, MN) x [0, kend).  VEC needs 16-byte aligned rows and MN / K multiples of 4.
template <bool KMAJ, bool VEC>
__device__ __forceinline__ void load_tile(const float *__restrict__ base, long ld, int mn0, int MN,
                                          int k0, int kend, int tid, float4 (&r)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int mn, k;
        long off;
        if (KMAJ) {
            mn = mn0 + (tid >> 3) + 32 * i;
            k = k0 + (tid & 7) * 4;
            off = (long)mn * ld + k;
        } else {
            k = k0 + (tid >> 5) + 8 * i;
            mn = mn0 + (tid & 31) * 4;
            off = (long)k * ld + mn;
        }
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (VEC) {
            if (mn < MN && k < kend) v = *reinterpret_cast<const float4 *>(base + off);
        } else if (KMAJ) {
            if (mn < MN) {
                if (k + 0 < kend) v.x = base[off + 0];
                if (k + 1 < kend) v.y = base[off + 1];
                if (k + 2 < kend) v.z = base[off + 2];
                if (k + 3 < kend) v.w = base[off + 3];
            }
        } else {
            if (k < kend) {
                if (mn + 0 < MN) v.x = base[off + 0];
                if (mn + 1 < MN) v.y = base[off + 1];
                if (mn + 2 < MN) v.z = base[off + 2];
                if (mn + 3 < MN) v.w = base[off + 3];
            }
        }
        r[i] = v;
    }
}

// ---- LDS-DMA pipeline pieces (K step 16, lane-linear LDS images) ------------------------------
constexpr int GK = 16;                         // K step of the DMA pipeline
constexpr int G_TILE = BM * GK;                // floats per operand tile (8 KB)
constexpr int G_STAGE = 2 * G_TILE;            // A + B

typedef __attribute__((address_space(3))) void lds_void;

// One LDS-DMA piece: 64 lanes x 16 bytes from (descriptor, per-lane voffset + scalar soffset) to 1 KiB of
// LDS at `lds` (wave-uniform).  A plain function on purpose: called with value-dependent arguments straight
// from a kernel TEMPLATE, hipcc 7.2 silently fails to emit that template's host-side launch stub.
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rsrc, float *lds, unsigned voffset, unsigned soffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void *)lds, 16, voffset, soffset, 0, 0);
}

// The barrier of an LDS-DMA pipeline: every piece THIS wave issued has landed (vmcnt), then the workgroup barrier, so
// every wave's pieces have.  The wait is explicit on purpose: hipcc's wait-count insertion does not treat a pending
// buffer_load ... lds as something a later ds_read of those bytes depends on, and emits `s_waitcnt lgkmcnt(0)` only in
// front of a __syncthreads() unless an ordinary load happens to be outstanding too (found with the HALF blocks of the
// Conv2D filter gradient under the split-bf16 modes: short MFMA phases, stale tiles, 1e-2 errors).
__device__ __forceinline__ void dma_barrier() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------------
// LDS-DMA issued from inline assembly.  Through the builtin, hipcc (ROCm 7.2) knows that an LDS-DMA piece is a
// pending LDS write on the vector-memory counter and puts `s_waitcnt vmcnt(0)` in front of the next ds_read in the same
// basic block -- i.e. right after the NEXT tile's pieces were issued it waits for them, and every other outstanding
// load, before reading the CURRENT tile: the prefetch never overlapped anything (the attention kernels of round 2 had that wait;
// whether it appears depends on the control flow between issue and read).  These pipelines order their stages
// themselves (a counted `s_waitcnt vmcnt` + the workgroup barrier at the top of every tile), so the pieces are
// invisible to the compiler: no wait of its own, no VALU for the addresses (the tile offset is the SCALAR offset of
// the instruction, range-checked together with the lane's offset), M0 saved and restored around each group.
// hipcc's counted waits for its OWN loads do not see these pieces; a piece issued after such a load only makes that
// wait stricter than needed (loads return in order), never too lax.
// ---------------------------------------------------------------------------------------------------------------
typedef int i32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ i32x4_t make_desc(const void *base, long bytes) {      // the words of make_rsrc, in SGPRs
    const unsigned long a = (unsigned long)base;
    i32x4_t d;
    d.x = __builtin_amdgcn_readfirstlane((int)a);
    d.y = __builtin_amdgcn_readfirstlane((int)(a >> 32) & 0xffff);
    d.z = __builtin_amdgcn_readfirstlane((int)(bytes < 0 ? 0 : bytes > 0x7FFFFFF0L ? 0x7FFFFFF0L : bytes));
    d.w = 0x00020000;
    return d;
}
__device__ __forceinline__ unsigned lds_offset(const float *p) {                  // byte address inside LDS (wave-uniform)
    return __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(lds_void *)p);
}

// N consecutive 1 KiB pieces starting at LDS byte address `lds`: piece i copies 16 bytes per lane from
// (desc, voff[i] + soff).  The wait state between a write of M0 and the instruction that reads it is the s_nop.
#define NPM_DMA_FIRST "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
#define NPM_DMA_NEXT "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
#define NPM_DMA_LOAD(v) "buffer_load_dwordx4 " v ", %2, %3 offen lds\n\t"
#define NPM_DMA_LAST "s_mov_b32 m0, %0"
__device__ __forceinline__ void dma_group1(i32x4_t desc, unsigned lds, unsigned soff, unsigned v0) {
    unsigned keep;
    asm volatile(NPM_DMA_FIRST NPM_DMA_LOAD("%4") NPM_DMA_LAST
                 : "=&s"(keep) : "s"(lds), "s"(desc), "s"(soff), "v"(v0) : "memory", "scc");
}
__device__ __forceinline__ void dma_group2(i32x4_t desc, unsigned lds, unsigned soff, unsigned v0, unsigned v1) {
    unsigned keep;
    asm volatile(NPM_DMA_FIRST NPM_DMA_LOAD("%4") NPM_DMA_NEXT NPM_DMA_LOAD("%5") NPM_DMA_LAST
                 : "=&s"(keep) : "s"(lds), "s"(desc), "s"(soff), "v"(v0), "v"(v1) : "memory", "scc");
}
__device__ __forceinline__ void dma_group3(i32x4_t desc, unsigned lds, unsigned soff, unsigned v0, unsigned v1, unsigned v2) {
    unsigned keep;
    asm volatile(NPM_DMA_FIRST NPM_DMA_LOAD("%4") NPM_DMA_NEXT NPM_DMA_LOAD("%5") NPM_DMA_NEXT NPM_DMA_LOAD("%6") NPM_DMA_LAST
                 : "=&s"(keep) : "s"(lds), "s"(desc), "s"(soff), "v"(v0), "v"(v1), "v"(v2) : "memory", "scc");
}
__device__ __forceinline__ void dma_group4(i32x4_t desc, unsigned lds, unsigned soff, unsigned v0, unsigned v1, unsigned v2, unsigned v3) {
    unsigned keep;
    asm volatile(NPM_DMA_FIRST NPM_DMA_LOAD("%4") NPM_DMA_NEXT NPM_DMA_LOAD("%5") NPM_DMA_NEXT NPM_DMA_LOAD("%6") NPM_DMA_NEXT NPM_DMA_LOAD("%7") NPM_DMA_LAST
                 : "=&s"(keep) : "s"(lds), "s"(desc), "s"(soff), "v"(v0), "v"(v1), "v"(v2), "v"(v3) : "memory", "scc");
}
template <int N>
__device__ __forceinline__ void dma_group(i32x4_t desc, unsigned lds, unsigned soff, const unsigned (&v)[N]) {
    static_assert(N == 1 || N == 2 || N == 3 || N % 4 == 0, "pieces per wave");
    if constexpr (N == 1) dma_group1(desc, lds, soff, v[0]);
    else if constexpr (N == 2) dma_group2(desc, lds, soff, v[0], v[1]);
    else if constexpr (N == 3) dma_group3(desc, lds, soff, v[0], v[1], v[2]);
    else {
#pragma unroll
        for (int i = 0; i < N; i += 4) dma_group4(desc, lds + i * 1024, soff, v[i], v[i + 1], v[i + 2], v[i + 3]);
    }
}


// WIDTH: rows (K-major) / floats per k row (MN-major) of the operand tile: 128, or 256 for the wide tile.
template <bool KMAJ, int WIDTH = BM>
__device__ __forceinline__ unsigned glds_voffset(int lane, int j, long ld) {
    // byte offset of this lane's 16-byte chunk for DMA piece j (0 .. WIDTH/16 - 1) of a tile, relative
    // to the tile's first row (K-major) / first k row (MN-major), K offset excluded
    if (KMAJ) {
        const int row = 16 * j + (lane >> 2);
        const int c = (lane & 3) ^ ((row >> 2) & 3);
        return (unsigned)(row * ld * 4 + c * 16);
    } else {
        constexpr int LANES_PER_ROW = WIDTH / 4, ROWS_PER_PIECE = 64 / LANES_PER_ROW;
        const int krow = ROWS_PER_PIECE * j + lane / LANES_PER_ROW;
        return (unsigned)(krow * ld * 4 + (lane % LANES_PER_ROW) * 16);
    }
}

// MN_PITCH: floats per k row of an MN-major tile (the tile's width: 128, or 64 for the tall variant)
template <bool KMAJ, int MN_PITCH = BM>
__device__ __forceinline__ float4 read_frag16(const float *__restrict__ s, int row, int g, int half) {
    if (KMAJ) {
        const int c = (2 * g + half) ^ ((row >> 2) & 3);
        return *reinterpret_cast<const float4 *>(s + row * GK + c * 4);
    } else {
        const float *p = s + (8 * g + 4 * half) * MN_PITCH + row;
        return make_float4(p[0], p[MN_PITCH], p[2 * MN_PITCH], p[3 * MN_PITCH]);
    }
}

// The 32 MFMAs of one 16-deep K tile for one wave.
template <bool A_KMAJ, bool B_KMAJ, int B_PITCH = BM, int A_PITCH = BM>
__device__ __forceinline__ void mma_tile16(const float *__restrict__ sA, const float *__restrict__ sB,
                                           int arow, int brow, int half, f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int g = 0; g < GK / 8; ++g) {
        const float4 a0 = read_frag16<A_KMAJ, A_PITCH>(sA, arow, g, half);
        const float4 a1 = read_frag16<A_KMAJ, A_PITCH>(sA, arow + 32, g, half);
        const float4 b0 = read_frag16<B_KMAJ, B_PITCH>(sB, brow, g, half);
        const float4 b1 = read_frag16<B_KMAJ, B_PITCH>(sB, brow + 32, g, half);
        const float av[2][4] = {{a0.x, a0.y, a0.z, a0.w}, {a1.x, a1.y, a1.z, a1.w}};
        const float bv[2][4] = {{b0.x, b0.y, b0.z, b0.w}, {b1.x, b1.y, b1.z, b1.w}};
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][s], bv[j][s], acc[i][j], 0, 0, 0);
    }
}

// ---- fp32 product on the bf16 matrix pipe: three-way split, six MFMAs ---------------------------------------
// a = a_hi + a_mid + a_lo with each part a bf16 (8 mantissa bits; the parts add up to a within 2^-23 |a|), and
// a b ~= hi hi + hi mid + mid hi + mid mid + hi lo + lo hi  (the dropped terms are below 2^-21 |a b|, zero-mean).  bf16 x bf16 products are exact in fp32 and the MFMA accumulates in fp32, so the result carries
// fp32-class error while running on v_mfma_f32_32x32x16_bf16 (16x the rate of the f32 MFMA, six issues per
// product: 2.67x).  Operand fragments are split in registers right after the LDS read -- the tiles in LDS and
// everything before them stay fp32, the accumulator layout is that of every 32x32 MFMA, so pipeline and
// epilogues are shared with the f32 path.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct Split3 { u32x4 hi, mid, lo; };       // 8 bf16 each: element t = k 8 half + t

// The 8 consecutive k (8 half .. 8 half + 7) of one operand row, from either tile layout.
template <bool KMAJ, int MN_PITCH>
__device__ __forceinline__ void read_k8(const float *__restrict__ s, int row, int half, float (&v)[8]) {
    if (KMAJ) {
        const int sw = (row >> 2) & 3;
        const float4 x = *reinterpret_cast<const float4 *>(s + row * GK + ((2 * half) ^ sw) * 4);
        const float4 y = *reinterpret_cast<const float4 *>(s + row * GK + ((2 * half + 1) ^ sw) * 4);
        v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w; v[4] = y.x; v[5] = y.y; v[6] = y.z; v[7] = y.w;
    } else {
        const float *p = s + 8 * half * MN_PITCH + row;
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] = p[t * MN_PITCH];
    }
}

// Plain v_sub_f32 on purpose: beside MFMAs a packed v_pk_add_f32 costs about four times the issue time of a
// scalar one (MI355X_MICROARCH.md, per-instruction constants), so the subtractions go through an asm statement
// the SLP vectoriser cannot pair.
__device__ __forceinline__ float sub_f32(float a, float b) {
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// hi is ROUNDED to nearest (v_cvt_pk_bf16_f32), so the residual has either sign and everything cut off further
// down -- the bits below lo and the dropped cross terms mid lo, lo mid, lo lo -- is zero-mean.  (With a truncated
// hi all parts share the sign of a, the dropped terms all carry the sign of a b and column checksums drift by
// 2^-22 of sum |a b|: measured 1e-5 relative on the C2 checksum.)  mid and lo are cut by truncation: the
// residuals are exact either way and the pack takes the upper halves directly.
__device__ __forceinline__ Split3 split3(const float (&v)[8]) {
    Split3 out;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[2 * t], v[2 * t + 1]}, bf16x2));
        const float p0 = sub_f32(v[2 * t], __uint_as_float(h << 16));
        const float p1 = sub_f32(v[2 * t + 1], __uint_as_float(h & 0xffff0000u));
        const unsigned m0 = __float_as_uint(p0), m1 = __float_as_uint(p1);
        const float q0 = sub_f32(p0, __uint_as_float(m0 & 0xffff0000u));
        const float q1 = sub_f32(p1, __uint_as_float(m1 & 0xffff0000u));
        out.hi[t] = h;
        out.mid[t] = __builtin_amdgcn_perm(m1, m0, 0x07060302);       // the upper halves of two words
        out.lo[t] = __builtin_amdgcn_perm(__float_as_uint(q1), __float_as_uint(q0), 0x07060302);
    }
    return out;
}

__device__ __forceinline__ void mfma_bf16(const u32x4 &a, const u32x4 &b, f32x16 &c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// The 24 MFMAs of one 16-deep K tile for one wave (64 x 64).  The five small terms go to their own accumulators
// (`small`), added to `acc` once after the K loop: the matrix pipe aligns every addend to the accumulator and
// cuts what falls below it (floor), so small terms added straight into a large accumulator leave a negative bias
// (measured -0.5 ulp at K = 4096, and it adds up coherently in column checksums); among themselves they are of
// like magnitude and the cut is 2^8 times smaller.
template <bool A_KMAJ, bool B_KMAJ, int B_PITCH = BM, int A_PITCH = BM>
__device__ __forceinline__ void mma_tile16_bf16x6(const float *__restrict__ sA, const float *__restrict__ sB,
                                                  int arow, int brow, int half, f32x16 (&acc)[2][2], f32x16 (&small)[2][2]) {
    Split3 a[2], b[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float v[8];
        read_k8<A_KMAJ, A_PITCH>(sA, arow + 32 * i, half, v);
        a[i] = split3(v);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        float v[8];
        read_k8<B_KMAJ, B_PITCH>(sB, brow + 32 * j, half, v);
        b[j] = split3(v);
    }
    // per term the four accumulators are independent: no MFMA waits for the one before it
#define NPM_TERM(PA, PB, ACC)                                                                \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                            \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) mfma_bf16(a[i].PA, b[j].PB, ACC[i][j]);
    NPM_TERM(lo, hi, small)
    NPM_TERM(hi, lo, small)
    NPM_TERM(mid, mid, small)
    NPM_TERM(mid, hi, small)
    NPM_TERM(hi, mid, small)
    NPM_TERM(hi, hi, acc)
#undef NPM_TERM
}

// MATH 0: exact-f32 MFMA (v_mfma_f32_32x32x2_f32); 1: three-way bf16 split, six bf16 MFMAs per product.
// MATH 1: the small terms share `acc`; MATH 2: they have their own accumulators (64 more registers, unbiased).
template <int MATH, bool A_KMAJ, bool B_KMAJ, int B_PITCH = BM, int A_PITCH = BM>
__device__ __forceinline__ void mma_tile16_math(const float *__restrict__ sA, const float *__restrict__ sB,
                                                int arow, int brow, int half, f32x16 (&acc)[2][2], f32x16 (&small)[2][2]) {
    if (MATH == 2) mma_tile16_bf16x6<A_KMAJ, B_KMAJ, B_PITCH, A_PITCH>(sA, sB, arow, brow, half, acc, small);
    else if (MATH == 1) mma_tile16_bf16x6<A_KMAJ, B_KMAJ, B_PITCH, A_PITCH>(sA, sB, arow, brow, half, acc, acc);
    else mma_tile16<A_KMAJ, B_KMAJ, B_PITCH, A_PITCH>(sA, sB, arow, brow, half, acc);
}

// ---- K rendezvous of co-resident blocks ----------------------------------------------------------------------------
// The blocks of a split-K launch that fits the chip at once (weight gradients: K = 131072 samples) stream the same operand
// panels through their XCD's L2, each at its own pace; after a few hundred K tiles they are further apart than the L2 holds and
// every panel is fetched from the Infinity Cache several times (15 GB per FFN weight-gradient launch against 2.7 GB algorithmic).
// Every `every` K tiles thread 0 of each block checks in at a counter of its XCD group (blockIdx & 7, the XCD the dispatcher
// gave it) and waits until the whole group has: a SOFT rendezvous -- bounded spinning, and the first block that gives up
// raises a flag that ends all waiting for the launch, so a block that is not resident (another kernel holding CUs) costs one
// timeout, never a hang.  Results do not depend on it.  Measured on the FFN weight gradients: L2 read requests to the fabric
// 1.17e8 -> 6.2e7 per launch, 7.93 -> 7.65 ms.
struct KSync {
    unsigned *slice;     // this launch's counters: [8 groups][32 words], word 0 = arrivals, word 1 = "stop waiting"; or null
    int every;           // K tiles between rendezvous, a power of two
    int epochs;          // rendezvous every block of the launch reaches (the shortest split decides)
};

// The rendezvous itself: out of line on purpose -- inlined into a K loop its spin loop (and the per-lane test in front of it) cost
// every launch of the kernel 1.4 %, the ones that never rendezvous included (A/B of the two builds in one session).
__device__ __attribute__((noinline)) void ksync_rendezvous(unsigned *slice, unsigned epoch) {
    if (threadIdx.x != 0) return;
    unsigned *ctr = slice + (blockIdx.x & 7) * 32;
    if (__hip_atomic_load(ctr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    const unsigned target = epoch * ((gridDim.x - (blockIdx.x & 7) + 7) >> 3);
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int spin = 0; __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target; ++spin) {
        if (spin >= 256 || __hip_atomic_load(ctr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            __hip_atomic_store(ctr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
        __builtin_amdgcn_s_sleep(8);
    }
}

// In the K loop: wave-uniform tests only (scalar instructions cost the matrix pipe nothing), then the call.
__device__ __forceinline__ void ksync_wait(const KSync &k, int kt) {
    if (k.slice != nullptr && kt > 0 && (kt & (k.every - 1)) == 0 && kt / k.every <= k.epochs)
        ksync_rendezvous(k.slice, (unsigned)(kt / k.every));
}

// Split-K factor for a grid of `tiles` output tiles over `nkt` K tiles.  All blocks of such a launch are
// resident at once (`resident` fit a CU: 4 with the f32 MFMA, 3 / 2 with the split-bf16 modes) and run equally
// long, so the launch lasts as long as the fullest CU: cost(s) = ceil(tiles * s / CUs) / s.  Candidates leave the
// fullest CU with `resident` or `resident - 1` blocks (fewer cannot keep the matrix pipe busy, more would queue a
// second generation).  `force_per_cu` > 0 pins the blocks-per-CU target.
inline int pick_splits(long tiles, int nkt, long cus, int force_per_cu = 0, int resident = 4, int generations = 0) {
    const long max_s = nkt / 8;
    if (force_per_cu > resident) {                       // several generations of `resident` blocks, pinned (experiments)
        const long s = (long)force_per_cu * cus / tiles;
        return (int)(s < 1 ? 1 : s > max_s ? max_s : s);
    }
    int best = 1;
    double best_cost = 1e30;
    // candidates in order of preference: 3 blocks per CU where that many fit (measured on the FFN weight gradients,
    // f32 math: 768 blocks 7.57 ms, 1024 blocks 8.15 ms), then the other neighbour of `resident`
    const int first = resident < 3 ? resident : 3;
    const int second = first == resident ? resident - 1 : resident;
    const int order[2] = {first, second < 2 ? first : second};
    for (int c = 0; c < 2; ++c) {
        const int per_cu = order[c];
        if (force_per_cu > 0 && force_per_cu <= resident && per_cu != force_per_cu) continue;
        long s = per_cu * cus / tiles;
        if (s > max_s) s = max_s;
        if (s < 1) s = 1;
        const double cost = (double)((tiles * s + cus - 1) / cus) / (double)s;
        if (cost < best_cost * (1.0 - 1e-6)) { best_cost = cost; best = (int)s; }
    }
    // Long K ranges (round 4, `generations` > 0): the blocks of one generation stream the same operand panels through their XCD's
    // L2 and drift apart over a thousand K tiles (the packed q/k/v weight gradient, 192 tiles x 4 splits of 1024 K tiles of 32:
    // 6.00 ms; x 16 splits of 256: 5.71 ms).  When the chosen range is longer than 768 K tiles and an exact number of FULL
    // generations (`resident` blocks on every CU, 3 then 2 of them) exists with at least 128 K tiles per block, take that.
    if (generations > 0 && force_per_cu == 0 && nkt / best > 768) {
        for (int gens = 3; gens >= 2; --gens) {
            const long blocks = (long)resident * gens * cus;
            if (blocks % tiles) continue;
            const long s = blocks / tiles;
            if (s <= best || s > max_s || nkt / s < 128) continue;
            return (int)s;
        }
    }
    return best;
}

// Sum split-K slabs in split order and apply the linear part of the epilogue.
struct ReduceArgs {
    const float *ws;
    long slab;          // elements per split = batch * M * N
    int splits, M, N, batch1;
    long sC0, sC1;
    Epilogue e;
};

int launch_splitk_reduce(const ReduceArgs &r, hipStream_t stream);

// fp32 GEMM on the f16 matrix pipe, two-way split with row scaling (npm_gemm_f16x2.hip; NPM_MATH_F16X2).  One product, no
// batch: C[M, N] = op(A) op(B) with the epilogue / split-K machinery of the other kernels.
struct F16x2Args {
    const float *A, *B;
    long lda, ldb;
    int M, N, K;
    int a_kmaj, b_kmaj;      // A stored [M][K] (else [K][M]); B stored [N][K] (else [K][N])
    int tiles_m, tiles_n, group_m, splits, k_per_split;
    long slab;
    const float *sa, *sb, *inv_sa, *inv_sb;      // scales along M and N (f16x2_scales) and their reciprocals
    Epilogue e;
};
// scale[i] = 2^(14 - e_i) where the largest magnitude along K of row / column i is f 2^e_i, and inv[i] = 1 / scale[i];
// `umax` is scratch of `extent` words.  kmaj: x is [extent][k] (pitch ld), else [k][extent].
// colsum_out (MN-major operands only): also out[i] = sum over k of column i, from the same pass.
int f16x2_scales(const float *x, long ld, bool kmaj, long extent, long k, unsigned *umax, float *scale, float *inv, hipStream_t stream,
                 float *colsum_out = nullptr);
int launch_f16x2(const F16x2Args &a, hipStream_t stream);

}  // namespace npm_tile
