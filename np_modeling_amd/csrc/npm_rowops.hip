// HBM-bound kernels of the hot path: elementwise ops, column sums and the
// one-wavefront-per-row softmax / LayerNorm forward and backward.
//
// Every kernel streams its operands once with 16-byte loads per lane (64 lanes x 16 B =
// 1 KiB per wave instruction) and keeps the row in registers between the reduction and
// the write, so the HBM traffic is the algorithmic minimum (DESIGN.md "Row kernels").
// Reductions across the 64-lane wavefront use DPP/shuffle butterflies, not LDS.
#include <algorithm>

#include "npm_internal.h"

namespace {

int g_ln_bwd_blocks_per_cu = 4;     // NPM_TUNE_LN_BWD_BLOCKS
// NPM_TUNE_LN_NT_SPLIT = backward mode + 4 * forward mode; a mode: 0 nontemporal hint on loads and stores, 1 on the loads only, 2 on
// the stores only (d in (512, 1024] -- the encoder's rows; other widths take mode 0).  Default 1 + 4 * 1: measured INSIDE the
// encoder step (profiles/r05_ln_nt_split.log) the backward runs 0.377 -> 0.362 ms with its dx stored under the default policy
// (the GEMM behind it reads dx at once), although from cold caches and alone the same variant is the slower one (5.1 against 5.5 TB/s);
// the forward likewise 0.176 -> 0.171 ms with z under the default policy (the hint on its STORES only: 0.205 ms)
int g_ln_nt_split = 1 + 4 * 1;
int g_stream_nt = 1;                // NPM_TUNE_STREAM_NT
constexpr int COLSUM_BLOCKS_PER_CU = 8;     // whole-line column sums: grid = this x CUs (2 .. 64 measured within 3 % of each other)

// Streaming tensors (read once / written once, far larger than the 32 MB of L2) move with the NONTEMPORAL hint: they
// do not displace what the GEMMs around them keep in L2 and the Infinity Cache, and from cold caches the kernels
// themselves run faster (LayerNorm backward 131072 x 1024: 0.439 -> 0.381 ms, profiles/r02_stream_nt.log).  Small
// tensors (parameters, partial sums) keep the default policy.
inline bool stream_nt(size_t bytes) { return g_stream_nt && bytes >= ((size_t)32 << 20); }

typedef float f32x4v __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ float4 ldg4(const float *p) {
    if (NT) {
        const f32x4v v = __builtin_nontemporal_load(reinterpret_cast<const f32x4v *>(p));
        return make_float4(v.x, v.y, v.z, v.w);
    }
    return *reinterpret_cast<const float4 *>(p);
}
template <bool NT>
__device__ __forceinline__ void stg4(float *p, const float4 &v) {
    if (NT) __builtin_nontemporal_store(f32x4v{v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4v *>(p));
    else *reinterpret_cast<float4 *>(p) = v;
}
// launch KERNEL<..., NT> with NT chosen at run time
#define NPM_NT_LAUNCH(nt, LAUNCH) do { if (nt) { constexpr bool NT = true; LAUNCH; } else { constexpr bool NT = false; LAUNCH; } } while (0)

constexpr int WAVE = 64;
constexpr int ROWS_PER_BLOCK = 4;          // 256 threads = 4 waves, one row per wave

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, WAVE);
    return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, WAVE));
    return v;
}

int g_ew_grid_cap = 1 << 20;        // NPM_TUNE_EW_GRID_CAP: measured 4.7 TB/s at 8192 blocks, 6.1 TB/s uncapped (268 M elements)

inline int grid_for(size_t n_vec, int block = 256, int cap = 0) {
    if (cap == 0) cap = g_ew_grid_cap;
    size_t g = (n_vec + block - 1) / block;
    return (int)std::max<size_t>(1, std::min<size_t>(g, cap));
}

inline bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }

// ---------------------------------------------------------------------------------
// elementwise: float4 grid-stride body + scalar tail
// ---------------------------------------------------------------------------------
// Each thread moves 4 float4 per iteration with all loads issued before the first use (64 B in flight per
// operand and lane), grid-stride over the rest; the scalar tail handles n % 4.
constexpr int EW_UNROLL = 4;

template <typename F, bool NT>
__global__ void __launch_bounds__(256) ew1_kernel(const float *a, float *out, size_t n, F f) {
    const size_t nv = n / 4;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (EW_UNROLL - 1) * stride < nv; i += EW_UNROLL * stride) {
        float4 v[EW_UNROLL];
#pragma unroll
        for (int u = 0; u < EW_UNROLL; ++u) v[u] = ldg4<NT>(a + 4 * (i + u * stride));
#pragma unroll
        for (int u = 0; u < EW_UNROLL; ++u) {
            v[u].x = f(v[u].x); v[u].y = f(v[u].y); v[u].z = f(v[u].z); v[u].w = f(v[u].w);
            stg4<NT>(out + 4 * (i + u * stride), v[u]);
        }
    }
    for (; i < nv; i += stride) {
        float4 v = ldg4<NT>(a + 4 * (i));
        v.x = f(v.x); v.y = f(v.y); v.z = f(v.z); v.w = f(v.w);
        stg4<NT>(out + 4 * (i), v);
    }
    for (size_t t = nv * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) out[t] = f(a[t]);
}

template <typename F, bool NT>
__global__ void __launch_bounds__(256) ew2_kernel(const float *a, const float *b, float *out, size_t n, F f) {
    const size_t nv = n / 4;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (EW_UNROLL - 1) * stride < nv; i += EW_UNROLL * stride) {
        float4 x[EW_UNROLL], y[EW_UNROLL];
#pragma unroll
        for (int u = 0; u < EW_UNROLL; ++u) {
            x[u] = ldg4<NT>(a + 4 * (i + u * stride));
            y[u] = ldg4<NT>(b + 4 * (i + u * stride));
        }
#pragma unroll
        for (int u = 0; u < EW_UNROLL; ++u) {
            float4 v;
            v.x = f(x[u].x, y[u].x); v.y = f(x[u].y, y[u].y); v.z = f(x[u].z, y[u].z); v.w = f(x[u].w, y[u].w);
            stg4<NT>(out + 4 * (i + u * stride), v);
        }
    }
    for (; i < nv; i += stride) {
        const float4 x = ldg4<NT>(a + 4 * (i));
        const float4 y = ldg4<NT>(b + 4 * (i));
        float4 v;
        v.x = f(x.x, y.x); v.y = f(x.y, y.y); v.z = f(x.z, y.z); v.w = f(x.w, y.w);
        stg4<NT>(out + 4 * (i), v);
    }
    for (size_t t = nv * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) out[t] = f(a[t], b[t]);
}

template <typename F, bool NT>
__global__ void __launch_bounds__(256) ew3_kernel(const float *a, const float *b, const float *c, float *out, size_t n, F f) {
    const size_t nv = n / 4;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) {
        const float4 x = ldg4<NT>(a + 4 * (i));
        const float4 y = ldg4<NT>(b + 4 * (i));
        const float4 z = ldg4<NT>(c + 4 * (i));
        float4 v;
        v.x = f(x.x, y.x, z.x); v.y = f(x.y, y.y, z.y); v.z = f(x.z, y.z, z.z); v.w = f(x.w, y.w, z.w);
        stg4<NT>(out + 4 * (i), v);
    }
    for (size_t i = nv * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        out[i] = f(a[i], b[i], c[i]);
}

struct ReluF { __device__ float operator()(float x) const { return fmaxf(x, 0.f); } };
struct ReluBwdF { __device__ float operator()(float x, float dy) const { return x >= 0.f ? dy : 0.f; } };
struct AddF { __device__ float operator()(float a, float b) const { return a + b; } };
struct Add3F { __device__ float operator()(float a, float b, float c) const { return (a + b) + c; } };
struct AxpyF { float alpha; __device__ float operator()(float y, float x) const { return y + alpha * x; } };
struct ScaleF { float alpha; __device__ float operator()(float x) const { return alpha * x; } };
struct FillF { float v; __device__ float operator()(float) const { return v; } };

// scalar fallbacks for unaligned pointers
template <typename F>
__global__ void ew1_scalar(const float *a, float *out, size_t n, F f) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = f(a[i]);
}
template <typename F>
__global__ void ew2_scalar(const float *a, const float *b, float *out, size_t n, F f) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = f(a[i], b[i]);
}
template <typename F>
__global__ void ew3_scalar(const float *a, const float *b, const float *c, float *out, size_t n, F f) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = f(a[i], b[i], c[i]);
}

template <typename F>
int ew1(const float *a, float *out, size_t n, F f) {
    if (n == 0) return NPM_OK;
    hipStream_t s = npm::ctx().stream;
    if (aligned16(a) && aligned16(out)) NPM_NT_LAUNCH(stream_nt(4 * n), hipLaunchKernelGGL((ew1_kernel<F, NT>), dim3(grid_for(n / 4 + 1)), dim3(256), 0, s, a, out, n, f));
    else hipLaunchKernelGGL(ew1_scalar<F>, dim3(grid_for(n)), dim3(256), 0, s, a, out, n, f);
    NPM_CHECK_LAUNCH();
    return NPM_OK;
}
template <typename F>
int ew2(const float *a, const float *b, float *out, size_t n, F f) {
    if (n == 0) return NPM_OK;
    hipStream_t s = npm::ctx().stream;
    if (aligned16(a) && aligned16(b) && aligned16(out)) NPM_NT_LAUNCH(stream_nt(4 * n), hipLaunchKernelGGL((ew2_kernel<F, NT>), dim3(grid_for(n / 4 + 1)), dim3(256), 0, s, a, b, out, n, f));
    else hipLaunchKernelGGL(ew2_scalar<F>, dim3(grid_for(n)), dim3(256), 0, s, a, b, out, n, f);
    NPM_CHECK_LAUNCH();
    return NPM_OK;
}
template <typename F>
int ew3(const float *a, const float *b, const float *c, float *out, size_t n, F f) {
    if (n == 0) return NPM_OK;
    hipStream_t s = npm::ctx().stream;
    if (aligned16(a) && aligned16(b) && aligned16(c) && aligned16(out))
        NPM_NT_LAUNCH(stream_nt(4 * n), hipLaunchKernelGGL((ew3_kernel<F, NT>), dim3(grid_for(n / 4 + 1)), dim3(256), 0, s, a, b, c, out, n, f));
    else hipLaunchKernelGGL(ew3_scalar<F>, dim3(grid_for(n)), dim3(256), 0, s, a, b, c, out, n, f);
    NPM_CHECK_LAUNCH();
    return NPM_OK;
}

// ---------------------------------------------------------------------------------
// column sum: out[c] = sum_r x[r, c].  Stage 1: grid (col strips of 64, row chunks);
// 16 threads x float4 cover a 256-B row segment, 16 row lanes per block; partial sums
// cross the row lanes through LDS.  Stage 2 sums the chunk partials (fixed order).
// ---------------------------------------------------------------------------------
template <bool RELU_BWD, bool NT>
__global__ void __launch_bounds__(256)
colsum_kernel(const float *__restrict__ x, float *__restrict__ out, long rows, long cols, long ld,
              long rows_per_chunk, long out_ld, const float *__restrict__ dy, float *__restrict__ g) {
    // RELU_BWD: x is the pre-activation; g = (x >= 0 ? dy : 0) is stored and summed (activations.py:19 + conv.py:55)
    __shared__ float red[16][65];
    const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const long c0 = (long)blockIdx.x * 64 + cq * 4;
    const long r_beg = (long)blockIdx.y * rows_per_chunk;
    const long r_end = min(rows, r_beg + rows_per_chunk);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    bool full = (c0 + 3 < cols) && (ld % 4 == 0) && ((((uintptr_t)x) & 15) == 0);
    if (RELU_BWD) full = full && ((((uintptr_t)dy) & 15) == 0) && ((((uintptr_t)g) & 15) == 0);
    long r = r_beg + rl;
    if (full) {
        // four rows per thread and iteration, every load issued before the first use (128 B in flight per lane)
        for (; r + 48 < r_end; r += 64) {
            float4 v[4], d[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long at = (r + 16 * u) * ld + c0;
                v[u] = ldg4<NT>(x + at);
                if (RELU_BWD) d[u] = ldg4<NT>(dy + at);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (RELU_BWD) {
                    v[u].x = v[u].x >= 0.f ? d[u].x : 0.f; v[u].y = v[u].y >= 0.f ? d[u].y : 0.f;
                    v[u].z = v[u].z >= 0.f ? d[u].z : 0.f; v[u].w = v[u].w >= 0.f ? d[u].w : 0.f;
                    stg4<NT>(g + (r + 16 * u) * ld + c0, v[u]);
                }
                acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w;
            }
        }
    }
    for (; r < r_end; r += 16) {
        const long at = r * ld + c0;
        const float *p = x + at;
        if (full) {
            float4 v = ldg4<NT>(p);
            if (RELU_BWD) {
                const float4 d = ldg4<NT>(dy + at);
                v.x = v.x >= 0.f ? d.x : 0.f; v.y = v.y >= 0.f ? d.y : 0.f;
                v.z = v.z >= 0.f ? d.z : 0.f; v.w = v.w >= 0.f ? d.w : 0.f;
                stg4<NT>(g + at, v);
            }
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        } else {
            float *a4 = &acc.x;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (c0 + j < cols) {
                    float v = p[j];
                    if (RELU_BWD) { v = v >= 0.f ? dy[at + j] : 0.f; g[at + j] = v; }
                    a4[j] += v;
                }
            }
        }
    }
    red[rl][cq * 4 + 0] = acc.x; red[rl][cq * 4 + 1] = acc.y;
    red[rl][cq * 4 + 2] = acc.z; red[rl][cq * 4 + 3] = acc.w;
    __syncthreads();
    if (threadIdx.x < 64) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += red[i][threadIdx.x];
        const long c = (long)blockIdx.x * 64 + threadIdx.x;
        if (c < cols) out[(long)blockIdx.y * out_ld + c] = s;
    }
}

// Narrow dense matrices (cols divides 1024, e.g. the 128 output channels of C3's Conv2D: 12.8 M rows): 1024 / cols
// consecutive rows are one 4 KB line of 1024 floats; a block streams whole lines (256 threads x 16 bytes, four lines
// per iteration, every load issued before the first use) and thread t keeps the sums of its four columns in registers:
// no LDS, no strided 256-byte strips.  part[block][1024] then folds to [cols] in the ordinary second stage
// (its rows are (block, line position) pairs).
template <bool RELU_BWD, bool NT>
__global__ void __launch_bounds__(256)
colsum_lines_kernel(const float *__restrict__ x, float *__restrict__ part, long lines, long lines_per_block,
                    const float *__restrict__ dy, float *__restrict__ g) {
    // Block b owns lines [b lpb, (b + 1) lpb) (blocks striding through the tensor together measured slower: 3.9 vs
    // 3.5-3.8 ms at C3).  Software pipeline: the next four lines are requested before the current four are reduced and
    // stored, so loads and stores of a wave overlap instead of alternating.
    const long l_beg = (long)blockIdx.x * lines_per_block, l_end = min(lines, l_beg + lines_per_block);
    const int c = threadIdx.x * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    auto fetch = [&](long l, float4 (&v)[4], float4 (&d)[4]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            v[u] = ldg4<NT>(x + (l + u) * 1024 + c);
            if (RELU_BWD) d[u] = ldg4<NT>(dy + (l + u) * 1024 + c);
        }
    };
    auto consume = [&](long l, float4 (&v)[4], float4 (&d)[4]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (RELU_BWD) {
                v[u].x = v[u].x >= 0.f ? d[u].x : 0.f; v[u].y = v[u].y >= 0.f ? d[u].y : 0.f;
                v[u].z = v[u].z >= 0.f ? d[u].z : 0.f; v[u].w = v[u].w >= 0.f ? d[u].w : 0.f;
                stg4<NT>(g + (l + u) * 1024 + c, v[u]);
            }
            acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w;
        }
    };
    long l = l_beg;
    if (l + 3 < l_end) {
        float4 v0[4], d0[4], v1[4], d1[4];
        fetch(l, v0, d0);
        for (; l + 11 < l_end; l += 8) {            // two groups per trip: the buffers keep their names
            fetch(l + 4, v1, d1);
            consume(l, v0, d0);
            fetch(l + 8, v0, d0);
            consume(l + 4, v1, d1);
        }
        if (l + 7 < l_end) {
            fetch(l + 4, v1, d1);
            consume(l, v0, d0);
            consume(l + 4, v1, d1);
            l += 8;
        } else {
            consume(l, v0, d0);
            l += 4;
        }
    }
    for (; l < l_end; ++l) {
        float4 v = ldg4<NT>(x + l * 1024 + c);
        if (RELU_BWD) {
            const float4 d = ldg4<NT>(dy + l * 1024 + c);
            v.x = v.x >= 0.f ? d.x : 0.f; v.y = v.y >= 0.f ? d.y : 0.f;
            v.z = v.z >= 0.f ? d.z : 0.f; v.w = v.w >= 0.f ? d.w : 0.f;
            stg4<NT>(g + l * 1024 + c, v);
        }
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    *reinterpret_cast<float4 *>(part + (long)blockIdx.x * 1024 + c) = acc;
}

template <bool RELU_BWD>
int colsum_run(const float *x, float *out, long rows, long cols, long ld, const float *dy, float *g) {
    hipStream_t s = npm::ctx().stream;
    if (rows <= 0) return ew1(out, out, (size_t)cols, FillF{0.f});          // an empty batch sums to zero
    if (ld == cols && cols >= 4 && cols < 1024 && 1024 % cols == 0 && rows % (1024 / cols) == 0 && rows * cols >= (1L << 22) &&
        aligned16(x) && (!RELU_BWD || (aligned16(dy) && aligned16(g)))) {
        const long lines = rows / (1024 / cols);
        const long blocks = std::min<long>(lines / 16, (long)COLSUM_BLOCKS_PER_CU * npm::ctx().num_cus);   // >= 16 lines per block
        const long lpb = (lines + blocks - 1) / blocks;
        const long used = (lines + lpb - 1) / lpb;
        npm::Scratch part;
        int rc = part.alloc(sizeof(float) * (size_t)used * 1024);
        if (rc) return rc;
        NPM_NT_LAUNCH(stream_nt(sizeof(float) * (size_t)rows * cols),
                      hipLaunchKernelGGL((colsum_lines_kernel<RELU_BWD, NT>), dim3((int)used), dim3(256), 0, s, x, (float *)part.ptr, lines, lpb, dy, g));
        NPM_CHECK_LAUNCH();
        return colsum_run<false>((const float *)part.ptr, out, used * (1024 / cols), cols, cols, nullptr, nullptr);
    }
    const int strips = (int)((cols + 63) / 64);
    long chunks = std::max<long>(1, std::min<long>((rows + 255) / 256, std::max<long>(1, 2048 / strips)));
    const long rpc = (rows + chunks - 1) / chunks;
    chunks = (rows + rpc - 1) / rpc;
    if (chunks <= 1) {
        hipLaunchKernelGGL((colsum_kernel<RELU_BWD, false>), dim3(strips, 1), dim3(256), 0, s, x, out, rows, cols, ld,
                           std::max<long>(rows, 1), cols, dy, g);
        NPM_CHECK_LAUNCH();
        return NPM_OK;
    }
    npm::Scratch part;
    int rc = part.alloc(sizeof(float) * (size_t)chunks * cols);
    if (rc) return rc;
    NPM_NT_LAUNCH(stream_nt(sizeof(float) * (size_t)rows * cols),
                  hipLaunchKernelGGL((colsum_kernel<RELU_BWD, NT>), dim3(strips, (int)chunks), dim3(256), 0, s, x, (float *)part.ptr, rows, cols, ld,
                                     rpc, cols, dy, g));
    NPM_CHECK_LAUNCH();
    hipLaunchKernelGGL((colsum_kernel<false, false>), dim3(strips, 1), dim3(256), 0, s, (const float *)part.ptr, out, chunks, cols, cols,
                       chunks, cols, (const float *)nullptr, (float *)nullptr);
    NPM_CHECK_LAUNCH();
    return NPM_OK;
}

int colsum_impl(const float *x, float *out, long rows, long cols, long ld) {
    return colsum_run<false>(x, out, rows, cols, ld, nullptr, nullptr);
}

// ---------------------------------------------------------------------------------
// Row kernels.  A wave owns a row; lane l holds float4 chunks l, l+64, ... (VPL chunks),
// i.e. every wave instruction reads 1 KiB contiguous.  VPL is a template so the row
// lives in registers (n <= 256*VPL).  Rows that are not a multiple of 4 long, or longer
// than 4096, take the generic re-reading kernels further down.
// ---------------------------------------------------------------------------------
template <int VPL, bool NT>
__device__ __forceinline__ void load_row(const float *__restrict__ p, int nvec, int lane, float4 (&v)[VPL], float fill) {
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        const int c = lane + WAVE * j;
        v[j] = (c < nvec) ? ldg4<NT>(p + 4 * c) : make_float4(fill, fill, fill, fill);
    }
}

template <int VPL, bool NT>
__device__ __forceinline__ void store_row(float *__restrict__ p, int nvec, int lane, const float4 (&v)[VPL]) {
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        const int c = lane + WAVE * j;
        if (c < nvec) stg4<NT>(p + 4 * c, v[j]);
    }
}

template <int VPL, bool NT>
__global__ void __launch_bounds__(256)
softmax_fwd_kernel(const float *__restrict__ x, float *__restrict__ y, long rows, int n, float scale) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nvec = n >> 2;
    float4 v[VPL];
    load_row<VPL, NT>(x + row * n, nvec, lane, v, -INFINITY);
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        v[j].x *= scale; v[j].y *= scale; v[j].z *= scale; v[j].w *= scale;
        m = fmaxf(m, fmaxf(fmaxf(v[j].x, v[j].y), fmaxf(v[j].z, v[j].w)));
    }
    m = wave_max(m);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        v[j].x = expf(v[j].x - m); v[j].y = expf(v[j].y - m);
        v[j].z = expf(v[j].z - m); v[j].w = expf(v[j].w - m);
        s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
    }
    s = wave_sum(s);
    const float inv = 1.0f / s;
#pragma unroll
    for (int j = 0; j < VPL; ++j) { v[j].x *= inv; v[j].y *= inv; v[j].z *= inv; v[j].w *= inv; }
    store_row<VPL, NT>(y + row * n, nvec, lane, v);
}

template <int VPL, bool NT>
__global__ void __launch_bounds__(256)
softmax_bwd_kernel(const float *__restrict__ y, const float *__restrict__ dy, float *__restrict__ dx,
                   long rows, int n, float scale) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nvec = n >> 2;
    float4 p[VPL], g[VPL];
    load_row<VPL, NT>(y + row * n, nvec, lane, p, 0.f);
    load_row<VPL, NT>(dy + row * n, nvec, lane, g, 0.f);
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < VPL; ++j) dot += (p[j].x * g[j].x + p[j].y * g[j].y) + (p[j].z * g[j].z + p[j].w * g[j].w);
    dot = wave_sum(dot);
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        g[j].x = scale * p[j].x * (g[j].x - dot); g[j].y = scale * p[j].y * (g[j].y - dot);
        g[j].z = scale * p[j].z * (g[j].z - dot); g[j].w = scale * p[j].w * (g[j].w - dot);
    }
    store_row<VPL, NT>(dx + row * n, nvec, lane, g);
}

// generic (any n, any alignment): re-reads the row; served by L2 after the first pass
__global__ void __launch_bounds__(256)
softmax_fwd_generic(const float *__restrict__ x, float *__restrict__ y, long rows, long n, float scale) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *xr = x + row * n;
    float *yr = y + row * n;
    float m = -INFINITY;
    for (long c = lane; c < n; c += WAVE) m = fmaxf(m, scale * xr[c]);
    m = wave_max(m);
    float s = 0.f;
    for (long c = lane; c < n; c += WAVE) s += expf(scale * xr[c] - m);
    s = wave_sum(s);
    const float inv = 1.0f / s;
    for (long c = lane; c < n; c += WAVE) yr[c] = expf(scale * xr[c] - m) * inv;
}

__global__ void __launch_bounds__(256)
softmax_bwd_generic(const float *__restrict__ y, const float *__restrict__ dy, float *__restrict__ dx,
                    long rows, long n, float scale) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *yr = y + row * n, *gr = dy + row * n;
    float dot = 0.f;
    for (long c = lane; c < n; c += WAVE) dot += yr[c] * gr[c];
    dot = wave_sum(dot);
    for (long c = lane; c < n; c += WAVE) dx[row * n + c] = scale * yr[c] * (gr[c] - dot);
}

// out[(b*H + h)*S + s] = sum_d a[b,s,h,d] * b[b,s,h,d].  Half a wavefront (32 lanes x float4) covers a
// 128-wide head; general dims loop.  Reads both tensors once (8 B/element).
template <bool NT>
__global__ void __launch_bounds__(256)
attn_rowdot_kernel(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ out,
                   long batch, long seq, long heads, long dim) {
    const long rows = batch * seq * heads;
    const int sub = threadIdx.x & 31;
    const long row = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 5;      // (b, s, h) flattened
    if (row >= rows) return;
    const float *pa = a + row * dim, *pb = b + row * dim;
    float acc = 0.f;
    if ((dim & 3) == 0 && ((((uintptr_t)a) | ((uintptr_t)b)) & 15) == 0) {
        for (long c = sub * 4; c < dim; c += 128) {
            const float4 x = ldg4<NT>(pa + c), y = ldg4<NT>(pb + c);
            acc += (x.x * y.x + x.y * y.y) + (x.z * y.z + x.w * y.w);
        }
    } else {
        for (long c = sub; c < dim; c += 32) acc += pa[c] * pb[c];
    }
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) acc += __shfl_xor(acc, off, WAVE);
    if (sub == 0) {
        const long h = row % heads, bs = row / heads;
        const long s_ = bs % seq, b_ = bs / seq;
        out[(b_ * heads + h) * seq + s_] = acc;
    }
}

// ---- LayerNorm ---------------------------------------------------------------------
// DROP: the row is DropOut's output, formed on the way in (reference transformer.py:35-36,40-41,49-50,55-56: a dropout
// always sits directly in front of a LayerNormalization): x_i <- mask_i ? x_i / keep : 0 (normalizations.py:21-23), one byte
// of mask per element, four per 32-bit load.  The dropped tensor is never written.
__device__ __forceinline__ float4 drop4(float4 v, unsigned m, float keep) {
    return make_float4((m & 0xffu) ? v.x / keep : 0.f, (m & 0xff00u) ? v.y / keep : 0.f,
                       (m & 0xff0000u) ? v.z / keep : 0.f, (m & 0xff000000u) ? v.w / keep : 0.f);
}

template <int VPL, bool NT, bool NTS = NT, bool DROP = false>
__global__ void __launch_bounds__(256)
layernorm_fwd_kernel(const float *__restrict__ x, const float *__restrict__ gamma, const float *__restrict__ beta,
                     float eps, long rows, int d, float *__restrict__ z, float *__restrict__ mean_out,
                     float *__restrict__ rstd_out, const unsigned char *__restrict__ mask = nullptr, float keep = 1.f) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nvec = d >> 2;
    float4 v[VPL];
    load_row<VPL, NT>(x + row * d, nvec, lane, v, 0.f);
    if (DROP) {
        const unsigned *mrow = reinterpret_cast<const unsigned *>(mask + row * d);
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
            const int c = lane + WAVE * j;
            if (c < nvec) v[j] = drop4(v[j], mrow[c], keep);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < VPL; ++j) s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
    const float mean = wave_sum(s) / (float)d;
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        if (lane + WAVE * j < nvec) {
            const float a = v[j].x - mean, b = v[j].y - mean, c = v[j].z - mean, e = v[j].w - mean;
            ss += (a * a + b * b) + (c * c + e * e);
        }
    }
    const float var = wave_sum(ss) / (float)d;          // biased, as np.var
    const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        const int c = lane + WAVE * j;
        if (c < nvec) {
            const float4 gm = reinterpret_cast<const float4 *>(gamma)[c];
            const float4 bt = reinterpret_cast<const float4 *>(beta)[c];
            v[j].x = gm.x * ((v[j].x - mean) * rstd) + bt.x;
            v[j].y = gm.y * ((v[j].y - mean) * rstd) + bt.y;
            v[j].z = gm.z * ((v[j].z - mean) * rstd) + bt.z;
            v[j].w = gm.w * ((v[j].w - mean) * rstd) + bt.w;
        }
    }
    store_row<VPL, NTS>(z + row * d, nvec, lane, v);
    if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
}

__global__ void __launch_bounds__(256)
layernorm_fwd_generic(const float *__restrict__ x, const float *__restrict__ gamma, const float *__restrict__ beta,
                      float eps, long rows, long d, float *__restrict__ z, float *__restrict__ mean_out,
                      float *__restrict__ rstd_out) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *xr = x + row * d;
    float s = 0.f;
    for (long c = lane; c < d; c += WAVE) s += xr[c];
    const float mean = wave_sum(s) / (float)d;
    float ss = 0.f;
    for (long c = lane; c < d; c += WAVE) { const float a = xr[c] - mean; ss += a * a; }
    const float var = wave_sum(ss) / (float)d;
    const float rstd = 1.0f / sqrtf(var + eps);
    for (long c = lane; c < d; c += WAVE) z[row * d + c] = gamma[c] * ((xr[c] - mean) * rstd) + beta[c];
    if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
}

// Backward: each wave walks rows (grid-stride) and keeps its dgamma/dbeta partials for the
// columns it owns in registers; every wave writes one partial row and a column sum over
// the wave partials (fixed order, reproducible) finishes the job.
// DROP: x is the input of the DropOut in front of this LayerNorm: the row is dropped again on the way in (as the forward
// did) and the input gradient goes through DropOut.backward (normalizations.py:27-30) on the way out, before the residual.
template <int VPL, bool NT, bool NTS = NT, bool DROP = false>
__global__ void __launch_bounds__(256)
layernorm_bwd_kernel(const float *__restrict__ dz, const float *__restrict__ x, const float *__restrict__ mean,
                     const float *__restrict__ rstd, const float *__restrict__ gamma,
                     const float *__restrict__ residual, long rows, int d, float *__restrict__ dx,
                     float *__restrict__ part, const unsigned char *__restrict__ mask = nullptr, float keep = 1.f) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nvec = d >> 2;
    float4 gm[VPL], dg[VPL], db[VPL];
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        const int c = lane + WAVE * j;
        gm[j] = (c < nvec) ? reinterpret_cast<const float4 *>(gamma)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        dg[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        db[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float inv_d = 1.0f / (float)d;
    // The residual row is requested together with dz and x, before anything waits (it used to be loaded after the
    // two reductions: 0.459 -> 0.429 ms at 131072 x 1024).  Prefetching the next row's dz / x under the reductions
    // doubled the registers and changed nothing (0.334 -> 0.330 ms): this kernel is not latency-bound.
    for (long row = (long)blockIdx.x * ROWS_PER_BLOCK + wave; row < rows; row += (long)gridDim.x * ROWS_PER_BLOCK) {
        float4 g[VPL], yh[VPL], res[VPL];
        load_row<VPL, NT>(dz + row * d, nvec, lane, g, 0.f);
        load_row<VPL, NT>(x + row * d, nvec, lane, yh, 0.f);
        if (residual) load_row<VPL, NT>(residual + row * d, nvec, lane, res, 0.f);
        unsigned mk[VPL];
        if (DROP) {
            const unsigned *mrow = reinterpret_cast<const unsigned *>(mask + row * d);
#pragma unroll
            for (int j = 0; j < VPL; ++j) {
                const int c = lane + WAVE * j;
                mk[j] = c < nvec ? mrow[c] : 0u;
                yh[j] = drop4(yh[j], mk[j], keep);
            }
        }
        const float mu = mean[row], rs = rstd[row];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
            const bool on = lane + WAVE * j < nvec;
            yh[j].x = on ? (yh[j].x - mu) * rs : 0.f; yh[j].y = on ? (yh[j].y - mu) * rs : 0.f;
            yh[j].z = on ? (yh[j].z - mu) * rs : 0.f; yh[j].w = on ? (yh[j].w - mu) * rs : 0.f;
            db[j].x += g[j].x; db[j].y += g[j].y; db[j].z += g[j].z; db[j].w += g[j].w;
            dg[j].x += g[j].x * yh[j].x; dg[j].y += g[j].y * yh[j].y;
            dg[j].z += g[j].z * yh[j].z; dg[j].w += g[j].w * yh[j].w;
            g[j].x *= gm[j].x; g[j].y *= gm[j].y; g[j].z *= gm[j].z; g[j].w *= gm[j].w;
            s1 += (g[j].x + g[j].y) + (g[j].z + g[j].w);
            s2 += (g[j].x * yh[j].x + g[j].y * yh[j].y) + (g[j].z * yh[j].z + g[j].w * yh[j].w);
        }
        const float m1 = wave_sum(s1) * inv_d, m2 = wave_sum(s2) * inv_d;
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
            const int c = lane + WAVE * j;
            float4 o;
            o.x = rs * (g[j].x - m1 - yh[j].x * m2); o.y = rs * (g[j].y - m1 - yh[j].y * m2);
            o.z = rs * (g[j].z - m1 - yh[j].z * m2); o.w = rs * (g[j].w - m1 - yh[j].w * m2);
            if (c < nvec) {
                if (DROP) o = drop4(o, mk[j], keep);
                if (residual) { o.x += res[j].x; o.y += res[j].y; o.z += res[j].z; o.w += res[j].w; }
                stg4<NTS>(dx + row * d + 4 * c, o);
            }
        }
    }
    // One partial row per BLOCK, dgamma and dbeta side by side ([blocks][2 d]): the waves' partials meet in LDS and
    // wave 0 adds them in wave order (fixed: reproducible); one column sum over the block partials finishes both.
    __shared__ float4 red[3][VPL * WAVE];
    float4 *prow = reinterpret_cast<float4 *>(part + (long)blockIdx.x * 2 * d);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        float4 (&mine)[VPL] = pass == 0 ? dg : db;
        if (pass) __syncthreads();
        if (wave > 0) {
#pragma unroll
            for (int j = 0; j < VPL; ++j) red[wave - 1][lane + WAVE * j] = mine[j];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int j = 0; j < VPL; ++j) {
                const int c = lane + WAVE * j;
                float4 t = mine[j];
#pragma unroll
                for (int w = 0; w < 3; ++w) {
                    const float4 o = red[w][c];
                    t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w;
                }
                if (c < nvec) prow[pass * nvec + c] = t;
            }
        }
    }
}

// generic LayerNorm backward: dx only (any d); dgamma/dbeta then come from two column sums
__global__ void __launch_bounds__(256)
layernorm_bwd_dx_generic(const float *__restrict__ dz, const float *__restrict__ x, const float *__restrict__ mean,
                         const float *__restrict__ rstd, const float *__restrict__ gamma,
                         const float *__restrict__ residual, long rows, long d, float *__restrict__ dx,
                         float *__restrict__ dz_yhat) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float mu = mean[row], rs = rstd[row];
    const float *gr = dz + row * d, *xr = x + row * d;
    float s1 = 0.f, s2 = 0.f;
    for (long c = lane; c < d; c += WAVE) {
        const float yh = (xr[c] - mu) * rs, g = gr[c] * gamma[c];
        s1 += g; s2 += g * yh;
    }
    const float m1 = wave_sum(s1) / (float)d, m2 = wave_sum(s2) / (float)d;
    for (long c = lane; c < d; c += WAVE) {
        const float yh = (xr[c] - mu) * rs, g = gr[c] * gamma[c];
        float o = rs * (g - m1 - yh * m2);
        if (residual) o += residual[row * d + c];
        dx[row * d + c] = o;
        dz_yhat[row * d + c] = gr[c] * yh;
    }
}

}  // namespace

namespace npm {
void set_ln_bwd_blocks(int v) { g_ln_bwd_blocks_per_cu = v > 0 ? v : 4; }
void set_ln_nt_split(int v) { g_ln_nt_split = v; }
void set_stream_nt(int v) { g_stream_nt = v != 0; }
bool stream_nt_enabled(size_t bytes) { return stream_nt(bytes); }
void set_ew_grid_cap(int v) { g_ew_grid_cap = v > 0 ? v : (1 << 20); }
int colsum_launch(const float *x, float *out, long rows, long cols, long ld) { return colsum_impl(x, out, rows, cols, ld); }
}  // namespace npm

// =====================================================================================
extern "C" {

int npm_fill_f32(float *dst, float value, size_t n) {
    NPM_REQUIRE_INIT();
    NPM_ARG(dst != nullptr || n == 0);
    return ew1(dst, dst, n, FillF{value});
}

int npm_relu_fwd(const float *x, float *y, size_t n) {
    NPM_REQUIRE_INIT();
    NPM_ARG((x && y) || n == 0);
    return ew1(x, y, n, ReluF{});
}

int npm_relu_bwd(const float *x_pre, const float *dy, float *dx, size_t n) {
    NPM_REQUIRE_INIT();
    NPM_ARG((x_pre && dy && dx) || n == 0);
    return ew2(x_pre, dy, dx, n, ReluBwdF{});
}

int npm_add(const float *a, const float *b, float *out, size_t n) {
    NPM_REQUIRE_INIT();
    NPM_ARG((a && b && out) || n == 0);
    return ew2(a, b, out, n, AddF{});
}

int npm_add3(const float *a, const float *b, const float *c, float *out, size_t n) {
    NPM_REQUIRE_INIT();
    NPM_ARG((a && b && c && out) || n == 0);
    return ew3(a, b, c, out, n, Add3F{});
}

int npm_axpy(float *y, const float *x, float alpha, size_t n) {
    NPM_REQUIRE_INIT();
    NPM_ARG((x && y) || n == 0);
    return ew2(y, x, y, n, AxpyF{alpha});
}

int npm_scale(const float *x, float *y, float alpha, size_t n) {
    NPM_REQUIRE_INIT();
    NPM_ARG((x && y) || n == 0);
    return ew1(x, y, n, ScaleF{alpha});
}

int npm_colsum(const float *x, float *out, int64_t rows, int64_t cols, int64_t ld) {
    NPM_REQUIRE_INIT();
    NPM_ARG(rows >= 0 && cols >= 0 && ld >= cols);
    if (cols == 0) return NPM_OK;
    NPM_ARG(out != nullptr && (x != nullptr || rows == 0));
    return colsum_impl(x, out, rows, cols, ld);
}

int npm_relu_bwd_colsum(const float *x_pre, const float *dy, float *dx, float *colsum, int64_t rows, int64_t cols) {
    NPM_REQUIRE_INIT();
    NPM_ARG(rows >= 0 && cols >= 0);
    if (cols == 0) return NPM_OK;
    NPM_ARG(colsum != nullptr && ((x_pre && dy && dx) || rows == 0));
    return colsum_run<true>(x_pre, colsum, rows, cols, cols, dy, dx);
}

#define NPM_ROW_DISPATCH_NT(KERNEL, NT, n, ...)                                                          \
    do {                                                                                                 \
        if (n <= 256) hipLaunchKernelGGL((KERNEL<1, NT>), dim3(grid), dim3(256), 0, s, __VA_ARGS__);       \
        else if (n <= 512) hipLaunchKernelGGL((KERNEL<2, NT>), dim3(grid), dim3(256), 0, s, __VA_ARGS__);  \
        else if (n <= 1024) hipLaunchKernelGGL((KERNEL<4, NT>), dim3(grid), dim3(256), 0, s, __VA_ARGS__); \
        else if (n <= 2048) hipLaunchKernelGGL((KERNEL<8, NT>), dim3(grid), dim3(256), 0, s, __VA_ARGS__); \
        else hipLaunchKernelGGL((KERNEL<16, NT>), dim3(grid), dim3(256), 0, s, __VA_ARGS__);               \
    } while (0)
// one wave per row; rows x n floats stream through once: nontemporal when that is far more than the caches hold
#define NPM_ROW_DISPATCH(KERNEL, n, ...)                                                       \
    do {                                                                                       \
        const int grid = (int)((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK);                  \
        if (stream_nt(sizeof(float) * (size_t)rows * (size_t)(n))) NPM_ROW_DISPATCH_NT(KERNEL, true, n, __VA_ARGS__); \
        else NPM_ROW_DISPATCH_NT(KERNEL, false, n, __VA_ARGS__);                               \
    } while (0)

int npm_attn_rowdot(const float *a, const float *b, float *out, int64_t batch, int64_t seq, int64_t heads, int64_t dim) {
    NPM_REQUIRE_INIT();
    NPM_ARG(batch >= 0 && seq >= 0 && heads >= 1 && dim >= 1);
    const long rows = batch * seq * heads;
    if (rows == 0) return NPM_OK;
    NPM_ARG(a && b && out);
    const long blocks = (rows * 32 + 255) / 256;
    NPM_ARG(blocks < (1L << 31));
    NPM_NT_LAUNCH(stream_nt(sizeof(float) * (size_t)rows * (size_t)dim),
                  hipLaunchKernelGGL(attn_rowdot_kernel<NT>, dim3((int)blocks), dim3(256), 0, npm::ctx().stream, a, b, out,
                                     (long)batch, (long)seq, (long)heads, (long)dim));
    NPM_CHECK_LAUNCH();
    return NPM_OK;
}

int npm_softmax_fwd(const float *x, float *y, int64_t rows, int64_t n, float scale) {
    NPM_REQUIRE_INIT();
    NPM_ARG(rows >= 0 && n >= 0);
    if (rows == 0 || n == 0) return NPM_OK;
    NPM_ARG(x != nullptr && y != nullptr);
    NPM_ARG((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK < (1L << 31));
    hipStream_t s = npm::ctx().stream;
    const bool fast = n % 4 == 0 && n <= 4096 && aligned16(x) && aligned16(y);
    if (fast) {
        NPM_ROW_DISPATCH(softmax_fwd_kernel, n, x, y, (long)rows, (int)n, scale);
    } else {
        const int grid = (int)((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK);
        hipLaunchKernelGGL(softmax_fwd_generic, dim3(grid), dim3(256), 0, s, x, y, (long)rows, (long)n, scale);
    }
    NPM_CHECK_LAUNCH();
    return NPM_OK;
}

int npm_softmax_bwd(const float *y, const float *dy, float *dx, int64_t rows, int64_t n, float scale) {
    NPM_REQUIRE_INIT();
    NPM_ARG(rows >= 0 && n >= 0);
    if (rows == 0 || n == 0) return NPM_OK;
    NPM_ARG(y != nullptr && dy != nullptr && dx != nullptr);
    NPM_ARG((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK < (1L << 31));
    hipStream_t s = npm::ctx().stream;
    const bool fast = n % 4 == 0 && n <= 4096 && aligned16(y) && aligned16(dy) && aligned16(dx);
    if (fast) {
        NPM_ROW_DISPATCH(softmax_bwd_kernel, n, y, dy, dx, (long)rows, (int)n, scale);
    } else {
        const int grid = (int)((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK);
        hipLaunchKernelGGL(softmax_bwd_generic, dim3(grid), dim3(256), 0, s, y, dy, dx, (long)rows, (long)n, scale);
    }
    NPM_CHECK_LAUNCH();
    return NPM_OK;
}

#define NPM_ROW_DISPATCH_DROP(KERNEL, NT, n, ...)                                                                    \
    do {                                                                                                            \
        if (n <= 256) hipLaunchKernelGGL((KERNEL<1, NT, NT, true>), dim3(grid), dim3(256), 0, s, __VA_ARGS__);       \
        else if (n <= 512) hipLaunchKernelGGL((KERNEL<2, NT, NT, true>), dim3(grid), dim3(256), 0, s, __VA_ARGS__);  \
        else if (n <= 1024) hipLaunchKernelGGL((KERNEL<4, NT, NT, true>), dim3(grid), dim3(256), 0, s, __VA_ARGS__); \
        else if (n <= 2048) hipLaunchKernelGGL((KERNEL<8, NT, NT, true>), dim3(grid), dim3(256), 0, s, __VA_ARGS__); \
        else hipLaunchKernelGGL((KERNEL<16, NT, NT, true>), dim3(grid), dim3(256), 0, s, __VA_ARGS__);               \
    } while (0)

// The row-in-registers kernels with a dropout mask in front (npm_layernorm_dropout_fwd / _bwd): same hint placement as the
// plain kernels at d in (512, 1024] (NPM_TUNE_LN_NT_SPLIT mode 1: loads only), the default elsewhere.
static bool ln_dropout_ok(int64_t d, const void *mask) { return d % 4 == 0 && d <= 4096 && ((uintptr_t)mask & 3) == 0; }

int npm_layernorm_dropout_fwd(const float *x, const unsigned char *mask, float keep_prob, const float *gamma, const float *beta,
                              float eps, int64_t rows, int64_t d, float *z, float *mean, float *rstd) {
    NPM_REQUIRE_INIT();
    NPM_ARG(rows >= 0 && d >= 1 && keep_prob > 0.f);
    if (rows == 0) return NPM_OK;
    NPM_ARG(x && mask && gamma && beta && z && mean && rstd);
    NPM_ARG((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK < (1L << 31));
    if (!(ln_dropout_ok(d, mask) && aligned16(x) && aligned16(z) && aligned16(gamma) && aligned16(beta)))
        return npm::fail(NPM_E_UNSUPPORTED, "npm_layernorm_dropout_fwd: d %% 4 == 0, d <= 4096 and 16-byte aligned rows (npm_mask_scale + npm_layernorm_fwd otherwise)");
    hipStream_t s = npm::ctx().stream;
    const int grid = (int)((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK);
    const bool nt = stream_nt(sizeof(float) * (size_t)rows * (size_t)d);
    if (nt && (g_ln_nt_split >> 2) == 1 && d > 512 && d <= 1024)
        hipLaunchKernelGGL((layernorm_fwd_kernel<4, true, false, true>), dim3(grid), dim3(256), 0, s, x, gamma, beta, eps, (long)rows, (int)d, z, mean, rstd, mask, keep_prob);
    else if (nt) NPM_ROW_DISPATCH_DROP(layernorm_fwd_kernel, true, d, x, gamma, beta, eps, (long)rows, (int)d, z, mean, rstd, mask, keep_prob);
    else NPM_ROW_DISPATCH_DROP(layernorm_fwd_kernel, false, d, x, gamma, beta, eps, (long)rows, (int)d, z, mean, rstd, mask, keep_prob);
    NPM_CHECK_LAUNCH();
    return NPM_OK;
}

int npm_layernorm_dropout_bwd(const float *dz, const float *x, const unsigned char *mask, float keep_prob, const float *mean,
                              const float *rstd, const float *gamma, const float *residual, int64_t rows, int64_t d,
                              float *dx, float *dgamma, float *dbeta) {
    NPM_REQUIRE_INIT();
    NPM_ARG(rows >= 0 && d >= 1 && keep_prob > 0.f);
    NPM_ARG(dgamma != nullptr && dbeta != nullptr);
    if (rows == 0) {
        int rc = npm_fill_f32(dgamma, 0.f, (size_t)d);
        return rc ? rc : npm_fill_f32(dbeta, 0.f, (size_t)d);
    }
    NPM_ARG(dz && x && mask && mean && rstd && gamma && dx);
    if (!(ln_dropout_ok(d, mask) && aligned16(dz) && aligned16(x) && aligned16(dx) && aligned16(gamma) && (residual == nullptr || aligned16(residual))))
        return npm::fail(NPM_E_UNSUPPORTED, "npm_layernorm_dropout_bwd: d %% 4 == 0, d <= 4096 and 16-byte aligned rows");
    hipStream_t s = npm::ctx().stream;
    const bool nt = stream_nt(sizeof(float) * (size_t)rows * (size_t)d);
    const long row_blocks = (rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
    const int grid = (int)std::min<long>(row_blocks, (long)g_ln_bwd_blocks_per_cu * npm::ctx().num_cus);
    npm::Scratch part;
    int rc = part.alloc(sizeof(float) * 2 * (size_t)grid * d);
    if (rc) return rc;
    float *pp = (float *)part.ptr;
    if (nt && (g_ln_nt_split & 3) == 1 && d > 512 && d <= 1024)
        hipLaunchKernelGGL((layernorm_bwd_kernel<4, true, false, true>), dim3(grid), dim3(256), 0, s, dz, x, mean, rstd, gamma, residual, (long)rows, (int)d, dx, pp, mask, keep_prob);
    else if (nt) NPM_ROW_DISPATCH_DROP(layernorm_bwd_kernel, true, d, dz, x, mean, rstd, gamma, residual, (long)rows, (int)d, dx, pp, mask, keep_prob);
    else NPM_ROW_DISPATCH_DROP(layernorm_bwd_kernel, false, d, dz, x, mean, rstd, gamma, residual, (long)rows, (int)d, dx, pp, mask, keep_prob);
    NPM_CHECK_LAUNCH();
    if (dbeta == dgamma + d) return colsum_impl(pp, dgamma, grid, 2 * d, 2 * d);
    rc = colsum_impl(pp, dgamma, grid, d, 2 * d);
    if (rc) return rc;
    return colsum_impl(pp + d, dbeta, grid, d, 2 * d);
}

int npm_layernorm_fwd(const float *x, const float *gamma, const float *beta, float eps,
                      int64_t rows, int64_t d, float *z, float *mean, float *rstd) {
    NPM_REQUIRE_INIT();
    NPM_ARG(rows >= 0 && d >= 1);
    if (rows == 0) return NPM_OK;
    NPM_ARG(x && gamma && beta && z && mean && rstd);
    NPM_ARG((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK < (1L << 31));
    hipStream_t s = npm::ctx().stream;
    const bool fast = d % 4 == 0 && d <= 4096 && aligned16(x) && aligned16(z) && aligned16(gamma) && aligned16(beta);
    const int fwd_mode = g_ln_nt_split >> 2;
    if (fast && fwd_mode && d > 512 && d <= 1024 && stream_nt(sizeof(float) * (size_t)rows * (size_t)d)) {
        const int grid = (int)((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK);
        if (fwd_mode == 1) hipLaunchKernelGGL((layernorm_fwd_kernel<4, true, false>), dim3(grid), dim3(256), 0, s, x, gamma, beta, eps, (long)rows, (int)d, z, mean, rstd);
        else hipLaunchKernelGGL((layernorm_fwd_kernel<4, false, true>), dim3(grid), dim3(256), 0, s, x, gamma, beta, eps, (long)rows, (int)d, z, mean, rstd);
    } else if (fast) {
        NPM_ROW_DISPATCH(layernorm_fwd_kernel, d, x, gamma, beta, eps, (long)rows, (int)d, z, mean, rstd);
    } else {
        const int grid = (int)((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK);
        hipLaunchKernelGGL(layernorm_fwd_generic, dim3(grid), dim3(256), 0, s, x, gamma, beta, eps, (long)rows, (long)d, z, mean, rstd);
    }
    NPM_CHECK_LAUNCH();
    return NPM_OK;
}

int npm_layernorm_bwd(const float *dz, const float *x, const float *mean, const float *rstd,
                      const float *gamma, const float *residual, int64_t rows, int64_t d,
                      float *dx, float *dgamma, float *dbeta) {
    NPM_REQUIRE_INIT();
    NPM_ARG(rows >= 0 && d >= 1);
    NPM_ARG(dgamma != nullptr && dbeta != nullptr);
    if (rows == 0) {
        int rc = npm_fill_f32(dgamma, 0.f, (size_t)d);
        return rc ? rc : npm_fill_f32(dbeta, 0.f, (size_t)d);
    }
    NPM_ARG(dz && x && mean && rstd && gamma && dx);
    hipStream_t s = npm::ctx().stream;
    const bool fast = d % 4 == 0 && d <= 4096 && aligned16(dz) && aligned16(x) && aligned16(dx) &&
                      aligned16(gamma) && (residual == nullptr || aligned16(residual));
    if (fast) {
        const bool nt_rows = stream_nt(sizeof(float) * (size_t)rows * (size_t)d);
        const long row_blocks = (rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
        const int grid = (int)std::min<long>(row_blocks, (long)g_ln_bwd_blocks_per_cu * npm::ctx().num_cus);
        npm::Scratch part;
        int rc = part.alloc(sizeof(float) * 2 * (size_t)grid * d);
        if (rc) return rc;
        float *pp = (float *)part.ptr;                     // [grid][2 d]: dgamma partials | dbeta partials
        if (nt_rows && (g_ln_nt_split & 3) == 1 && d > 512 && d <= 1024)
            hipLaunchKernelGGL((layernorm_bwd_kernel<4, true, false>), dim3(grid), dim3(256), 0, s, dz, x, mean, rstd, gamma, residual, (long)rows, (int)d, dx, pp);
        else if (stream_nt(sizeof(float) * (size_t)rows * (size_t)d) && (g_ln_nt_split & 3) == 2 && d > 512 && d <= 1024)
            hipLaunchKernelGGL((layernorm_bwd_kernel<4, false, true>), dim3(grid), dim3(256), 0, s, dz, x, mean, rstd, gamma, residual, (long)rows, (int)d, dx, pp);
        else if (stream_nt(sizeof(float) * (size_t)rows * (size_t)d))
            NPM_ROW_DISPATCH_NT(layernorm_bwd_kernel, true, d, dz, x, mean, rstd, gamma, residual, (long)rows, (int)d, dx, pp);
        else
            NPM_ROW_DISPATCH_NT(layernorm_bwd_kernel, false, d, dz, x, mean, rstd, gamma, residual, (long)rows, (int)d, dx, pp);
        NPM_CHECK_LAUNCH();
        if (dbeta == dgamma + d) return colsum_impl(pp, dgamma, grid, 2 * d, 2 * d);      // adjacent outputs (the gradient bucket): one pass
        rc = colsum_impl(pp, dgamma, grid, d, 2 * d);
        if (rc) return rc;
        return colsum_impl(pp + d, dbeta, grid, d, 2 * d);
    }
    NPM_ARG((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK < (1L << 31));
    npm::Scratch tmp;
    int rc = tmp.alloc(sizeof(float) * (size_t)rows * d);
    if (rc) return rc;
    const int grid = (int)((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK);
    hipLaunchKernelGGL(layernorm_bwd_dx_generic, dim3(grid), dim3(256), 0, s, dz, x, mean, rstd, gamma, residual,
                       (long)rows, (long)d, dx, (float *)tmp.ptr);
    NPM_CHECK_LAUNCH();
    rc = colsum_impl((const float *)tmp.ptr, dgamma, rows, d, d);
    if (rc) return rc;
    return colsum_impl(dz, dbeta, rows, d, d);
}

}  // extern "C"
