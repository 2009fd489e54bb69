// fp32 GEMM on the f16 matrix pipe: TWO-way split with row scaling, three v_mfma_f32_32x32x16_f16 per product
// (math mode NPM_MATH_F16X2; the three-way bf16 split of npm_mfma_tile.h needs six MFMAs).
//
//   x s = hi + lo,  hi = fp16(x s) rounded to nearest (11 bits), lo = fp16(x s - hi) (the next 11);
//   s = a power of two per ROW of op(A) (index m) and per COLUMN of op(B) (index n) that puts the largest magnitude
//       along K in [2^13, 2^14): an element within 2^-17 of it keeps all 22 bits, a smaller one loses at most 2^-39 of
//       that maximum (its lo part goes subnormal);
//   a b ~ (hi hi + hi lo + lo hi) / (s_a s_b); the dropped lo lo is below 2^-22 |a b|.
// The error contract is row-normwise (what a dot product's error is), not elementwise as for the bf16 split: measured
// against fp64 at K = 4096 the error relative to the largest element of the output row is rms 1.4e-7 -- below the
// 3.8e-7 of a k-ordered fp32 fma chain -- for N(0,1) rows, rows 2^40 apart and elements 2^30 apart inside a row
// (tools/microbench/f16x2_gemm.hip, profiles/r02_f16x2_gemm.log).  hi hi accumulates apart from the two cross terms
// (the matrix pipe cuts small addends to the accumulator's exponent), the two are added once after the K loop.
//
// Structure (the "split once per block" form of tools/microbench/coop_split_gemm.hip): operands go global -> registers,
// are scaled and split by the 256 threads of the block (one unit of 8 consecutive k of one row per thread and operand),
// and land in LDS as f16 planes [128 rows][16 k]; MFMA fragments are plain ds_read_b128.  128 x 128 tile, 4 waves
// 2 x 2, two LDS stages of 16 KB, one barrier per K tile, two blocks per CU.  One pass over each operand finds the
// maxima along K first (row maxima for a K-major operand, column maxima for an MN-major one).
// inf / nan in an operand row poison that row's scale and hence its whole output row / column (a plain fp32 product
// would carry them into the same outputs unless they met zeros).
#include <algorithm>

#include "npm_mfma_tile.h"

namespace {

using namespace npm_tile;

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

constexpr int PLANE = BM * GK * 2;                  // bytes of one f16 plane tile [128 rows][16 k]: 4 KB
constexpr int STAGE = 4 * PLANE;                    // A hi / lo, B hi / lo: 16 KB

// ---- maxima along K -> scales -----------------------------------------------------------------------------------
// K-major operand [rows][k] (row pitch ld): one wave per row
__global__ void __launch_bounds__(256)
rowmax_kernel(const float *__restrict__ x, long ld, long rows, long k, unsigned *__restrict__ umax, int vec) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float *p = x + row * ld;
    float m = 0.f;
    if (vec) {
        for (long c = lane * 4; c < k; c += 256) {
            const float4 v = *reinterpret_cast<const float4 *>(p + c);
            m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
        }
    } else {
        for (long c = lane; c < k; c += 64) m = fmaxf(m, fabsf(p[c]));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if (lane == 0) umax[row] = __float_as_uint(m);
}

// MN-major operand [k][cols] (row pitch ld): blocks of 64 lanes x 4 columns x 4 row lanes over chunks of rows; the
// bit patterns of non-negative floats order like unsigned integers, so atomicMax gives the same result in any order
// SUMS: the same pass also leaves the column sums of its chunk in part[chunk][cols] (the bias gradient that goes with a
// weight gradient, mlp.py:34): summed per thread in row order, across the four row lanes in lane order, and across the
// chunks by the ordinary fixed-order column sum -- reproducible.
template <bool SUMS>
__global__ void __launch_bounds__(256)
colmax_kernel(const float *__restrict__ x, long ld, long krows, long cols, long rows_per_chunk, unsigned *__restrict__ umax, int vec,
              float *__restrict__ part) {
    __shared__ float red[4][256];
    const int cq = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const long c0 = (long)blockIdx.x * 256 + cq * 4;
    const long r_beg = (long)blockIdx.y * rows_per_chunk, r_end = min(krows, r_beg + rows_per_chunk);
    float m[4] = {0.f, 0.f, 0.f, 0.f}, s[4] = {0.f, 0.f, 0.f, 0.f};
    if (vec && c0 + 3 < cols) {
        for (long r = r_beg + rl; r < r_end; r += 4) {
            const float4 v = *reinterpret_cast<const float4 *>(x + r * ld + c0);
            m[0] = fmaxf(m[0], fabsf(v.x)); m[1] = fmaxf(m[1], fabsf(v.y));
            m[2] = fmaxf(m[2], fabsf(v.z)); m[3] = fmaxf(m[3], fabsf(v.w));
            if (SUMS) { s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w; }
        }
    } else {
        for (long r = r_beg + rl; r < r_end; r += 4)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (c0 + j < cols) {
                    const float v = x[r * ld + c0 + j];
                    m[j] = fmaxf(m[j], fabsf(v));
                    if (SUMS) s[j] += v;
                }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) red[rl][cq * 4 + j] = m[j];
    __syncthreads();
    const long c = (long)blockIdx.x * 256 + threadIdx.x;
    if (c < cols) {
        const float v = fmaxf(fmaxf(red[0][threadIdx.x], red[1][threadIdx.x]), fmaxf(red[2][threadIdx.x], red[3][threadIdx.x]));
        atomicMax(umax + c, __float_as_uint(v));
    }
    if (SUMS) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) red[rl][cq * 4 + j] = s[j];
        __syncthreads();
        if (c < cols) part[(long)blockIdx.y * cols + c] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
    }
}

// s = 2^(14 - e) with max = f 2^e, f in [0.5, 1): max s in [2^13, 2^14).  The exponent is cut at 126 so that s and 1 / s stay
// normal fp32 (only rows whose maximum is below 2^-112 are scaled short of the target, down to rows of subnormals);
// the largest finite maximum, e = 128, needs 2^-114.  An all-zero row gets 1.
__global__ void scales_kernel(const unsigned *__restrict__ umax, float *__restrict__ scale, float *__restrict__ inv, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float m = __uint_as_float(umax[i]);
    int e = 0;
    float s = 1.f, r = 1.f;
    if (m > 0.f && m < INFINITY) {
        (void)frexpf(m, &e);
        const int sh = min(126, 14 - e);
        s = ldexpf(1.f, sh);
        r = ldexpf(1.f, -sh);
    } else if (!(m == 0.f)) {                        // inf or nan anywhere in the row: poison it
        s = r = __uint_as_float(0x7fc00000u);
    }
    scale[i] = s;
    inv[i] = r;
}

// ---- the product --------------------------------------------------------------------------------------------------
struct Split2 { u32x4v hi, lo; };

__device__ __forceinline__ Split2 split2(const float (&v)[8], float s) {
    Split2 out;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const float x0 = v[2 * t] * s, x1 = v[2 * t + 1] * s;
        const f16x2 h = __builtin_convertvector(f32x2v{x0, x1}, f16x2);          // v_cvt_pk_f16_f32, round to nearest
        const f16x2 l = __builtin_convertvector(f32x2v{x0 - (float)h.x, x1 - (float)h.y}, f16x2);
        out.hi[t] = __builtin_bit_cast(unsigned, h);
        out.lo[t] = __builtin_bit_cast(unsigned, l);
    }
    return out;
}

__device__ __forceinline__ void mfma_f16(const u32x4v &a, const u32x4v &b, f32x16 &c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// One thread's share of one operand tile (128 rows x 16 k): the 8 consecutive k of one row.
//   K-major  ([rows][K]):  row = tid >> 1, h = tid & 1    -> two 16-byte loads (32 contiguous bytes)
//   MN-major ([K][rows]):  row = tid & 127, h = tid >> 7  -> eight 4-byte loads (a wave reads 256-byte runs)
template <bool KMAJ>
struct Loader {
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned voff, kstep, lds_off;
    long ld;
    float s;
    // base: the operand; first: first row of the tile; extent: rows of the operand; kbeg: first k of this block
    __device__ __forceinline__ void init(const float *base, long ld_, int first, int extent, int kbeg, int K,
                                         const float *__restrict__ scale, int tid) {
        ld = ld_;
        const int row = KMAJ ? tid >> 1 : tid & 127, h = KMAJ ? tid & 1 : tid >> 7;
        const float *panel = KMAJ ? base + (long)first * ld + kbeg : base + (long)kbeg * ld + first;
        const long left = KMAJ ? ((long)(extent - first - 1) * ld + (K - kbeg)) * 4 : ((long)(K - kbeg - 1) * ld + (extent - first)) * 4;
        rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)panel, 0, (int)min(max(left, 0L), 0xFFFFFFFFL), 0x00020000);
        voff = KMAJ ? (unsigned)(row * ld * 4 + h * 32) : (unsigned)((8 * h * ld + row) * 4);
        kstep = KMAJ ? GK * 4u : (unsigned)(GK * ld * 4);
        lds_off = row * 32 + ((h ^ ((row >> 3) & 1)) << 4);
        s = first + row < extent ? scale[first + row] : 1.f;
    }
    __device__ __forceinline__ void load(int kt, float (&v)[8]) const {
        const unsigned so = kt * kstep;
        if (KMAJ) {
            const u32x4v x = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, so, 0);
            const u32x4v y = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + 16, so, 0);
#pragma unroll
            for (int t = 0; t < 4; ++t) { v[t] = __uint_as_float(x[t]); v[4 + t] = __uint_as_float(y[t]); }
        } else {
#pragma unroll
            for (int t = 0; t < 8; ++t) v[t] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, voff + (unsigned)(t * ld * 4), so, 0));
        }
    }
    __device__ __forceinline__ void store(char *planes, const float (&v)[8]) const {
        const Split2 sp = split2(v, s);
        *reinterpret_cast<u32x4v *>(planes + lds_off) = sp.hi;
        *reinterpret_cast<u32x4v *>(planes + PLANE + lds_off) = sp.lo;
    }
};

template <bool A_KMAJ, bool B_KMAJ>
__global__ void __launch_bounds__(NTHREADS, 2)
sgemm_f16x2_kernel(const F16x2Args p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, l32 = lane & 31, half = lane >> 5;

    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int tiles = p.tiles_m * p.tiles_n;
    const int t = logical % tiles, split = logical / tiles;
    int tm, tn;
    tile_coords(t, p.tiles_m, p.tiles_n, p.group_m, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = split * p.k_per_split;
    const int kend = min(p.K, kbeg + p.k_per_split);
    const int nkt = (kend - kbeg) / GK;

    Loader<A_KMAJ> la;
    Loader<B_KMAJ> lb;
    la.init(p.A, p.lda, m0, p.M, kbeg, p.K, p.sa, tid);
    lb.init(p.B, p.ldb, n0, p.N, kbeg, p.K, p.sb, tid);
    // fragment addresses: row = 64 w + 32 i + l32, so (row >> 3) & 1 = (l32 >> 3) & 1
    const int fsw = (half ^ ((l32 >> 3) & 1)) << 4;
    const char *fa = smem + (wm * 64 + l32) * 32 + fsw;
    const char *fb = smem + 2 * PLANE + (wn * 64 + l32) * 32 + fsw;

    f32x16 acc[2][2], cross[2][2];
    zero_acc(acc);
    zero_acc(cross);
    if (nkt > 0) {
        float ga[8], gb[8];
        la.load(0, ga);
        lb.load(0, gb);
        la.store(smem, ga);
        lb.store(smem + 2 * PLANE, gb);
        la.load(1, ga);                                 // past the end: the descriptor's zero or in-range bytes, never stored
        lb.load(1, gb);
        __syncthreads();
        auto mma = [&](int kt) {
            const int st = (kt & 1) * STAGE;
            u32x4v a[2][2], b[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    a[i][pl] = *reinterpret_cast<const u32x4v *>(fa + st + pl * PLANE + i * 1024);
                    b[i][pl] = *reinterpret_cast<const u32x4v *>(fb + st + pl * PLANE + i * 1024);
                }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    mfma_f16(a[i][1], b[j][0], cross[i][j]);
                    mfma_f16(a[i][0], b[j][1], cross[i][j]);
                }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) mfma_f16(a[i][0], b[j][0], acc[i][j]);
        };
        for (int kt = 0; kt + 1 < nkt; ++kt) {              // branch-free body: K tile kt from LDS, kt + 1 split and stored, kt + 2 requested
            mma(kt);
            char *nx = smem + ((kt + 1) & 1) * STAGE;
            la.store(nx, ga);
            lb.store(nx + 2 * PLANE, gb);
            la.load(kt + 2, ga);
            lb.load(kt + 2, gb);
            __syncthreads();
        }
        mma(nkt - 1);
    }
    // back to the operands' scale: C[m, n] /= s_a[m] s_b[n].  The 128 + 128 reciprocals of this tile go through LDS.
    __syncthreads();
    float *rs = reinterpret_cast<float *>(smem);
    rs[tid] = tid < 128 ? (m0 + tid < p.M ? p.inv_sa[m0 + tid] : 0.f) : (n0 + tid - 128 < p.N ? p.inv_sb[n0 + tid - 128] : 0.f);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float cb = rs[128 + wn * 64 + j * 32 + l32];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float ra = rs[wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half];
                acc[i][j][r] = (acc[i][j][r] + cross[i][j][r]) * (ra * cb);
            }
    }
    Epilogue e = p.e;
    const bool raw = p.splits > 1;
    if (raw) e.ws += (long)split * p.slab;
    if (e.buf_ok) write_tile_buf(acc, e, raw, m0, n0, p.M, p.N, wm, wn, l32, half);
    else write_tile(acc, e, raw, m0, n0, p.M, p.N, wm, wn, l32, half);
}

}  // namespace

namespace npm_tile {

int f16x2_scales(const float *x, long ld, bool kmaj, long extent, long k, unsigned *umax, float *scale, float *inv, hipStream_t stream,
                 float *colsum_out) {
    if (extent <= 0) return NPM_OK;
    NPM_ARG(colsum_out == nullptr || !kmaj);
    npm::Scratch part;
    const int vec = (((uintptr_t)x & 15) == 0 && ld % 4 == 0 && (kmaj ? k % 4 == 0 : true)) ? 1 : 0;
    if (kmaj) {
        NPM_ARG((extent + 3) / 4 < (1L << 31));
        hipLaunchKernelGGL(rowmax_kernel, dim3((int)((extent + 3) / 4)), dim3(256), 0, stream, x, ld, extent, k, umax, vec);
    } else {
        NPM_HIP(hipMemsetAsync(umax, 0, sizeof(unsigned) * (size_t)extent, stream));
        const long strips = (extent + 255) / 256;
        long chunks = std::max<long>(1, std::min<long>((k + 63) / 64, std::max<long>(1, 8192 / strips)));
        const long rpc = (k + chunks - 1) / chunks;
        chunks = (k + rpc - 1) / rpc;
        NPM_ARG(strips < (1L << 31) && chunks < 65536);
        if (colsum_out) {
            int rc = part.alloc(sizeof(float) * (size_t)chunks * extent);
            if (rc) return rc;
            hipLaunchKernelGGL(colmax_kernel<true>, dim3((int)strips, (int)chunks), dim3(256), 0, stream, x, ld, k, extent, rpc, umax, vec, (float *)part.ptr);
            NPM_CHECK_LAUNCH();
            rc = npm::colsum_launch((const float *)part.ptr, colsum_out, chunks, extent, extent);
            if (rc) return rc;
        } else if (k > 0) {
            hipLaunchKernelGGL(colmax_kernel<false>, dim3((int)strips, (int)chunks), dim3(256), 0, stream, x, ld, k, extent, rpc, umax, vec, (float *)nullptr);
        }
    }
    NPM_CHECK_LAUNCH();
    hipLaunchKernelGGL(scales_kernel, dim3((int)((extent + 255) / 256)), dim3(256), 0, stream, umax, scale, inv, extent);
    NPM_CHECK_LAUNCH();
    return NPM_OK;
}

int launch_f16x2(const F16x2Args &a, hipStream_t stream) {
    const long grid = (long)a.tiles_m * a.tiles_n * a.splits;
    NPM_ARG(grid > 0 && grid < (1L << 31));
    if (a.a_kmaj && a.b_kmaj) hipLaunchKernelGGL((sgemm_f16x2_kernel<true, true>), dim3((int)grid), dim3(NTHREADS), 0, stream, a);
    else if (a.a_kmaj) hipLaunchKernelGGL((sgemm_f16x2_kernel<true, false>), dim3((int)grid), dim3(NTHREADS), 0, stream, a);
    else if (!a.b_kmaj) hipLaunchKernelGGL((sgemm_f16x2_kernel<false, false>), dim3((int)grid), dim3(NTHREADS), 0, stream, a);
    else return npm::fail(NPM_E_UNSUPPORTED, "f16x2 GEMM: trans_a && trans_b is not used by the hot path");
    NPM_CHECK_LAUNCH();
    return NPM_OK;
}

}  // namespace npm_tile
