// Conv2D, NHWC x HWIO, 'SAME' padding, stride 1, odd k, as implicit-im2col GEMMs on the
// fp32 MFMA (reference layers/conv.py:74-194).
//
// The reference pads x (conv.py:97), then adds k*k shifted [NHW,C0]x[C0,C1] matmuls
// (conv.py:101-105); flattening the taps into the contraction index kk = (i*k + j)*C0 + c
// gives ONE GEMM  y[NHW, C1] = A[NHW, k*k*C0] . filt[k*k*C0, C1]  whose A operand is never
// built: the tile loader computes, per 16-byte chunk, which input pixel/channel it is and
// reads x directly (out-of-image taps are zero: the padding).  In NHWC the row index m of the
// GEMM IS the linear pixel index, so a tap is the pixel m + di*W + dj.
//   forward : A = gather(x), K-major (channels contiguous)          B = filt [K, C1]
//   grad_x  : same kernel on dy with the flipped, channel-transposed filter (conv.py:130,153)
//   grad_w  : dw[k*k*C0, C1] = gather(x)^T dy; A is M-major (K = pixel index), split-K over
//             the N*H*W pixels with a fixed-order slab reduction (conv.py:185-194)
// Tile machinery (LDS layouts, MFMA loop, epilogue) is shared with npm_gemm.hip.
#include <algorithm>

#include "npm_mfma_tile.h"

namespace {

using namespace npm_tile;

struct ConvArgs {
    const float *X;      // gathered activation [NB, H, W, C]
    const float *F;      // dense operand: filter [K, N] (fwd) or dy [pixels, N] (grad_w)
    int H, W, C, ks, pad;
    int M, N, K;         // GEMM view
    int tiles_m, tiles_n, splits, k_per_split, group_m;
    int korder;          // forward / grad_x K loop: 0 = taps outermost (kk = tap * C + c), 1 = 16-channel chunks outermost, the
                         // k * k taps of a chunk back to back (NPM_TUNE_CONV_KORDER)
    long slab;
    unsigned w_mul, w_shift, h_mul, h_shift;   // n / W and n / H for 0 <= n < 2^31 as (mulhi(n, mul) >> shift), see fast_div
    Epilogue e;
};

// Division of 0 <= n < 2^31 by a run-time constant d >= 1 as one multiply-high and one shift (the compiler's own sequence
// for `n / d` with d unknown at compile time is ~25 vector-ALU instructions, and every lane of every block divides its
// pixel index by W and by H in the prologue -- beside three co-resident blocks of back-to-back MFMAs each of those
// instructions costs the matrix pipe its issue cycles).  k = 31 + ceil(log2 d), mul = ceil(2^k / d) < 2^32,
// shift = k - 32:  floor(n / d) = (n * mul) >> k exactly, because n (mul d - 2^k) < 2^31 2^ceil(log2 d) = 2^k.
inline void fast_div_setup(unsigned d, unsigned &mul, unsigned &shift) {
    unsigned lg = 0;
    while ((1ull << lg) < d) ++lg;
    const unsigned k = 31 + lg;
    mul = (unsigned)(((1ull << k) + d - 1) / d);
    shift = k - 32;                                   // d = 1: k = 31 -> handled below (shift would be -1)
    if (d == 1) { mul = 0; shift = 0; }
}
__device__ __forceinline__ int fast_div(int n, unsigned d, unsigned mul, unsigned shift) {
    return d == 1 ? n : (int)(__umulhi((unsigned)n, mul) >> shift);
}

// ---- forward / grad_x: A(m, kk) = X[pixel m shifted by tap(kk)][c(kk)] ----------------------
template <bool VEC>
__device__ __forceinline__ void gather_rows(const ConvArgs &p, int m0, int k0, int tid,
                                            const int (&ph)[4], const int (&pw)[4], float4 (&r)[4]) {
    const int kk = k0 + (tid & 7) * 4;
    if (VEC) {
        // C % 4 == 0: the 4 consecutive kk share one tap
        const int tap = kk / p.C, c = kk - tap * p.C;
        const int ti = tap / p.ks, tj = tap - ti * p.ks;
        const int di = ti - p.pad, dj = tj - p.pad;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + (tid >> 3) + 32 * i;
            const int hh = ph[i] + di, ww = pw[i] + dj;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < p.M && kk < p.K && (unsigned)hh < (unsigned)p.H && (unsigned)ww < (unsigned)p.W)
                v = *reinterpret_cast<const float4 *>(p.X + ((long)m + di * p.W + dj) * p.C + c);
            r[i] = v;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + (tid >> 3) + 32 * i;
            float e[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int kj = kk + j;
                const int tap = kj / p.C, c = kj - tap * p.C;
                const int ti = tap / p.ks, tj = tap - ti * p.ks;
                const int di = ti - p.pad, dj = tj - p.pad;
                const int hh = ph[i] + di, ww = pw[i] + dj;
                e[j] = (m < p.M && kj < p.K && (unsigned)hh < (unsigned)p.H && (unsigned)ww < (unsigned)p.W)
                           ? p.X[((long)m + di * p.W + dj) * p.C + c] : 0.f;
            }
            r[i] = make_float4(e[0], e[1], e[2], e[3]);
        }
    }
}

template <bool VEC>
__global__ void __launch_bounds__(NTHREADS)
conv_fwd_kernel(const ConvArgs p) {
    __shared__ __attribute__((aligned(16))) float smem[2 * TILE_FLOATS];
    float *sA = smem, *sB = smem + TILE_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l32 = lane & 31, half = lane >> 5;

    const int t = xcd_remap(blockIdx.x, gridDim.x);
    int tm, tn;
    tile_coords(t, p.tiles_m, p.tiles_n, p.group_m, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;

    int ph[4], pw[4];           // (h, w) of this thread's 4 gather rows
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + (tid >> 3) + 32 * i;
        pw[i] = m % p.W;
        ph[i] = (m / p.W) % p.H;
    }
    const int nkt = (p.K + BK - 1) / BK;
    f32x16 acc[2][2];
    zero_acc(acc);
    float4 ra[4], rb[4];
    gather_rows<VEC>(p, m0, 0, tid, ph, pw, ra);
    load_tile<false, VEC>(p.F, p.N, n0, p.N, 0, p.K, tid, rb);
    store_tile<true>(sA, tid, ra);
    store_tile<false>(sB, tid, rb);
    __syncthreads();
    const int arow = wm * 64 + l32, brow = wn * 64 + l32;
    for (int kt = 0; kt < nkt; ++kt) {
        const bool more = kt + 1 < nkt;
        if (more) {
            gather_rows<VEC>(p, m0, (kt + 1) * BK, tid, ph, pw, ra);
            load_tile<false, VEC>(p.F, p.N, n0, p.N, (kt + 1) * BK, p.K, tid, rb);
        }
        mma_tile<true, false>(sA, sB, arow, brow, half, acc);
        __syncthreads();
        if (more) {
            store_tile<true>(sA, tid, ra);
            store_tile<false>(sB, tid, rb);
        }
        __syncthreads();
    }
    write_tile(acc, p.e, false, m0, n0, p.M, p.N, wm, wn, l32, half);
}

// ---- grad_w: A(kidx = pixel, m' = (tap, c0)) = X[pixel shifted by tap][c0], M-major ---------
template <bool VEC>
__global__ void __launch_bounds__(NTHREADS)
conv_wgrad_kernel(const ConvArgs p) {
    __shared__ __attribute__((aligned(16))) float smem[2 * TILE_FLOATS];
    float *sA = smem, *sB = smem + TILE_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l32 = lane & 31, half = lane >> 5;

    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int tiles = p.tiles_m * p.tiles_n;
    const int t = logical % tiles, split = logical / tiles;
    int tm, tn;
    tile_coords(t, p.tiles_m, p.tiles_n, p.group_m, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = split * p.k_per_split;
    const int kend = min(p.K, kbeg + p.k_per_split);
    const int nkt = (kend - kbeg + BK - 1) / BK;

    // this thread's 4 consecutive m' = (tap, c): fixed for the whole K loop
    const int mq = m0 + (tid & 31) * 4;
    int tdi[4], tdj[4], tc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int mj = mq + j;
        const int tap = mj / p.C;
        tc[j] = mj - tap * p.C;
        const int ti = tap / p.ks;
        tdi[j] = ti - p.pad;
        tdj[j] = tap - ti * p.ks - p.pad;
    }
    auto gather = [&](int k0, float4 (&r)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pix = k0 + (tid >> 5) + 8 * i;
            const int w = pix % p.W, h = (pix / p.W) % p.H;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pix < kend) {
                if (VEC) {
                    const int hh = h + tdi[0], ww = w + tdj[0];
                    if (mq < p.M && (unsigned)hh < (unsigned)p.H && (unsigned)ww < (unsigned)p.W)
                        v = *reinterpret_cast<const float4 *>(p.X + ((long)pix + tdi[0] * p.W + tdj[0]) * p.C + tc[0]);
                } else {
                    float e[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int hh = h + tdi[j], ww = w + tdj[j];
                        e[j] = (mq + j < p.M && (unsigned)hh < (unsigned)p.H && (unsigned)ww < (unsigned)p.W)
                                   ? p.X[((long)pix + tdi[j] * p.W + tdj[j]) * p.C + tc[j]] : 0.f;
                    }
                    v = make_float4(e[0], e[1], e[2], e[3]);
                }
            }
            r[i] = v;
        }
    };

    f32x16 acc[2][2];
    zero_acc(acc);
    float4 ra[4], rb[4];
    gather(kbeg, ra);
    load_tile<false, VEC>(p.F, p.N, n0, p.N, kbeg, kend, tid, rb);
    store_tile<false>(sA, tid, ra);
    store_tile<false>(sB, tid, rb);
    __syncthreads();
    const int arow = wm * 64 + l32, brow = wn * 64 + l32;
    for (int kt = 0; kt < nkt; ++kt) {
        const bool more = kt + 1 < nkt;
        if (more) {
            gather(kbeg + (kt + 1) * BK, ra);
            load_tile<false, VEC>(p.F, p.N, n0, p.N, kbeg + (kt + 1) * BK, kend, tid, rb);
        }
        mma_tile<false, false>(sA, sB, arow, brow, half, acc);
        __syncthreads();
        if (more) {
            store_tile<false>(sA, tid, ra);
            store_tile<false>(sB, tid, rb);
        }
        __syncthreads();
    }
    Epilogue e = p.e;
    if (p.splits > 1) {
        e.ws += (long)split * p.slab;
        write_tile(acc, e, true, m0, n0, p.M, p.N, wm, wn, l32, half);
    } else {
        write_tile(acc, e, false, m0, n0, p.M, p.N, wm, wn, l32, half);
    }
}

// ---- LDS-DMA versions (same pipeline as sgemm_glds_kernel) ------------------------------------
// The gather becomes a per-lane buffer offset: a lane whose tap falls outside the image gets an
// offset beyond the descriptor's range and the hardware returns the zero of the padding.  The
// descriptor is rebased per block to (first pixel - halo), so offsets stay small.
constexpr int OOB_OFFSET = 0x7FFFFFFF;

// forward / grad_x: requires C % 16 == 0 (a 16-deep K tile lies inside ONE tap), N % 4 == 0.
// TALL = false: 128 x 128 block tile, waves 2 x 2.   TALL = true: 256 x 64 block tile, waves 4 x 1 --
// for N <= 64 (grad_x of a layer with <= 64 input channels) a 128-wide tile would idle half the MFMAs.
template <bool TALL, int MATH = 0>
__global__ void __launch_bounds__(NTHREADS, MATH == 2 ? 2 : MATH ? 3 : 4)
conv_fwd_glds_kernel(const ConvArgs p) {
    constexpr int TM = TALL ? 256 : 128, TN = TALL ? 64 : 128;
    constexpr int A_TILE = TM * GK, B_TILE = TN * GK, STAGE = A_TILE + B_TILE;
    constexpr int A_PIECES = TM / 64, B_PIECES = TN / 64;     // 1 KiB DMA pieces per wave and tile
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];
    prio_high(p.e.prio & 1);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = TALL ? wave : wave >> 1, wn = TALL ? 0 : wave & 1, l32 = lane & 31, half = lane >> 5;

    const int t = xcd_remap(blockIdx.x, gridDim.x);
    int tm, tn;
    tile_coords(t, p.tiles_m, p.tiles_n, p.group_m, tm, tn);
    const int m0 = tm * TM, n0 = tn * TN;
    const int halo = p.pad * p.W + p.pad;                   // pixels a tap can reach back / forward
    const long first = (long)m0 - halo;                     // may be negative: never dereferenced
    const long last = min((long)m0 + TM + halo, (long)p.M);
    const auto rsrcA = __builtin_amdgcn_make_buffer_rsrc((void *)(p.X + first * p.C), 0,
                                                         (int)((last - first) * p.C * 4), 0x00020000);
    const long b_left = ((long)(p.K - 1) * p.N + p.N - n0) * 4;
    const auto rsrcB = __builtin_amdgcn_make_buffer_rsrc((void *)(p.F + n0), 0, (int)min(b_left, 0x7FFFFFFEL), 0x00020000);

    // this wave's A pieces: rows 16 j + (lane >> 2), source chunk (lane & 3) ^ swizzle(row)
    int rowv[A_PIECES], ph[A_PIECES], pw[A_PIECES];
    unsigned vbase[A_PIECES];
    // A piece is 16 consecutive pixels.  Unless that run touches an image border (or wraps to the next image row, or
    // leaves the tensor) every lane's tap is in range, and which taps those are is known per piece: bit t of hmask /
    // wmask = "row offset t - pad / column offset t - pad keeps all 16 pixels inside".  The per-lane validity test is
    // then needed for the border pieces only (7 % of the (piece, tap) pairs at 224 x 224) -- a scalar test per piece.
    unsigned hmask[A_PIECES], wmask[A_PIECES];
#pragma unroll
    for (int i = 0; i < A_PIECES; ++i) {
        const int row = 16 * (A_PIECES * wave + i) + (lane >> 2);
        const int c = (lane & 3) ^ ((row >> 2) & 3);
        const int m = m0 + row;
        rowv[i] = m;
        const int mrow = fast_div(m, p.W, p.w_mul, p.w_shift);           // m / W
        pw[i] = m - mrow * p.W;
        ph[i] = mrow - fast_div(mrow, p.H, p.h_mul, p.h_shift) * p.H;
        vbase[i] = (unsigned)((row * p.C + c * 4) * 4);
        const int mp = m0 + 16 * (A_PIECES * wave + i);         // first pixel of the piece (wave-uniform)
        const int prow = fast_div(mp, p.W, p.w_mul, p.w_shift);
        const int pwf = mp - prow * p.W, phf = prow - fast_div(prow, p.H, p.h_mul, p.h_shift) * p.H;
        unsigned hm = 0, wm = 0;
        if (p.ks <= 31 && pwf + 15 < p.W && mp + 15 < p.M) {
            for (int t = 0; t < p.ks; ++t) {
                const int d = t - p.pad;
                hm |= (unsigned)((unsigned)(phf + d) < (unsigned)p.H) << t;
                wm |= (unsigned)(pwf + d >= 0 && pwf + 15 + d < p.W) << t;
            }
        }
        hmask[i] = __builtin_amdgcn_readfirstlane(hm);
        wmask[i] = __builtin_amdgcn_readfirstlane(wm);
    }
    // B pieces ([16 k][TN] rows of TN floats): a 1 KiB piece covers 1024 / (4 TN) k rows
    constexpr int ROWS_PER_PIECE = 256 / TN, LANES_PER_ROW = TN / 4;
    unsigned vb[B_PIECES];
#pragma unroll
    for (int i = 0; i < B_PIECES; ++i) {
        const int krow = ROWS_PER_PIECE * (B_PIECES * wave + i) + lane / LANES_PER_ROW;
        vb[i] = (unsigned)(krow * p.N * 4 + (lane % LANES_PER_ROW) * 16);
    }
    const int nkt = p.K / GK;

    int c0 = 0, ti = 0, tj = 0;                             // tap / channel base of the NEXT tile to issue
    // (a macro, not a lambda: hipcc 7.2 fails to emit the host stub of this kernel template when a
    // lambda with these builtins is called from it)
#define NPM_CONV_ISSUE(KT, STG)                                                                              \
    do {                                                                                                     \
        float *sa = smem + (STG) * STAGE + (A_PIECES * wave) * 256;                                          \
        float *sb = smem + (STG) * STAGE + A_TILE + (B_PIECES * wave) * 256;                                 \
        const int di = ti - p.pad, dj = tj - p.pad;                                                          \
        const unsigned soff = (unsigned)(((halo + di * p.W + dj) * p.C + c0) * 4);                           \
        _Pragma("unroll") for (int i = 0; i < A_PIECES; ++i) {                                               \
            if ((hmask[i] >> ti) & (wmask[i] >> tj) & 1u) {                                                  \
                lds_dma16(rsrcA, sa + 256 * i, vbase[i], soff);                                              \
            } else {                                                                                         \
                const bool ok = rowv[i] < p.M && (unsigned)(ph[i] + di) < (unsigned)p.H &&                   \
                                (unsigned)(pw[i] + dj) < (unsigned)p.W;                                      \
                asm volatile("; border piece" ::: "memory");                                                 \
                lds_dma16(rsrcA, sa + 256 * i, ok ? vbase[i] : (unsigned)OOB_OFFSET, soff);                  \
            }                                                                                                \
        }                                                                                                    \
        const unsigned kb = (unsigned)((((ti * p.ks + tj) * p.C) + c0) * p.N * 4);   /* filter row (tap, c0) */  \
        _Pragma("unroll") for (int i = 0; i < B_PIECES; ++i)                                                 \
            lds_dma16(rsrcB, sb + 256 * i, vb[i], kb);                                                       \
        if (p.korder) {                                                                                      \
            if (++tj == p.ks) { tj = 0; if (++ti == p.ks) { ti = 0; c0 += GK; } }                            \
        } else {                                                                                             \
            c0 += GK;                                                                                        \
            if (c0 == p.C) { c0 = 0; if (++tj == p.ks) { tj = 0; ++ti; } }                                   \
        }                                                                                                    \
    } while (0)

    f32x16 acc[2][2];
    zero_acc(acc);
    // The bias is the accumulators' INITIAL VALUE (a lane's 16 registers of an MFMA tile are one output column): the 64
    // vector-ALU adds of the epilogue -- issued beside three co-resident blocks of back-to-back MFMAs -- are gone, for the
    // price of the zeroing that happens anyway.  (Sums start from the bias instead of ending with it: rounding order only.)
    Epilogue epi = p.e;
    if (epi.flags & NPM_EPI_BIAS) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + l32;
            const float bj = col < p.N ? epi.bias[col] : 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = bj;
        }
        epi.flags &= ~NPM_EPI_BIAS;
    }
    f32x16 small[MATH == 2 ? 2 : 1][MATH == 2 ? 2 : 1];      // split-bf16 math, mode 2: the small terms (npm_mfma_tile.h)
    if (MATH == 2) zero_acc(reinterpret_cast<f32x16 (&)[2][2]>(small));
    const int arow = wm * 64 + l32, brow = wn * 64 + l32;
    if (nkt > 0) NPM_CONV_ISSUE(0, 0);
    prio_low(p.e.prio & 1);
    // unrolled by two: compile-time LDS stages, no vector-ALU address arithmetic between the MFMAs (npm_gemm.hip)
#define NPM_CONV_TILE(KT, STG)                                                                               \
    do {                                                                                                     \
        dma_barrier();                                                                                       \
        if ((KT) + 1 < nkt) NPM_CONV_ISSUE((KT) + 1, (STG) ^ 1);                                             \
        const float *sA = smem + (STG) * STAGE;                                                              \
        mma_tile16_math<MATH, true, false, TN>(sA, sA + A_TILE, arow, brow, half, acc, reinterpret_cast<f32x16 (&)[2][2]>(small)); \
    } while (0)
    for (int kt = 0; kt < nkt; kt += 2) {
        NPM_CONV_TILE(kt, 0);
        if (kt + 1 < nkt) NPM_CONV_TILE(kt + 1, 1);
    }
#undef NPM_CONV_TILE
#undef NPM_CONV_ISSUE
    if (MATH == 2) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] += reinterpret_cast<f32x16 (&)[2][2]>(small)[i][j];
    }
    prio_high(p.e.prio & 2);
    if (epi.buf_ok) write_tile_buf(acc, epi, false, m0, n0, p.M, p.N, wm, wn, l32, half);
    else write_tile(acc, epi, false, m0, n0, p.M, p.N, wm, wn, l32, half);
}

// grad_w: A(k = pixel, m' = (tap, c)) gathered M-major; requires C % 4 == 0, pixels % 16 == 0.
template <int MATH = 0>
__global__ void __launch_bounds__(NTHREADS, MATH == 2 ? 2 : MATH ? 3 : 4)
conv_wgrad_glds_kernel(const ConvArgs p) {
    __shared__ __attribute__((aligned(16))) float smem[2 * G_STAGE];
    prio_high(p.e.prio & 1);
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

    const int halo = p.pad * p.W + p.pad;
    const long first = (long)kbeg - halo;
    const long last = min((long)kend + halo, (long)p.K);
    const auto rsrcA = __builtin_amdgcn_make_buffer_rsrc((void *)(p.X + first * p.C), 0,
                                                         (int)((last - first) * p.C * 4), 0x00020000);
    const long b_left = ((long)(p.K - kbeg - 1) * p.N + p.N - n0) * 4;
    const auto rsrcB = __builtin_amdgcn_make_buffer_rsrc((void *)(p.F + (long)kbeg * p.N + n0), 0,
                                                         (int)min(b_left, 0x7FFFFFFEL), 0x00020000);
    // this lane's 4 consecutive m' = (tap, c): fixed for the whole K loop
    const int mq = m0 + (lane & 31) * 4;
    const int tap = mq / p.C, c = mq - tap * p.C;
    const int ti = tap / p.ks;
    const int di = ti - p.pad, dj = tap - ti * p.ks - p.pad;
    const bool m_ok = mq < p.M;
    // its two pixel rows per tile: krow = 2 (2 wave + i) + (lane >> 5)
    int ph[2], pw[2];
    unsigned vbase[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int krow = 2 * (2 * wave + i) + (lane >> 5);
        const int pix = kbeg + krow;
        pw[i] = pix % p.W;
        ph[i] = (pix / p.W) % p.H;
        vbase[i] = (unsigned)(((krow + halo + di * p.W + dj) * p.C + c) * 4);
    }
    const int dw_ = GK % p.W, dh_ = GK / p.W;               // advance of (h, w) per 16 pixels
    const unsigned vb0 = glds_voffset<false>(lane, 2 * wave, p.N), vb1 = glds_voffset<false>(lane, 2 * wave + 1, p.N);

    auto issue = [&](int kt, int stage) {
        float *sa = smem + stage * G_STAGE + (2 * wave) * 256;
        float *sb = sa + G_TILE;
        const unsigned ka = (unsigned)(kt * GK * p.C * 4), kb = (unsigned)(kt * GK * p.N * 4);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const bool ok = m_ok && (unsigned)(ph[i] + di) < (unsigned)p.H && (unsigned)(pw[i] + dj) < (unsigned)p.W;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (lds_void *)(sa + 256 * i), 16, ok ? vbase[i] : OOB_OFFSET, ka, 0, 0);
            pw[i] += dw_;
            ph[i] += dh_;
            if (pw[i] >= p.W) { pw[i] -= p.W; ++ph[i]; }
            while (ph[i] >= p.H) ph[i] -= p.H;
        }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, (lds_void *)sb, 16, vb0, kb, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, (lds_void *)(sb + 256), 16, vb1, kb, 0, 0);
    };

    f32x16 acc[2][2];
    zero_acc(acc);
    f32x16 small[MATH == 2 ? 2 : 1][MATH == 2 ? 2 : 1];
    if (MATH == 2) zero_acc(reinterpret_cast<f32x16 (&)[2][2]>(small));
    const int arow = wm * 64 + l32, brow = wn * 64 + l32;
    // k*k*C0 is rarely a multiple of 128: waves whose 64 rows lie past M only feed the DMA pipeline
    const bool wave_has_rows = m0 + wm * 64 < p.M && n0 + wn * 64 < p.N;
    if (nkt > 0) issue(0, 0);
    prio_low(p.e.prio & 1);
    for (int kt = 0; kt < nkt; ++kt) {
        dma_barrier();
        if (kt + 1 < nkt) issue(kt + 1, (kt + 1) & 1);
        const float *sA = smem + (kt & 1) * G_STAGE;
        if (wave_has_rows) mma_tile16_math<MATH, false, false>(sA, sA + G_TILE, arow, brow, half, acc, reinterpret_cast<f32x16 (&)[2][2]>(small));
    }
    if (MATH == 2) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] += reinterpret_cast<f32x16 (&)[2][2]>(small)[i][j];
    }
    prio_high(p.e.prio & 2);
    Epilogue e = p.e;
    const bool raw = p.splits > 1;
    if (raw) e.ws += (long)split * p.slab;
    if (e.buf_ok) write_tile_buf(acc, e, raw, m0, n0, p.M, p.N, wm, wn, l32, half);
    else write_tile(acc, e, raw, m0, n0, p.M, p.N, wm, wn, l32, half);
}

// ---- grad_w with the ReLU backward inside (conv.py:54-56,185-194 in one launch) ---------------------------------
// dw = gather(x)^T g with g = where(pre >= 0, dy, 0) (activations.py:19), db = sum over pixels of g (conv.py:55) and
// g itself written out once for the grad_x convolution.  The standalone ReLU-backward pass (12 B per element of
// [N, H, W, C1]: 3.7 of the 48 ms of config C3) is gone: the DENSE operand of this GEMM is staged through registers
// (global -> registers -> mask -> LDS) instead of by LDS-DMA, which is where the mask is applied; the tile rows take
// turns storing g and summing its columns.  The gathered operand keeps the LDS-DMA path (from inline assembly,
// npm_mfma_tile.h), with the image-border test hoisted out of the lanes: a K tile is 16 consecutive pixels, and
// unless that run touches a border (or wraps to the next image row) every tap of every lane is in range -- a scalar
// test per tile; only border tiles (15 % at 224 x 224) compute per-lane offsets.
// Block tile (64 WR) x 128: waves 2 x 2, WR x 2 MFMA tiles per wave.  WR = 3 gives 192-row tiles: k*k*C0 = 576
// is 3 of them exactly (4.5 tiles of 128 rows left a tenth of the MFMAs multiplying padding).
// where(pre >= 0, dy, 0) on four elements: four compares into SGPR pairs, then four selects.  (Written out because
// hipcc funnels such selects through VCC one at a time, with a wait-state s_nop between each compare and its select.)
__device__ __forceinline__ float4 relu_mask4(const u32x4 &pre, const u32x4 &dy) {
    float4 r;
    unsigned long m0, m1, m2, m3;
    asm("v_cmp_le_f32_e64 %4, 0, %8\n\tv_cmp_le_f32_e64 %5, 0, %9\n\tv_cmp_le_f32_e64 %6, 0, %10\n\tv_cmp_le_f32_e64 %7, 0, %11\n\t"
        "v_cndmask_b32_e64 %0, 0, %12, %4\n\tv_cndmask_b32_e64 %1, 0, %13, %5\n\tv_cndmask_b32_e64 %2, 0, %14, %6\n\tv_cndmask_b32_e64 %3, 0, %15, %7"
        : "=&v"(r.x), "=&v"(r.y), "=&v"(r.z), "=&v"(r.w), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3)
        : "v"(pre.x), "v"(pre.y), "v"(pre.z), "v"(pre.w), "v"(dy.x), "v"(dy.y), "v"(dy.z), "v"(dy.w));
    return r;
}

struct WgradArgs {
    const float *X, *DY, *PRE;
    float *G;            // optional: the masked dy, [pixels, N]
    float *colpart;      // optional: [splits][tiles_m][N] column sums of the g tiles each block produced
    float *out;          // slabs [splits][M][N] (or dw itself when splits == 1)
    int H, W, C, ks, pad;
    int M, N, K;
    int tiles_m, tiles_n, splits, k_per_split;
    long slab;
    KSync ksync;         // rendezvous of the co-resident blocks (npm_mfma_tile.h), slice == null: off
    int parts_m;         // rows of colpart per split: tiles_m, or 1 (TRIO)
};

// TRIO (round 5, WR = 3 with exactly three tile rows -- k*k*C0 = 576 at C3): ONE block of twelve waves carries the three tile rows
// of a (split, column tile) and they share ONE masked dy tile in LDS -- the mask's compares / selects, the dy / pre loads, the g
// stores and the column sums run once per K tile instead of once per tile row (the three blocks of a split were co-resident and
// kept in step by the K rendezvous anyway).  Waves 0-7 stage the dense operand (one 16-byte piece of dy and of pre per thread),
// waves 4 r .. 4 r + 3 gather and multiply tile row r exactly as a block of the four-wave kernel does.
template <int WR, bool TRIO = false>
__global__ void __launch_bounds__(TRIO ? 3 * NTHREADS : NTHREADS, TRIO ? 1 : WR == 3 ? 3 : 4)
conv_wgrad_relu_kernel(const WgradArgs p) {
    constexpr int TM = 64 * WR, A_TILE = TM * GK, B_TILE = BN * GK, STAGE = (TRIO ? 3 : 1) * A_TILE + B_TILE;
    constexpr int PPW = TM / 64;                       // 1 KiB DMA pieces of the gathered tile per wave
    constexpr int CPR = TM / 4;                        // 16-byte chunks per k row of the gathered tile
    constexpr int OOB = 0x7FFFFFFF;
    static_assert(!TRIO || WR == 3, "three tile rows of 192");
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = TRIO ? wave_all & 3 : wave_all;   // within its tile row
    const int trio_row = TRIO ? wave_all >> 2 : 0;
    const bool dense = !TRIO || wave_all < 8;          // this wave stages the dense operand (wave-uniform)
    const int wm = wave >> 1, wn = wave & 1, l32 = lane & 31, half = lane >> 5;

    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int tiles = TRIO ? p.tiles_n : p.tiles_m * p.tiles_n;
    const int t = logical % tiles, split = logical / tiles;
    const int tm = TRIO ? trio_row : t % p.tiles_m, tn = TRIO ? t : t / p.tiles_m;  // the tile rows of one (split, tn) are neighbours: they share dy / pre / x in L2
    const int m0 = tm * TM, n0 = tn * BN;
    const int kbeg = split * p.k_per_split;
    const int kend = min(p.K, kbeg + p.k_per_split);
    const int nkt = (kend - kbeg) / GK;
    // g (and its column sums) is produced once per (split, column tile): the tile rows take turns, K tile kt belongs to
    // tile row kt % tiles_m -- all blocks of the launch are resident at once and end together, so the extra stores and
    // adds must not all fall on the blocks of one tile row.
    int turn = tm;                                     // K tiles until this block's next turn
    bool stored = false;                               // the previous staging issued g stores

    // ---- gathered operand: descriptor over pixels [kbeg - halo, kend + halo)
    const int halo = p.pad * p.W + p.pad;
    const long first = (long)kbeg - halo;
    const long last = min((long)kend + halo, (long)p.K);
    const i32x4_t descA = make_desc(p.X + first * p.C, (last - first) * p.C * 4);
    unsigned vfast[PPW];                               // in-image offsets (or out of range for rows beyond M)
    int kr[PPW], di[PPW], dj[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int L = (wave * PPW + i) * 64 + lane;
        kr[i] = L / CPR;
        const int mq = m0 + 4 * (L - kr[i] * CPR);
        const int tap = mq / p.C, c = mq - tap * p.C;
        const int ti = tap / p.ks;
        di[i] = ti - p.pad;
        dj[i] = tap - ti * p.ks - p.pad;
        vfast[i] = mq < p.M ? (unsigned)(((kr[i] + halo + di[i] * p.W + dj[i]) * p.C + c) * 4) : (unsigned)OOB;
    }
    const unsigned lds_a = lds_offset(smem + trio_row * A_TILE + wave * PPW * 256);
    // first pixel of the next tile to issue, as (row, column) of its image: scalars
    int w0 = __builtin_amdgcn_readfirstlane(kbeg % p.W), h0 = __builtin_amdgcn_readfirstlane((kbeg / p.W) % p.H);

    // ---- dense operand (dy, masked by pre): k rows (tid >> 5) and + 8, columns 4 (tid & 31) ..
    const int ncol = n0 + 4 * (tid & 31);
    const long rows_left = (long)(kend - kbeg);
    const auto rsrcDY = __builtin_amdgcn_make_buffer_rsrc((void *)(p.DY + (long)kbeg * p.N), 0, (int)(rows_left * p.N * 4), 0x00020000);
    const auto rsrcPRE = __builtin_amdgcn_make_buffer_rsrc((void *)(p.PRE + (long)kbeg * p.N), 0, (int)(rows_left * p.N * 4), 0x00020000);
    const auto rsrcG = __builtin_amdgcn_make_buffer_rsrc((void *)((p.G ? p.G : p.out) + (long)kbeg * p.N), 0,
                                                         p.G ? (int)(rows_left * p.N * 4) : 0, 0x00020000);
    const int vb0 = ncol < p.N ? ((tid >> 5) * p.N + ncol) * 4 : OOB;          // (TRIO: tid < 512, k rows 0 .. 15: one piece per thread)
    const int vb1 = ncol < p.N ? (((tid >> 5) + 8) * p.N + ncol) * 4 : OOB;
    float *const sBw = smem + (TRIO ? 3 : 1) * A_TILE + (tid >> 5) * BN + 4 * (tid & 31);      // + stage * STAGE (+ 8 BN for the second row)
    const int tile_bytes = GK * p.N * 4;

    f32x16 acc[WR][2];
#pragma unroll
    for (int i = 0; i < WR; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    float4 colacc = make_float4(0.f, 0.f, 0.f, 0.f);
    u32x4 d0, d1, q0, q1;                               // dy and pre of the tile in flight
    const int arow = wm * 32 * WR + l32, brow = wn * 64 + l32;

    // issue tile KT: the gathered pieces by LDS-DMA into stage STG, the dense operand into registers
#define NPM_WGRAD_ISSUE(KT, STG)                                                                                  \
    do {                                                                                                          \
        const bool interior = w0 + GK <= p.W - p.pad && w0 >= p.pad && h0 >= p.pad && h0 < p.H - p.pad;            \
        const unsigned soff = (unsigned)((KT) * GK * p.C * 4);                                                     \
        if (interior) {                                                                                            \
            dma_group<PPW>(descA, lds_a + (STG) * STAGE * 4, soff, vfast);                                        \
        } else {                                                                                                   \
            unsigned v[PPW];                                                                                       \
            _Pragma("unroll") for (int i = 0; i < PPW; ++i) {                                                      \
                int w = w0 + kr[i], h = h0;                                                                        \
                if (w >= p.W) { w -= p.W; h = h + 1 == p.H ? 0 : h + 1; }                                          \
                const bool ok = (unsigned)(h + di[i]) < (unsigned)p.H && (unsigned)(w + dj[i]) < (unsigned)p.W;    \
                v[i] = ok ? vfast[i] : (unsigned)OOB;                                                              \
            }                                                                                                      \
            dma_group<PPW>(descA, lds_a + (STG) * STAGE * 4, soff, v);                                            \
        }                                                                                                          \
        w0 += GK;                                                                                                  \
        if (w0 >= p.W) { w0 -= p.W; h0 = h0 + 1 == p.H ? 0 : h0 + 1; }                                             \
        if (!TRIO) {                                                                                               \
            d0 = __builtin_amdgcn_raw_buffer_load_b128(rsrcDY, vb0, (KT) * tile_bytes, 0);                         \
            d1 = __builtin_amdgcn_raw_buffer_load_b128(rsrcDY, vb1, (KT) * tile_bytes, 0);                         \
            q0 = __builtin_amdgcn_raw_buffer_load_b128(rsrcPRE, vb0, (KT) * tile_bytes, 0);                        \
            q1 = __builtin_amdgcn_raw_buffer_load_b128(rsrcPRE, vb1, (KT) * tile_bytes, 0);                        \
        } else if (dense) {                                                                                        \
            d0 = __builtin_amdgcn_raw_buffer_load_b128(rsrcDY, vb0, (KT) * tile_bytes, 0);                         \
            q0 = __builtin_amdgcn_raw_buffer_load_b128(rsrcPRE, vb0, (KT) * tile_bytes, 0);                        \
        }                                                                                                          \
    } while (0)
    // mask the dense tile in flight (activations.py:19: x >= 0 keeps dy, x = 0 and -0 included), put it into
    // stage STG; tile row 0 also writes it out as g and adds it to its column sums
#define NPM_WGRAD_STAGE(KT, STG)                                                                                   \
    do {                                                                                                           \
        if (TRIO) {                                                                                                \
            stored = dense;                                                                                        \
            if (dense) {                                                                                           \
                const float4 g0 = relu_mask4(q0, d0);                                                              \
                *reinterpret_cast<float4 *>(sBw + (STG) * STAGE) = g0;                                             \
                const u32x4 s0 = {__float_as_uint(g0.x), __float_as_uint(g0.y), __float_as_uint(g0.z), __float_as_uint(g0.w)};  \
                __builtin_amdgcn_raw_buffer_store_b128(s0, rsrcG, vb0, (KT) * tile_bytes, 0);                      \
                colacc.x += g0.x; colacc.y += g0.y; colacc.z += g0.z; colacc.w += g0.w;                            \
            }                                                                                                      \
            break;                                                                                                 \
        }                                                                                                          \
        const float4 g0 = relu_mask4(q0, d0), g1 = relu_mask4(q1, d1);                                             \
        *reinterpret_cast<float4 *>(sBw + (STG) * STAGE) = g0;                                                     \
        *reinterpret_cast<float4 *>(sBw + (STG) * STAGE + 8 * BN) = g1;                                            \
        stored = turn == 0;                                                                                        \
        turn = turn == 0 ? p.tiles_m - 1 : turn - 1;                                                               \
        if (stored) {                                                                                              \
            const u32x4 s0 = {__float_as_uint(g0.x), __float_as_uint(g0.y), __float_as_uint(g0.z), __float_as_uint(g0.w)};  \
            const u32x4 s1 = {__float_as_uint(g1.x), __float_as_uint(g1.y), __float_as_uint(g1.z), __float_as_uint(g1.w)};  \
            __builtin_amdgcn_raw_buffer_store_b128(s0, rsrcG, vb0, (KT) * tile_bytes, 0);                          \
            __builtin_amdgcn_raw_buffer_store_b128(s1, rsrcG, vb1, (KT) * tile_bytes, 0);                          \
            colacc.x += g0.x + g1.x; colacc.y += g0.y + g1.y; colacc.z += g0.z + g1.z; colacc.w += g0.w + g1.w;     \
        }                                                                                                          \
    } while (0)

    // one K tile: wait for it, issue the next one, multiply, stage the next one.  STG is a compile-time constant (the
    // loop is unrolled by two): every LDS address is a loop-invariant register plus an immediate.
#define NPM_WGRAD_TILE(KT, STG)                                                                                    \
    do {                                                                                                           \
        ksync_wait(p.ksync, (KT));                                                                                 \
        /* tile KT is in LDS for every wave (the DMA pieces: vmcnt; the masked rows: lgkmcnt); the other stage is   \
           free.  The g stores of the tile before are this wave's youngest vector-memory operations: they stay in  \
           flight. */                                                                                              \
        if (TRIO && stored) asm volatile("s_waitcnt vmcnt(1) ; npm:wait" ::: "memory");      /* TRIO: ONE g store */ \
        else if (stored) asm volatile("s_waitcnt vmcnt(2) ; npm:wait" ::: "memory");                               \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                      \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                         \
        __builtin_amdgcn_s_barrier();                                                                              \
        asm volatile("" ::: "memory");                                                                             \
        const bool more = (KT) + 1 < nkt;                                                                          \
        if (more) NPM_WGRAD_ISSUE((KT) + 1, (STG) ^ 1);                                                            \
        const float *sA = smem + (STG) * STAGE + trio_row * A_TILE, *sB = smem + (STG) * STAGE + (TRIO ? 3 : 1) * A_TILE;   \
        _Pragma("unroll") for (int g = 0; g < GK / 8; ++g) {                                                       \
            float4 a[WR], b[2];                                                                                    \
            _Pragma("unroll") for (int i = 0; i < WR; ++i) a[i] = read_frag16<false, TM>(sA, arow + 32 * i, g, half);   \
            _Pragma("unroll") for (int j = 0; j < 2; ++j) b[j] = read_frag16<false, BN>(sB, brow + 32 * j, g, half);    \
            _Pragma("unroll") for (int s = 0; s < 4; ++s)                                                          \
            _Pragma("unroll") for (int i = 0; i < WR; ++i)                                                         \
            _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                        \
                const float av = s == 0 ? a[i].x : s == 1 ? a[i].y : s == 2 ? a[i].z : a[i].w;                     \
                const float bv = s == 0 ? b[j].x : s == 1 ? b[j].y : s == 2 ? b[j].z : b[j].w;                     \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);                      \
            }                                                                                                      \
        }                                                                                                          \
        if (more) NPM_WGRAD_STAGE((KT) + 1, (STG) ^ 1);                                                            \
    } while (0)

    if (nkt > 0) {
        NPM_WGRAD_ISSUE(0, 0);
        NPM_WGRAD_STAGE(0, 0);
    }
    for (int kt = 0; kt < nkt; kt += 2) {
        NPM_WGRAD_TILE(kt, 0);
        if (kt + 1 < nkt) NPM_WGRAD_TILE(kt + 1, 1);
    }
#undef NPM_WGRAD_TILE
#undef NPM_WGRAD_ISSUE
#undef NPM_WGRAD_STAGE

    // ---- raw accumulators into this split's slab (pitch N): one buffer store each, row = scalar offset
    const int rows_here = min(p.M - m0, TM);
    float *optr = p.out + (long)split * p.slab + (long)m0 * p.N;
    const auto rsrcO = __builtin_amdgcn_make_buffer_rsrc((void *)optr, 0, (int)(((long)(rows_here - 1) * p.N + p.N) * 4), 0x00020000);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + l32;
        const int vc = col < p.N ? (4 * half * p.N + col) * 4 : OOB;
#pragma unroll
        for (int i = 0; i < WR; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[i][j][r]), rsrcO, vc,
                                                      (wm * 32 * WR + i * 32 + (r & 3) + 8 * (r >> 2)) * p.N * 4, 0);
    }
    // ---- column sums of g over this split's pixels: the 8 k-row groups of the block through LDS, fixed order
    if (p.colpart) {
        constexpr int KROWS = TRIO ? 16 : 8;            // k-row groups of the block's dense staging
        __syncthreads();                                // every wave is done with the operand stages
        if (dense) *reinterpret_cast<float4 *>(smem + (tid >> 5) * BN + 4 * (tid & 31)) = colacc;
        __syncthreads();
        if (tid < BN) {
            float total = 0.f;
#pragma unroll
            for (int r = 0; r < KROWS; ++r) total += smem[r * BN + tid];
            if (n0 + tid < p.N) p.colpart[((long)split * p.parts_m + (TRIO ? 0 : tm)) * p.N + n0 + tid] = total;
        }
    }
}

// out[(ti, tj, c1), c0] = filt[ks-1-ti, ks-1-tj, c0, c1]      (conv.py:130)
__global__ void flip_transpose_filter_kernel(const float *__restrict__ filt, float *__restrict__ out,
                                             int ks, int c0, int c1) {
    const int total = ks * ks * c0 * c1;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int o0 = idx % c0;
        const int o1 = (idx / c0) % c1;
        const int tap = idx / (c0 * c1);
        const int ti = tap / ks, tj = tap - ti * ks;
        out[idx] = filt[((long)((ks - 1 - ti) * ks + (ks - 1 - tj)) * c0 + o0) * c1 + o1];
    }
}

inline bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }

int g_conv_dma = 1;     // tuning knob NPM_TUNE_CONV_DMA
int g_conv_wave_prio = 0;   // NPM_TUNE_GEMM_WAVE_PRIO
int g_conv_math = 0;        // NPM_TUNE_GEMM_MATH
int g_conv_korder = 1;      // NPM_TUNE_CONV_KORDER: 1 (default) 16-channel chunks outermost, 0 taps outermost
int g_wgrad_fused = 1;           // NPM_TUNE_CONV_WGRAD_FUSED: 0 two passes (ReLU backward, then grad_w), 1 fused (tile height picked; three tile rows of 192: one block of twelve waves), 2 / 3 fused with 128- / 192-row tiles in four-wave blocks
int g_wgrad_blocks_per_cu = 0;   // NPM_TUNE_CONV_WGRAD_BLOCKS: 0 pick_splits chooses 3 or 4 blocks per CU, 3 / 4 pins it, -1 the old ceil(3 CUs / tiles)

int run_conv_gemm(const float *x, const float *filt_kn, int nb, int h, int w, int c, int n_out, int ks,
                  const Epilogue &e) {
    const long m = (long)nb * h * w;
    NPM_ARG(m < (1L << 31) - BM);
    ConvArgs a{};
    a.X = x; a.F = filt_kn;
    a.H = h; a.W = w; a.C = c; a.ks = ks; a.pad = ks / 2;
    a.M = (int)m; a.N = n_out; a.K = ks * ks * c;
    a.tiles_m = (a.M + BM - 1) / BM;
    a.tiles_n = (a.N + BN - 1) / BN;
    a.splits = 1; a.k_per_split = a.K; a.group_m = 8;
    a.korder = g_conv_korder;
    fast_div_setup((unsigned)w, a.w_mul, a.w_shift);
    fast_div_setup((unsigned)h, a.h_mul, a.h_shift);
    a.e = e;
    const bool vec = c % 4 == 0 && n_out % 4 == 0 && aligned16(x) && aligned16(filt_kn);
    hipStream_t s = npm::ctx().stream;
    const long halo = (long)a.pad * w + a.pad;
    a.e.buf_ok = g_conv_dma && (256L * n_out + n_out) * 4 < (1L << 31);
    a.e.prio = g_conv_wave_prio;
    const bool dma = g_conv_dma && vec && c % GK == 0 && (256 + 2 * halo) * c * 4 < (1L << 30) &&
                     (long)a.K * n_out * 4 < (1L << 31);
    const bool tall = dma && n_out <= 64 && n_out % 16 == 0 && g_conv_dma != 2;
    if (tall) {
        a.tiles_m = (a.M + 255) / 256;
        a.tiles_n = (a.N + 63) / 64;
    }
    const long grid = (long)a.tiles_m * a.tiles_n;
    NPM_ARG(grid < (1L << 31));
    npm::note_math((tall || dma) ? g_conv_math : 0);
    if (tall && g_conv_math == 2) hipLaunchKernelGGL((conv_fwd_glds_kernel<true, 2>), dim3((int)grid), dim3(NTHREADS), 0, s, a);
    else if (tall && g_conv_math == 1) hipLaunchKernelGGL((conv_fwd_glds_kernel<true, 1>), dim3((int)grid), dim3(NTHREADS), 0, s, a);
    else if (tall) hipLaunchKernelGGL((conv_fwd_glds_kernel<true, 0>), dim3((int)grid), dim3(NTHREADS), 0, s, a);
    else if (dma && g_conv_math == 2) hipLaunchKernelGGL((conv_fwd_glds_kernel<false, 2>), dim3((int)grid), dim3(NTHREADS), 0, s, a);
    else if (dma && g_conv_math == 1) hipLaunchKernelGGL((conv_fwd_glds_kernel<false, 1>), dim3((int)grid), dim3(NTHREADS), 0, s, a);
    else if (dma) hipLaunchKernelGGL((conv_fwd_glds_kernel<false, 0>), dim3((int)grid), dim3(NTHREADS), 0, s, a);
    else if (vec) hipLaunchKernelGGL(conv_fwd_kernel<true>, dim3((int)grid), dim3(NTHREADS), 0, s, a);
    else hipLaunchKernelGGL(conv_fwd_kernel<false>, dim3((int)grid), dim3(NTHREADS), 0, s, a);
    NPM_CHECK_LAUNCH();
    return NPM_OK;
}

}  // namespace

extern "C" int npm_conv_set_dma(int on) { g_conv_dma = on; return NPM_OK; }
extern "C" int npm_conv_set_wgrad_blocks(int per_cu) { g_wgrad_blocks_per_cu = per_cu; return NPM_OK; }
extern "C" int npm_conv_set_wave_prio(int bits) { g_conv_wave_prio = bits; return NPM_OK; }
extern "C" int npm_conv_set_math(int mode) { g_conv_math = mode; return NPM_OK; }
extern "C" int npm_conv_set_korder(int order) { g_conv_korder = order != 0; return NPM_OK; }
extern "C" int npm_conv_set_wgrad_fused(int mode) { g_wgrad_fused = mode; return NPM_OK; }

extern "C" {

int npm_conv2d_fwd(const npm_conv2d *c) {
    NPM_REQUIRE_INIT();
    NPM_ARG(c != nullptr);
    NPM_ARG(c->n >= 0 && c->h >= 1 && c->w >= 1 && c->c_in >= 1 && c->c_out >= 1);
    NPM_ARG(c->ksize >= 1 && c->ksize % 2 == 1);            // conv.py:94
    if (c->n == 0) return NPM_OK;
    NPM_ARG(c->x && c->filt && c->y);
    Epilogue e{};
    e.C = c->y; e.ldc = c->c_out; e.alpha = 1.f;
    if (c->bias) { e.flags |= NPM_EPI_BIAS; e.bias = c->bias; }
    if (c->relu) {
        if (c->pre) { e.flags |= NPM_EPI_RELU_SAVE; e.aux = c->pre; e.ldaux = c->c_out; }
        else e.flags |= NPM_EPI_RELU;
    }
    return run_conv_gemm(c->x, c->filt, c->n, c->h, c->w, c->c_in, c->c_out, c->ksize, e);
}

int npm_conv2d_bwd_x(const float *dy, const float *filt, float *dx,
                     int32_t n, int32_t h, int32_t w, int32_t c_in, int32_t c_out, int32_t ksize) {
    NPM_REQUIRE_INIT();
    NPM_ARG(n >= 0 && h >= 1 && w >= 1 && c_in >= 1 && c_out >= 1 && ksize >= 1 && ksize % 2 == 1);
    if (n == 0) return NPM_OK;
    NPM_ARG(dy && filt && dx);
    npm::Scratch flipped;
    const size_t fsz = (size_t)ksize * ksize * c_in * c_out;
    int rc = flipped.alloc(sizeof(float) * fsz);
    if (rc) return rc;
    hipLaunchKernelGGL(flip_transpose_filter_kernel, dim3((int)std::min<size_t>((fsz + 255) / 256, 1024)), dim3(256), 0,
                       npm::ctx().stream, filt, (float *)flipped.ptr, ksize, c_in, c_out);
    NPM_CHECK_LAUNCH();
    Epilogue e{};
    e.C = dx; e.ldc = c_in; e.alpha = 1.f;
    return run_conv_gemm(dy, (const float *)flipped.ptr, n, h, w, c_out, c_in, ksize, e);
}

int npm_conv2d_bwd_w(const float *dy, const float *x, float *dw,
                     int32_t n, int32_t h, int32_t w, int32_t c_in, int32_t c_out, int32_t ksize) {
    NPM_REQUIRE_INIT();
    NPM_ARG(n >= 0 && h >= 1 && w >= 1 && c_in >= 1 && c_out >= 1 && ksize >= 1 && ksize % 2 == 1);
    NPM_ARG(dw != nullptr);
    const long pixels = (long)n * h * w;
    NPM_ARG(pixels < (1L << 31) - BK);
    if (pixels == 0) return npm_fill_f32(dw, 0.f, (size_t)ksize * ksize * c_in * c_out);
    NPM_ARG(dy && x);
    ConvArgs a{};
    a.X = x; a.F = dy;
    a.H = h; a.W = w; a.C = c_in; a.ks = ksize; a.pad = ksize / 2;
    a.M = ksize * ksize * c_in; a.N = c_out; a.K = (int)pixels;
    a.tiles_m = (a.M + BM - 1) / BM;
    a.tiles_n = (a.N + BN - 1) / BN;
    a.group_m = 8;
    const long tiles = (long)a.tiles_m * a.tiles_n;
    const int nkt = (a.K + BK - 1) / BK;
    int splits = 1;
    // 5 tiles x 154 splits = 770 blocks left a quarter of the CUs with 4 blocks and the rest with 3 (19.9 ms at C3);
    // pick_splits keeps every CU equally full.
    if (tiles < 2L * npm::ctx().num_cus && nkt >= 16) {
        if (g_wgrad_blocks_per_cu < 0) splits = (int)std::min<long>((3L * npm::ctx().num_cus + tiles - 1) / tiles, nkt / 8);
        else splits = pick_splits(tiles, nkt, npm::ctx().num_cus, g_wgrad_blocks_per_cu, g_conv_math == 2 ? 2 : g_conv_math == 1 ? 3 : 4);
    }
    splits = std::max(1, splits);
    const int kt_per = (nkt + splits - 1) / splits;
    splits = (nkt + kt_per - 1) / kt_per;
    a.splits = splits;
    a.k_per_split = kt_per * BK;
    a.e.C = dw; a.e.ldc = c_out; a.e.alpha = 1.f;
    npm::Scratch ws;
    if (splits > 1) {
        a.slab = (long)a.M * a.N;
        int rc = ws.alloc(sizeof(float) * (size_t)a.slab * splits);
        if (rc) return rc;
        a.e.ws = (float *)ws.ptr;
    }
    const bool vec = c_in % 4 == 0 && c_out % 4 == 0 && aligned16(x) && aligned16(dy);
    hipStream_t s = npm::ctx().stream;
    const int grid = (int)(tiles * splits);
    const long halo = (long)a.pad * w + a.pad;
    a.e.buf_ok = g_conv_dma && (256L * a.N + a.N) * 4 < (1L << 31);
    a.e.prio = g_conv_wave_prio;
    const bool dma = g_conv_dma && vec && pixels % GK == 0 && a.k_per_split % GK == 0 &&
                     ((long)a.k_per_split + 2 * halo + GK) * c_in * 4 < (1L << 30) &&
                     (long)a.k_per_split * c_out * 4 < (1L << 30);
    npm::note_math(dma ? g_conv_math : 0);
    if (dma && g_conv_math == 2) hipLaunchKernelGGL(conv_wgrad_glds_kernel<2>, dim3(grid), dim3(NTHREADS), 0, s, a);
    else if (dma && g_conv_math == 1) hipLaunchKernelGGL(conv_wgrad_glds_kernel<1>, dim3(grid), dim3(NTHREADS), 0, s, a);
    else if (dma) hipLaunchKernelGGL(conv_wgrad_glds_kernel<0>, dim3(grid), dim3(NTHREADS), 0, s, a);
    else if (vec) hipLaunchKernelGGL(conv_wgrad_kernel<true>, dim3(grid), dim3(NTHREADS), 0, s, a);
    else hipLaunchKernelGGL(conv_wgrad_kernel<false>, dim3(grid), dim3(NTHREADS), 0, s, a);
    NPM_CHECK_LAUNCH();
    if (splits > 1) {
        ReduceArgs r{};
        r.ws = a.e.ws; r.slab = a.slab; r.splits = splits;
        r.M = a.M; r.N = a.N; r.batch1 = 1;
        r.e = a.e;
        return launch_splitk_reduce(r, s);
    }
    return NPM_OK;
}

/* dw, db and g = where(pre >= 0, dy, 0) of Conv2D.backward (conv.py:54-56 + activations.py:19) in one launch + the
 * slab / column reductions; `g` feeds npm_conv2d_bwd_x.  Shapes the fused kernel does not take (channels not a
 * multiple of 4, pixels not a multiple of 16, images narrower than 16, the split-bf16 math modes) run the two-pass
 * form: npm_relu_bwd_colsum, then npm_conv2d_bwd_w. */
int npm_conv2d_bwd_w_relu(const float *dy, const float *pre, const float *x, float *g, float *dw, float *db,
                          int32_t n, int32_t h, int32_t w, int32_t c_in, int32_t c_out, int32_t ksize) {
    NPM_REQUIRE_INIT();
    NPM_ARG(n >= 0 && h >= 1 && w >= 1 && c_in >= 1 && c_out >= 1 && ksize >= 1 && ksize % 2 == 1);
    NPM_ARG(dw != nullptr && db != nullptr);
    const long pixels = (long)n * h * w;
    NPM_ARG(pixels < (1L << 31) - BK);
    if (pixels == 0) {
        int rc = npm_fill_f32(dw, 0.f, (size_t)ksize * ksize * c_in * c_out);
        return rc ? rc : npm_fill_f32(db, 0.f, (size_t)c_out);
    }
    NPM_ARG(dy && pre && x && g);
    const int m = ksize * ksize * c_in;
    const long halo = (long)(ksize / 2) * w + ksize / 2;
    const bool fused = g_conv_dma && g_conv_math == 0 && g_wgrad_fused && c_in % 4 == 0 && c_out % 4 == 0 && pixels % GK == 0 && w >= GK &&
                       aligned16(x) && aligned16(dy) && aligned16(pre) && aligned16(g) && (long)m * c_out * 4 < (1L << 31);
    if (fused) {
        WgradArgs a{};
        a.X = x; a.DY = dy; a.PRE = pre; a.G = g;
        a.H = h; a.W = w; a.C = c_in; a.ks = ksize; a.pad = ksize / 2;
        a.M = m; a.N = c_out; a.K = (int)pixels;
        // tile height: the one that pads k*k*C0 least (192-row tiles: 3 blocks per CU, 128-row tiles: 4)
        const long pad128 = (long)((m + 127) / 128) * 128, pad192 = (long)((m + 191) / 192) * 192;
        const bool tall = g_wgrad_fused == 3 || (g_wgrad_fused != 2 && pad192 < pad128);
        const int tm_rows = tall ? 192 : 128, resident = tall ? 3 : 4;
        a.tiles_m = (m + tm_rows - 1) / tm_rows;
        a.tiles_n = (c_out + BN - 1) / BN;
        const long tiles = (long)a.tiles_m * a.tiles_n;
        const int nkt = (int)(pixels / GK);
        int splits = 1;
        if (tiles < 2L * npm::ctx().num_cus && nkt >= 16)
            splits = pick_splits(tiles, nkt, npm::ctx().num_cus, g_wgrad_blocks_per_cu > 0 ? g_wgrad_blocks_per_cu : 0, resident);
        splits = std::max(1, splits);
        const int kt_per = (nkt + splits - 1) / splits;
        splits = (nkt + kt_per - 1) / kt_per;
        a.splits = splits;
        a.k_per_split = kt_per * GK;
        const bool fits = ((long)a.k_per_split + 2 * halo + GK) * c_in * 4 < (1L << 30) && (long)a.k_per_split * c_out * 4 < (1L << 30) &&
                          tiles * splits < (1L << 31);
        if (fits) {
            // TRIO: the three tile rows of a split in one block of twelve waves (NPM_TUNE_CONV_WGRAD_FUSED = 3 keeps the four-wave blocks)
            const bool trio = tall && a.tiles_m == 3 && splits > 1 && g_wgrad_fused != 3;
            a.parts_m = trio ? 1 : a.tiles_m;
            npm::Scratch ws, parts;
            int rc = parts.alloc(sizeof(float) * (size_t)splits * a.parts_m * c_out);
            if (rc) return rc;
            a.colpart = (float *)parts.ptr;
            a.slab = (long)m * c_out;
            if (splits > 1) {
                rc = ws.alloc(sizeof(float) * (size_t)a.slab * splits);
                if (rc) return rc;
                a.out = (float *)ws.ptr;
            } else {
                a.out = dw;
            }
            hipStream_t s = npm::ctx().stream;
            npm::note_math(0);
            const int grid = (int)((trio ? a.tiles_n : tiles) * splits);
            // K rendezvous: all blocks resident at once (3 or 4 per CU), every split long enough
            if (!trio && splits > 1 && grid <= (long)resident * npm::ctx().num_cus && npm::ksync_every() > 0 && kt_per >= 2 * npm::ksync_every()) {
                a.ksync.slice = npm::ksync_slice();
                a.ksync.every = npm::ksync_every();
                a.ksync.epochs = (nkt - (splits - 1) * kt_per - 1) / a.ksync.every;
            }
            if (trio) hipLaunchKernelGGL((conv_wgrad_relu_kernel<3, true>), dim3(grid), dim3(3 * NTHREADS), 0, s, a);
            else if (tall) hipLaunchKernelGGL(conv_wgrad_relu_kernel<3>, dim3(grid), dim3(NTHREADS), 0, s, a);
            else hipLaunchKernelGGL(conv_wgrad_relu_kernel<2>, dim3(grid), dim3(NTHREADS), 0, s, a);
            NPM_CHECK_LAUNCH();
            if (splits > 1) {
                ReduceArgs r{};
                r.ws = a.out; r.slab = a.slab; r.splits = splits;
                r.M = m; r.N = c_out; r.batch1 = 1;
                r.e.C = dw; r.e.ldc = c_out; r.e.alpha = 1.f;
                rc = launch_splitk_reduce(r, s);
                if (rc) return rc;
            }
            return npm_colsum(a.colpart, db, (int64_t)splits * a.parts_m, c_out, c_out);       // fixed order over the partial rows
        }
    }
    int rc = npm_relu_bwd_colsum(pre, dy, g, db, pixels, c_out);
    if (rc) return rc;
    return npm_conv2d_bwd_w(g, x, dw, n, h, w, c_in, c_out, ksize);
}

}  // extern "C"
