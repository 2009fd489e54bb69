// Round-6 microbenchmark: fp32 GEMM C[M,N] = A[M,K] * B[N,K]^T from operands PRE-SPLIT into three bf16 planes (hi rounded, mid, lo),
// six v_mfma_f32_32x32x16_bf16 per product -- like tools/microbench/plane_gemm.hip (round 1: 35 % of the bf16 peak, its DMA pieces
// fetched 32 bytes per row out of planes laid out [rows][K]) but with the planes stored TILE-BLOCKED in HBM:
//     plane[pl][kt][row][16 k]   (kt = k / 16; rows padded to a multiple of 128)
// so that the 128 x 16 tile of one plane is 4 KiB of CONTIGUOUS memory: every LDS-DMA piece (64 lanes x 16 B) is 1 KiB contiguous,
// no partial sectors, and the LDS image is lane-linear.  The bank swizzle of the fragment reads (16-byte slot ^ ((row >> 3) & 1)) is
// baked into the stored layout.  Three LDS stages (72 KB: two blocks per CU), counted vmcnt, bare s_barrier, no vector-ALU
// instruction in the K loop.  Also times the split passes (fp32 [rows][K] -> planes, and the transposing form for [K][rows]).
//   hipcc -O3 --offload-arch=gfx950 plane_gemm_blocked.hip -o plane_gemm_blocked && ./plane_gemm_blocked
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

constexpr int GK = 16, TM = 128, TN = 128;
constexpr int PLANE_TILE = TM * GK;                 // bf16 elements per plane tile (4 KB)
constexpr int STAGE = 6 * PLANE_TILE;               // A hi/mid/lo + B hi/mid/lo (24 KB)
#ifndef NSTAGE
#define NSTAGE 3
#endif

__device__ __forceinline__ void split3(float a0, float a1, unsigned &h, unsigned &m, unsigned &l) {
    h = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a0, a1}, bf16x2));
    const float p0 = a0 - __uint_as_float(h << 16), p1 = a1 - __uint_as_float(h & 0xffff0000u);
    const unsigned m0 = __float_as_uint(p0), m1 = __float_as_uint(p1);
    const float q0 = p0 - __uint_as_float(m0 & 0xffff0000u), q1 = p1 - __uint_as_float(m1 & 0xffff0000u);
    m = __builtin_amdgcn_perm(m1, m0, 0x07060302);
    l = __builtin_amdgcn_perm(__float_as_uint(q1), __float_as_uint(q0), 0x07060302);
}

// byte offset of the 16-byte slot holding k = 8 g .. 8 g + 7 (g = k8 index) of `row` inside one plane
__device__ __forceinline__ size_t slot_offset(long rows_pad, long row, long k8) {
    const long kt = k8 >> 1, slot = (k8 & 1) ^ ((row >> 3) & 1);
    return ((size_t)(kt * rows_pad + row) * 2 + slot) * 16;
}

// K-major source [rows][K] (row pitch ld): thread = (row, k8); a wave covers 8 rows x 8 k8
__global__ void __launch_bounds__(256) split_kmajor(const float *__restrict__ src, long ld, long rows, long K, long rows_pad,
                                                      unsigned char *__restrict__ planes, size_t plane_bytes) {
    const long kblocks = K / 64, total = (rows + 31) / 32 * kblocks * 256;          // units of 32 rows x 64 k = 256 threads
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const long unit = t >> 8, kb = unit % kblocks, rbk = unit / kblocks;
        const long row = rbk * 32 + ((t >> 3) & 31), k8 = kb * 8 + (t & 7);
        if (row >= rows) continue;
        const float4 x = *reinterpret_cast<const float4 *>(src + row * ld + 8 * k8), y = *reinterpret_cast<const float4 *>(src + row * ld + 8 * k8 + 4);
        unsigned h[4], m[4], l[4];
        split3(x.x, x.y, h[0], m[0], l[0]); split3(x.z, x.w, h[1], m[1], l[1]);
        split3(y.x, y.y, h[2], m[2], l[2]); split3(y.z, y.w, h[3], m[3], l[3]);
        const size_t o = slot_offset(rows_pad, row, k8);
        *reinterpret_cast<u32x4 *>(planes + o) = u32x4{h[0], h[1], h[2], h[3]};
        *reinterpret_cast<u32x4 *>(planes + plane_bytes + o) = u32x4{m[0], m[1], m[2], m[3]};
        *reinterpret_cast<u32x4 *>(planes + 2 * plane_bytes + o) = u32x4{l[0], l[1], l[2], l[3]};
    }
}

// MN-major source [K][rows] (row pitch ld): block = 64 k x 64 rows through LDS (reads coalesced along rows, every thread then
// owns 8 consecutive k of one row twice)
__global__ void __launch_bounds__(256) split_mnmajor(const float *__restrict__ src, long ld, long rows, long K, long rows_pad,
                                                       unsigned char *__restrict__ planes, size_t plane_bytes) {
    __shared__ float tile[64][65];
    const long row_blocks = (rows + 63) / 64, kblocks = K / 64;
    for (long blk = blockIdx.x; blk < row_blocks * kblocks; blk += gridDim.x) {
        const long rb = blk % row_blocks, kb = blk / row_blocks;
        __syncthreads();
        for (int i = threadIdx.x; i < 64 * 16; i += 256) {
            const int kk = i / 16, c4 = i % 16;
            const long r0 = rb * 64 + 4 * c4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r0 + 3 < rows) v = *reinterpret_cast<const float4 *>(src + (kb * 64 + kk) * ld + r0);
            tile[kk][4 * c4] = v.x; tile[kk][4 * c4 + 1] = v.y; tile[kk][4 * c4 + 2] = v.z; tile[kk][4 * c4 + 3] = v.w;
        }
        __syncthreads();
        for (int u = threadIdx.x; u < 64 * 8; u += 256) {
            const int r = u % 64, g = u / 64;                         // row r of the block, k8 group g (8 per 64 k)
            const long row = rb * 64 + r;
            if (row >= rows) continue;
            unsigned h[4], m[4], l[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) split3(tile[8 * g + 2 * e][r], tile[8 * g + 2 * e + 1][r], h[e], m[e], l[e]);
            const size_t o = slot_offset(rows_pad, row, kb * 8 + g);
            *reinterpret_cast<u32x4 *>(planes + o) = u32x4{h[0], h[1], h[2], h[3]};
            *reinterpret_cast<u32x4 *>(planes + plane_bytes + o) = u32x4{m[0], m[1], m[2], m[3]};
            *reinterpret_cast<u32x4 *>(planes + 2 * plane_bytes + o) = u32x4{l[0], l[1], l[2], l[3]};
        }
    }
}

__device__ __forceinline__ void mfma(const u32x4 &a, const u32x4 &b, f32x16 &c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

__global__ void __launch_bounds__(256, 2)
plane_gemm(const unsigned char *__restrict__ Ap, size_t plane_a, long rows_a, const unsigned char *__restrict__ Bp, size_t plane_b, long rows_b,
           float *__restrict__ C, int M, int N, int K) {
    __shared__ __attribute__((aligned(16))) unsigned short smem[NSTAGE * STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, l32 = lane & 31, half = lane >> 5;
    const int tiles_n = N / TN;
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, q = nwg >> 3, r = nwg & 7;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    const int per_group = 8 * tiles_n, gid = logical / per_group, first = gid * 8;
    const int rows_in = min(8, M / TM - first), in = logical - gid * per_group;
    const int tm = first + in % rows_in, tn = in / rows_in;
    const int m0 = tm * TM, n0 = tn * TN;
    const int nkt = K / GK;

    // one descriptor per plane, rebased to this block's first row of K tile 0; K tile kt is kt * rows_pad * 32 bytes further
    __amdgpu_buffer_rsrc_t ra[3], rb[3];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        ra[pl] = __builtin_amdgcn_make_buffer_rsrc((void *)(Ap + pl * plane_a + (size_t)m0 * 32), 0, (int)(plane_a - (size_t)m0 * 32), 0x00020000);
        rb[pl] = __builtin_amdgcn_make_buffer_rsrc((void *)(Bp + pl * plane_b + (size_t)n0 * 32), 0, (int)(plane_b - (size_t)n0 * 32), 0x00020000);
    }
    const unsigned voff = (unsigned)(wave * 1024 + lane * 16);        // wave w moves piece w (rows 32 w .. 32 w + 31) of every plane tile
    const unsigned astep = (unsigned)(rows_a * 32), bstep = (unsigned)(rows_b * 32);
    auto issue = [&](int kt, int stage) {
        unsigned short *base = smem + stage * STAGE + wave * 512;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra[pl], (lds_void *)(base + pl * PLANE_TILE), 16, voff, kt * astep, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb[pl], (lds_void *)(base + (3 + pl) * PLANE_TILE), 16, voff, kt * bstep, 0, 0);
        }
    };
    f32x16 acc[2][2] = {}, small[2][2] = {};
    const int arow = wm * 64 + l32, brow = wn * 64 + l32;
    const int fsw = (half ^ ((l32 >> 3) & 1)) * 8;                    // (row >> 3) & 1 == (l32 >> 3) & 1 for row = 64 w + 32 i + l32
    issue(0, 0);
    if (NSTAGE == 3 && nkt > 1) issue(1, 1);
    int stage = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        if (NSTAGE == 3) {
            if (kt + 1 < nkt) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (kt + 2 < nkt) issue(kt + 2, stage == 0 ? 2 : stage - 1);
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (kt + 1 < nkt) issue(kt + 1, stage ^ 1);
        }
        const unsigned short *st = smem + stage * STAGE;
        stage = stage + 1 == NSTAGE ? 0 : stage + 1;
        u32x4 a[2][3], b[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                a[i][pl] = *reinterpret_cast<const u32x4 *>(st + pl * PLANE_TILE + (arow + 32 * i) * GK + fsw);
                b[i][pl] = *reinterpret_cast<const u32x4 *>(st + (3 + pl) * PLANE_TILE + (brow + 32 * i) * GK + fsw);
            }
#define TERM(PA, PB, ACC)                                                           \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                   \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) mfma(a[i][PA], b[j][PB], ACC[i][j]);
        TERM(2, 0, small) TERM(0, 2, small) TERM(1, 1, small) TERM(1, 0, small) TERM(0, 1, small)
        TERM(0, 0, acc)
#undef TERM
    }
    float *cbase = C + (size_t)(m0 + wm * 64) * N + n0 + wn * 64;
    const auto rc = __builtin_amdgcn_make_buffer_rsrc((void *)cbase, 0, (int)(64 * (size_t)N * 4), 0x00020000);
    const int vo = (4 * half * N + l32) * 4;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int rr = 0; rr < 16; ++rr)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[i][j][rr] + small[i][j][rr]), rc, vo + j * 128,
                                                      (i * 32 + (rr & 3) + 8 * (rr >> 2)) * N * 4, 0);
}

// ---- hybrid: A fp32 [M][K] staged as in the product kernel (LDS-DMA, K-major tile with the bank swizzle on the source address) and split
// into three bf16 parts in REGISTERS after the LDS read; B from pre-split tile-blocked planes (a weight matrix: split once per call for
// a few microseconds).  Half of the shipped bf16x3 kernel's split work, 20 KB per K step instead of 16 (fp32 both) or 24 (planes both).
__device__ __forceinline__ float sub_f32(float a, float b) { float r; asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ void split8(const float (&v)[8], u32x4 &hi, u32x4 &mid, u32x4 &lo) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[2 * t], v[2 * t + 1]}, bf16x2));
        const float p0 = sub_f32(v[2 * t], __uint_as_float(h << 16)), p1 = sub_f32(v[2 * t + 1], __uint_as_float(h & 0xffff0000u));
        const unsigned m0 = __float_as_uint(p0), m1 = __float_as_uint(p1);
        const float q0 = sub_f32(p0, __uint_as_float(m0 & 0xffff0000u)), q1 = sub_f32(p1, __uint_as_float(m1 & 0xffff0000u));
        hi[t] = h;
        mid[t] = __builtin_amdgcn_perm(m1, m0, 0x07060302);
        lo[t] = __builtin_amdgcn_perm(__float_as_uint(q1), __float_as_uint(q0), 0x07060302);
    }
}

constexpr int HSTAGE = TM * GK * 2 + 3 * PLANE_TILE;      // in 2-byte units: A fp32 tile 8 KB + B planes 12 KB = 20 KB

__global__ void __launch_bounds__(256, 2)
hybrid_gemm(const float *__restrict__ A, const unsigned char *__restrict__ Bp, size_t plane_b, long rows_b, float *__restrict__ C, int M, int N, int K) {
    __shared__ __attribute__((aligned(16))) unsigned short smem[2 * HSTAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, l32 = lane & 31, half = lane >> 5;
    const int tiles_n = N / TN;
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, q = nwg >> 3, r = nwg & 7;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    const int per_group = 8 * tiles_n, gid = logical / per_group, first = gid * 8;
    const int rows_in = min(8, M / TM - first), in = logical - gid * per_group;
    const int tm = first + in % rows_in, tn = in / rows_in;
    const int m0 = tm * TM, n0 = tn * TN;
    const int nkt = K / GK;
    const auto rA = __builtin_amdgcn_make_buffer_rsrc((void *)(A + (size_t)m0 * K), 0, (int)((size_t)TM * K * 4), 0x00020000);
    __amdgpu_buffer_rsrc_t rb[3];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
        rb[pl] = __builtin_amdgcn_make_buffer_rsrc((void *)(Bp + pl * plane_b + (size_t)n0 * 32), 0, (int)(plane_b - (size_t)n0 * 32), 0x00020000);
    unsigned va[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {                                     // A pieces 2 w, 2 w + 1: 16 rows x 64 B each
        const int row = 16 * (2 * wave + i) + (lane >> 2);
        va[i] = (unsigned)(row * K * 4 + (((lane & 3) ^ ((row >> 2) & 3)) * 16));
    }
    const unsigned vb = (unsigned)(wave * 1024 + lane * 16);
    const unsigned bstep = (unsigned)(rows_b * 32);
    auto issue = [&](int kt, int stage) {
        unsigned short *base = smem + stage * HSTAGE;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_void *)(base + (2 * wave + i) * 512), 16, va[i], kt * GK * 4, 0, 0);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb[pl], (lds_void *)(base + TM * GK * 2 + pl * PLANE_TILE + wave * 512), 16, vb, kt * bstep, 0, 0);
    };
    f32x16 acc[2][2] = {}, small[2][2] = {};
    const int arow = wm * 64 + l32, brow = wn * 64 + l32;
    const int fsw = (half ^ ((l32 >> 3) & 1)) * 8;
    const int asw = (arow >> 2) & 3;                                  // same for arow + 32
    issue(0, 0);
    for (int kt = 0; kt < nkt; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt + 1 < nkt) issue(kt + 1, (kt + 1) & 1);
        const unsigned short *st = smem + (kt & 1) * HSTAGE;
        const float *sA = reinterpret_cast<const float *>(st);
        u32x4 a[2][3], b[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float4 x = *reinterpret_cast<const float4 *>(sA + (arow + 32 * i) * GK + ((2 * half) ^ asw) * 4);
            const float4 y = *reinterpret_cast<const float4 *>(sA + (arow + 32 * i) * GK + ((2 * half + 1) ^ asw) * 4);
            const float v[8] = {x.x, x.y, x.z, x.w, y.x, y.y, y.z, y.w};
            split8(v, a[i][0], a[i][1], a[i][2]);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                b[i][pl] = *reinterpret_cast<const u32x4 *>(st + TM * GK * 2 + pl * PLANE_TILE + (brow + 32 * i) * GK + fsw);
        }
#define TERM(PA, PB, ACC)                                                           \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                   \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) mfma(a[i][PA], b[j][PB], ACC[i][j]);
        TERM(2, 0, small) TERM(0, 2, small) TERM(1, 1, small) TERM(1, 0, small) TERM(0, 1, small)
        TERM(0, 0, acc)
#undef TERM
    }
    float *cbase = C + (size_t)(m0 + wm * 64) * N + n0 + wn * 64;
    const auto rc = __builtin_amdgcn_make_buffer_rsrc((void *)cbase, 0, (int)(64 * (size_t)N * 4), 0x00020000);
    const int vo = (4 * half * N + l32) * 4;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int rr = 0; rr < 16; ++rr)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[i][j][rr] + small[i][j][rr]), rc, vo + j * 128,
                                                      (i * 32 + (rr & 3) + 8 * (rr >> 2)) * N * 4, 0);
}

static long pad128(long x) { return (x + 127) / 128 * 128; }

int main() {
    // ---- correctness on a small problem against fp64: A K-major [M][K], B given as [K][N] (the transposing split) --------------
    {
        const int M = 256, N = 384, K = 512;
        std::vector<float> a((size_t)M * K), b((size_t)K * N), c((size_t)M * N);
        srand(1);
        auto gauss = []() { double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0); return (float)(std::sqrt(-2 * std::log(u)) * std::cos(6.283185307179586 * v)); };
        for (auto &x : a) x = gauss();
        for (auto &x : b) x = gauss() / 16;
        float *da, *db, *dc; unsigned char *pa, *pb;
        const size_t pla = (size_t)pad128(M) * K * 2, plb = (size_t)pad128(N) * K * 2;
        CK(hipMalloc(&da, a.size() * 4)); CK(hipMalloc(&db, b.size() * 4)); CK(hipMalloc(&dc, c.size() * 4));
        CK(hipMalloc(&pa, 3 * pla)); CK(hipMalloc(&pb, 3 * plb));
        CK(hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice));
        split_kmajor<<<256, 256>>>(da, K, M, K, pad128(M), pa, pla);
        split_mnmajor<<<256, 256>>>(db, N, N, K, pad128(N), pb, plb);
        plane_gemm<<<(M / TM) * (N / TN), 256>>>(pa, pla, pad128(M), pb, plb, pad128(N), dc, M, N, K);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(c.data(), dc, c.size() * 4, hipMemcpyDeviceToHost));
        double se = 0, me = 0, mx = 0;
        for (int i = 0; i < M; ++i)
            for (int j = 0; j < N; ++j) {
                double ref = 0;
                for (int k = 0; k < K; ++k) ref += (double)a[(size_t)i * K + k] * b[(size_t)k * N + j];
                const double e = c[(size_t)i * N + j] - ref;
                se += e * e; me += e; mx = std::fmax(mx, std::fabs(ref));
            }
        printf("check: rms err %.3e, mean err %+.3e, max|ref| %.3g (NSTAGE %d)\n", std::sqrt(se / (M * N)), me / (M * N), mx, NSTAGE);
        hybrid_gemm<<<(M / TM) * (N / TN), 256>>>(da, pb, plb, pad128(N), dc, M, N, K);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(c.data(), dc, c.size() * 4, hipMemcpyDeviceToHost));
        se = me = 0;
        for (int i = 0; i < M; ++i)
            for (int j = 0; j < N; ++j) {
                double ref = 0;
                for (int k = 0; k < K; ++k) ref += (double)a[(size_t)i * K + k] * b[(size_t)k * N + j];
                const double e = c[(size_t)i * N + j] - ref;
                se += e * e; me += e;
            }
        printf("check hybrid (A fp32 split in registers, B planes): rms err %.3e, mean err %+.3e\n", std::sqrt(se / (M * N)), me / (M * N));
        CK(hipFree(da)); CK(hipFree(db)); CK(hipFree(dc)); CK(hipFree(pa)); CK(hipFree(pb));
    }
    // ---- speed at the encoder's GEMM shapes ------------------------------------------------------------------
    const int shapes[4][3] = {{131072, 1024, 1024}, {131072, 4096, 1024}, {131072, 1024, 4096}, {4096, 4096, 4096}};
    for (auto &s : shapes) {
        const int M = s[0], N = s[1], K = s[2];
        float *a32, *b32, *c; unsigned char *pa, *pb;
        const size_t pla = (size_t)pad128(M) * K * 2, plb = (size_t)pad128(N) * K * 2;
        CK(hipMalloc(&a32, (size_t)M * K * 4)); CK(hipMalloc(&b32, (size_t)N * K * 4)); CK(hipMalloc(&c, (size_t)M * N * 4));
        CK(hipMalloc(&pa, 3 * pla)); CK(hipMalloc(&pb, 3 * plb));
        std::vector<float> h((size_t)1 << 22);
        for (auto &x : h) x = (float)((rand() % 2001 - 1000) * 1e-3);
        for (size_t o = 0; o < (size_t)M * K; o += h.size()) CK(hipMemcpy(a32 + o, h.data(), std::min(h.size(), (size_t)M * K - o) * 4, hipMemcpyHostToDevice));
        for (size_t o = 0; o < (size_t)N * K; o += h.size()) CK(hipMemcpy(b32 + o, h.data(), std::min(h.size(), (size_t)N * K - o) * 4, hipMemcpyHostToDevice));
        split_kmajor<<<4096, 256>>>(a32, K, M, K, pad128(M), pa, pla);
        split_kmajor<<<4096, 256>>>(b32, K, N, K, pad128(N), pb, plb);
        CK(hipDeviceSynchronize());
        const int grid = (M / TM) * (N / TN), reps = 5;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        plane_gemm<<<grid, 256>>>(pa, pla, pad128(M), pb, plb, pad128(N), c, M, N, K);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) plane_gemm<<<grid, 256>>>(pa, pla, pad128(M), pb, plb, pad128(N), c, M, N, K);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= reps;
        hybrid_gemm<<<grid, 256>>>(a32, pb, plb, pad128(N), c, M, N, K);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) hybrid_gemm<<<grid, 256>>>(a32, pb, plb, pad128(N), c, M, N, K);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float hms = 0;
        CK(hipEventElapsedTime(&hms, e0, e1));
        hms /= reps;
        printf("M=%d N=%d K=%d: HYBRID (A fp32 split in registers, B planes) %.3f ms = %.1f TFLOP/s fp32-equivalent (%.1f %% of 2516)\n", M, N, K, hms,
               2.0 * M * N * K / hms * 1e-9, 100.0 * 12.0 * M * N * K / hms * 1e-9 / 2516.0);
        float sp[2] = {0, 0};
        for (int form = 0; form < 2; ++form) {           // the split passes over A: K-major source, and the same bytes read as [K][M]
            CK(hipEventRecord(e0));
            for (int i = 0; i < reps; ++i) {
                if (form == 0) split_kmajor<<<8192, 256>>>(a32, K, M, K, pad128(M), pa, pla);
                else split_mnmajor<<<8192, 256>>>(a32, M, M, K, pad128(M), pa, pla);
            }
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            CK(hipEventElapsedTime(&sp[form], e0, e1));
            sp[form] /= reps;
        }
        printf("M=%d N=%d K=%d: GEMM %.3f ms = %.1f TFLOP/s fp32-equivalent (%.0f TF of bf16 MFMA, %.1f %% of 2516); split of A: K-major source %.3f ms (%.0f GB/s), "
               "transposing %.3f ms (%.0f GB/s)\n", M, N, K, ms, 2.0 * M * N * K / ms * 1e-9, 12.0 * M * N * K / ms * 1e-9,
               100.0 * 12.0 * M * N * K / ms * 1e-9 / 2516.0, sp[0], 10.0 * M * K / sp[0] * 1e-6, sp[1], 10.0 * M * K / sp[1] * 1e-6);
        CK(hipFree(a32)); CK(hipFree(b32)); CK(hipFree(c)); CK(hipFree(pa)); CK(hipFree(pb));
    }
    return 0;
}
