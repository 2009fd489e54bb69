// Prototype: fp32 GEMM C[M,N] = A[M,K] * B[N,K]^T with BOTH operands pre-split into three bf16 planes (hi rounded,
// mid, lo: the parts add up to the fp32 value) and the product formed as six v_mfma_f32_32x32x16_bf16 terms -- the
// "split once per tensor" design of DESIGN.md section 8: no vector-ALU work in the GEMM loop, MFMA operands straight
// from LDS.  Measures what that loop sustains; the split itself is a separate HBM-bound pass (timed too).
//   hipcc -O3 --offload-arch=gfx950 plane_gemm.hip -o plane_gemm && ./plane_gemm
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

// fp32 [rows, k] -> three bf16 planes [3][rows][k]
__global__ void split_planes(const float *__restrict__ src, unsigned short *__restrict__ dst, size_t n) {
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 2; i < n; i += (size_t)gridDim.x * blockDim.x * 2) {
        const float a0 = src[i], a1 = src[i + 1];
        const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a0, a1}, bf16x2));
        const float p0 = a0 - __uint_as_float(h << 16), p1 = a1 - __uint_as_float(h & 0xffff0000u);
        const unsigned m0 = __float_as_uint(p0), m1 = __float_as_uint(p1);
        const float q0 = p0 - __uint_as_float(m0 & 0xffff0000u), q1 = p1 - __uint_as_float(m1 & 0xffff0000u);
        reinterpret_cast<unsigned *>(dst)[i / 2] = h;
        reinterpret_cast<unsigned *>(dst + n)[i / 2] = __builtin_amdgcn_perm(m1, m0, 0x07060302);
        reinterpret_cast<unsigned *>(dst + 2 * n)[i / 2] = __builtin_amdgcn_perm(__float_as_uint(q1), __float_as_uint(q0), 0x07060302);
    }
}

__device__ __forceinline__ void mfma(const u32x4 &a, const u32x4 &b, f32x16 &c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <bool TWO_ACC>
__global__ void __launch_bounds__(256, TWO_ACC ? 2 : 3)
plane_gemm(const unsigned short *__restrict__ Ap, const unsigned short *__restrict__ Bp, float *__restrict__ C,
           int M, int N, int K) {
    __shared__ __attribute__((aligned(16))) unsigned short smem[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, l32 = lane & 31, half = lane >> 5;
    const int tiles_n = N / TN;
    // XCD-contiguous remap + groups of 8 tile rows, as in the product kernel
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, q = nwg >> 3, r = nwg & 7;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    const int per_group = 8 * tiles_n, gid = logical / per_group, first = gid * 8;
    const int rows_in = min(8, M / TM - first), in = logical - gid * per_group;
    const int tm = first + in % rows_in, tn = in / rows_in;
    const int m0 = tm * TM, n0 = tn * TN;
    const size_t plane_a = (size_t)M * K, plane_b = (size_t)N * K;
    const int nkt = K / GK;

    // DMA piece j (32 rows x 32 B): lane -> row 32 j + (lane >> 1), 16-byte slot lane & 1 holding global chunk
    // slot ^ ((row >> 3) & 1) (makes the MFMA fragment reads conflict-free)
    unsigned voff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = 32 * j + (lane >> 1);
        voff[j] = (unsigned)(row * K * 2 + (((lane & 1) ^ ((row >> 3) & 1)) * 16));
    }
    __amdgpu_buffer_rsrc_t ra[3], rb[3];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        ra[pl] = __builtin_amdgcn_make_buffer_rsrc((void *)(Ap + pl * plane_a + (size_t)m0 * K), 0, (int)((size_t)TM * K * 2), 0x00020000);
        rb[pl] = __builtin_amdgcn_make_buffer_rsrc((void *)(Bp + pl * plane_b + (size_t)n0 * K), 0, (int)((size_t)TN * K * 2), 0x00020000);
    }
    // 24 pieces per K tile: wave w issues piece w of every plane tile (6 per wave)
    auto issue = [&](int kt, int stage) {
        unsigned short *base = smem + stage * STAGE + wave * 512;           // 512 bf16 = 1 KiB
        const unsigned ko = (unsigned)(kt * GK * 2);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra[pl], (lds_void *)(base + pl * PLANE_TILE), 16, voff[wave], ko, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb[pl], (lds_void *)(base + (3 + pl) * PLANE_TILE), 16, voff[wave], ko, 0, 0);
        }
    };
    auto frag = [&](const unsigned short *tile, int row) {
        return *reinterpret_cast<const u32x4 *>(tile + row * GK + ((half ^ ((row >> 3) & 1)) * 8));
    };

    f32x16 acc[2][2] = {}, small[TWO_ACC ? 2 : 1][TWO_ACC ? 2 : 1] = {};
    const int arow = wm * 64 + l32, brow = wn * 64 + l32;
    issue(0, 0);
    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();
        if (kt + 1 < nkt) issue(kt + 1, (kt + 1) & 1);
        const unsigned short *st = smem + (kt & 1) * STAGE;
        u32x4 a[2][3], b[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                a[i][pl] = frag(st + pl * PLANE_TILE, arow + 32 * i);
                b[i][pl] = frag(st + (3 + pl) * PLANE_TILE, brow + 32 * i);
            }
#define TERM(PA, PB, ACC)                                                           \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                   \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) mfma(a[i][PA], b[j][PB], ACC);
        if (TWO_ACC) {
            TERM(2, 0, small[i][j]) TERM(0, 2, small[i][j]) TERM(1, 1, small[i][j]) TERM(1, 0, small[i][j]) TERM(0, 1, small[i][j])
        } else {
            TERM(2, 0, acc[i][j]) TERM(0, 2, acc[i][j]) TERM(1, 1, acc[i][j]) TERM(1, 0, acc[i][j]) TERM(0, 1, acc[i][j])
        }
        TERM(0, 0, acc[i][j])
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
            for (int j = 0; j < 2; ++j) {
                float v = acc[i][j][rr];
                if (TWO_ACC) v += small[i][j][rr];
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rc, vo + j * 128, (i * 32 + (rr & 3) + 8 * (rr >> 2)) * N * 4, 0);
            }
}

template <bool TWO_ACC>
int bench(int M, int N, int K, const unsigned short *Ap, const unsigned short *Bp, float *C, const float *A32, unsigned short *scratch) {
    const int grid = (M / TM) * (N / TN), reps = 5;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    plane_gemm<TWO_ACC><<<grid, 256>>>(Ap, Bp, C, M, N, K);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) plane_gemm<TWO_ACC><<<grid, 256>>>(Ap, Bp, C, M, N, K);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) split_planes<<<4096, 256>>>(A32, scratch, (size_t)M * K);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float sp = 0;
    CK(hipEventElapsedTime(&sp, e0, e1));
    sp /= reps;
    printf("%s M=%d N=%d K=%d: GEMM %.3f ms = %.1f TFLOP/s fp32-equivalent (%.0f TF of bf16 MFMA, %.1f %% of 2516); split pass of A %.3f ms (%.0f GB/s)\n",
           TWO_ACC ? "two accumulators" : "one accumulator ", M, N, K, ms, 2.0 * M * N * K / ms * 1e-9, 12.0 * M * N * K / ms * 1e-9,
           100.0 * 12.0 * M * N * K / ms * 1e-9 / 2516.0, sp, 10.0 * M * K / sp * 1e-6);
    return 0;
}

int main() {
    // ---- correctness on a small problem against fp64 -------------------------------------------------------
    {
        const int M = 256, N = 256, K = 512;
        std::vector<float> a((size_t)M * K), b((size_t)N * K), c((size_t)M * N);
        srand(1);
        auto gauss = []() { double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0); return (float)(std::sqrt(-2 * std::log(u)) * std::cos(6.283185307179586 * v)); };
        for (auto &x : a) x = gauss();
        for (auto &x : b) x = gauss() / 16;
        float *da, *db, *dc; unsigned short *pa, *pb;
        CK(hipMalloc(&da, a.size() * 4)); CK(hipMalloc(&db, b.size() * 4)); CK(hipMalloc(&dc, c.size() * 4));
        CK(hipMalloc(&pa, a.size() * 6)); CK(hipMalloc(&pb, b.size() * 6));
        CK(hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice));
        split_planes<<<256, 256>>>(da, pa, a.size()); split_planes<<<256, 256>>>(db, pb, b.size());
        for (int two = 0; two < 2; ++two) {
            if (two) plane_gemm<true><<<(M / TM) * (N / TN), 256>>>(pa, pb, dc, M, N, K);
            else plane_gemm<false><<<(M / TM) * (N / TN), 256>>>(pa, pb, dc, M, N, K);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(c.data(), dc, c.size() * 4, hipMemcpyDeviceToHost));
            double se = 0, me = 0, mx = 0;
            for (int i = 0; i < M; ++i)
                for (int j = 0; j < N; ++j) {
                    double ref = 0;
                    for (int k = 0; k < K; ++k) ref += (double)a[(size_t)i * K + k] * b[(size_t)j * K + k];
                    const double e = c[(size_t)i * N + j] - ref;
                    se += e * e; me += e; mx = std::fmax(mx, std::fabs(ref));
                }
            printf("check %s: rms err %.3e, mean err %+.3e, max|ref| %.3g\n", two ? "two accumulators" : "one accumulator ", std::sqrt(se / (M * N)), me / (M * N), mx);
        }
        CK(hipFree(da)); CK(hipFree(db)); CK(hipFree(dc)); CK(hipFree(pa)); CK(hipFree(pb));
    }
    // ---- speed at the encoder's GEMM shapes ------------------------------------------------------------------
    const int shapes[3][3] = {{131072, 1024, 1024}, {131072, 4096, 1024}, {131072, 1024, 4096}};
    for (auto &s : shapes) {
        const int M = s[0], N = s[1], K = s[2];
        float *a32, *b32, *c; unsigned short *pa, *pb;
        CK(hipMalloc(&a32, (size_t)M * K * 4)); CK(hipMalloc(&b32, (size_t)N * K * 4)); CK(hipMalloc(&c, (size_t)M * N * 4));
        CK(hipMalloc(&pa, (size_t)M * K * 6)); CK(hipMalloc(&pb, (size_t)N * K * 6));
        std::vector<float> h((size_t)1 << 22);
        for (auto &x : h) x = (float)((rand() % 2001 - 1000) * 1e-3);
        for (size_t o = 0; o < (size_t)M * K; o += h.size()) CK(hipMemcpy(a32 + o, h.data(), std::min(h.size(), (size_t)M * K - o) * 4, hipMemcpyHostToDevice));
        for (size_t o = 0; o < (size_t)N * K; o += h.size()) CK(hipMemcpy(b32 + o, h.data(), std::min(h.size(), (size_t)N * K - o) * 4, hipMemcpyHostToDevice));
        split_planes<<<4096, 256>>>(a32, pa, (size_t)M * K); split_planes<<<4096, 256>>>(b32, pb, (size_t)N * K);
        CK(hipDeviceSynchronize());
        if (bench<false>(M, N, K, pa, pb, c, a32, pa)) return 1;
        if (bench<true>(M, N, K, pa, pb, c, a32, pa)) return 1;
        CK(hipFree(a32)); CK(hipFree(b32)); CK(hipFree(c)); CK(hipFree(pa)); CK(hipFree(pb));
    }
    return 0;
}
