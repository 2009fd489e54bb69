// Prototype: fp32 GEMM on the bf16 matrix pipe (three-way split, six v_mfma_f32_32x32x16_bf16 per product) where the
// split is done ONCE PER BLOCK, on the way from global memory to LDS: global -> registers -> hi/mid/lo bf16 -> LDS
// planes, and the MFMA fragments are plain ds_read_b128 of those planes.  Against the product kernel (fp32 tiles in
// LDS by LDS-DMA, every wave splits the fragments it reads): half the split instructions (no wave repeats its
// neighbour's rows), no LDS-DMA pieces (60-185 issue cycles each), no fp32 tile in LDS.
//   hipcc -O3 --offload-arch=gfx950 coop_split_gemm.hip -o coop_split_gemm && ./coop_split_gemm
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int GK = 16, TM = 128, TN = 128;
constexpr int PLANE = TM * GK * 2;                  // bytes per plane tile (4 KB): [128 rows][32 B]
constexpr int STAGE = 6 * PLANE;                    // A hi/mid/lo + B hi/mid/lo (24 KB)

__device__ __forceinline__ float sub_f32(float a, float b) {
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

struct Split3 { u32x4 hi, mid, lo; };

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
        out.mid[t] = __builtin_amdgcn_perm(m1, m0, 0x07060302);
        out.lo[t] = __builtin_amdgcn_perm(__float_as_uint(q1), __float_as_uint(q0), 0x07060302);
    }
    return out;
}

__device__ __forceinline__ void mfma(const u32x4 &a, const u32x4 &b, f32x16 &c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// One thread's share of one operand tile (128 rows x 16 k): the 8 consecutive k of one row.
//   K-major  ([rows][K], k contiguous):  row = tid >> 1, h = tid & 1   -> two 16-byte loads
//   MN-major ([K][rows], rows contiguous): row = tid & 127, h = tid >> 7 -> eight 4-byte loads (a wave reads 256 B runs)
template <bool KMAJ>
struct Loader {
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned voff, kstep, lds_off;
    long ld;
    __device__ __forceinline__ void init(const float *panel, long ld_, long bytes_left, int tid) {
        ld = ld_;
        rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)panel, 0, (int)(bytes_left > 0xFFFFFFFFL ? 0xFFFFFFFFL : bytes_left), 0x00020000);
        const int row = KMAJ ? tid >> 1 : tid & 127, h = KMAJ ? tid & 1 : tid >> 7;
        voff = KMAJ ? (unsigned)(row * ld * 4 + h * 32) : (unsigned)((8 * h * ld + row) * 4);
        kstep = KMAJ ? GK * 4u : (unsigned)(GK * ld * 4);
        lds_off = row * 32 + ((h ^ ((row >> 3) & 1)) << 4);
    }
    __device__ __forceinline__ void load(int kt, float (&v)[8]) const {
        const unsigned so = kt * kstep;
        if (KMAJ) {
            const u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, so, 0);
            const u32x4 y = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + 16, so, 0);
#pragma unroll
            for (int t = 0; t < 4; ++t) { v[t] = __uint_as_float(x[t]); v[4 + t] = __uint_as_float(y[t]); }
        } else {
#pragma unroll
            for (int t = 0; t < 8; ++t) v[t] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, voff + (unsigned)(t * ld * 4), so, 0));
        }
    }
    // TIMING ONLY (wrong data): the same bytes in the same number of instructions, but every wave instruction reads
    // 8 rows x 128 B (whole cache lines) instead of 32 rows x 64 B
    __device__ __forceinline__ void load_lines(int kt, float (&v)[8], int tid) const {
        const int lane = tid & 63, wave = tid >> 6;
        const unsigned so = (kt >> 1) * 128u;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = 64 * (kt & 1) + 16 * wave + 8 * i + (lane >> 3);
            const u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (unsigned)(row * ld * 4 + (lane & 7) * 16), so, 0);
#pragma unroll
            for (int t = 0; t < 4; ++t) v[4 * i + t] = __uint_as_float(x[t]);
        }
    }
    __device__ __forceinline__ void store(char *planes, const float (&v)[8]) const {
        const Split3 s = split3(v);
        *reinterpret_cast<u32x4 *>(planes + lds_off) = s.hi;
        *reinterpret_cast<u32x4 *>(planes + PLANE + lds_off) = s.mid;
        *reinterpret_cast<u32x4 *>(planes + 2 * PLANE + lds_off) = s.lo;
    }
};

// C[M, N] = op(A) op(B):  A_KMAJ: A is [M][K] else [K][M];  B_KMAJ: B is [N][K] else [K][N]
// ABL (timing only, results wrong): 4 no LDS stores, 1 no split/store, 2 also no global loads, 3 also no barrier
// NB: blocks per CU (2: the LDS array is padded to 72 KB so that every variant, whatever its registers, runs two)
// PF: how many K tiles ahead of their split the global loads are issued (1 or 2: two register sets)
// SCHED 1: ask the scheduler to spread the split, the LDS stores and the loads between the MFMAs
template <bool A_KMAJ, bool B_KMAJ, bool TWO_ACC, int ABL = 0, int NB = 2, int PF = 1, int SCHED = 0>
__global__ void __launch_bounds__(256, NB)
coop_gemm(const float *__restrict__ A, const float *__restrict__ B, float *__restrict__ C, int M, int N, int K, long lda, long ldb, long long *clk) {
    __shared__ __attribute__((aligned(16))) char smem[(NB == 2 ? 3 : 2) * STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, l32 = lane & 31, half = lane >> 5;
    const int tiles_n = N / TN;
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, q = nwg >> 3, r = nwg & 7;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    const int per_group = 8 * tiles_n, gid = logical / per_group, first = gid * 8;
    const int rows_in = min(8, M / TM - first), in = logical - gid * per_group;
    const int tm = first + in % rows_in, tn = in / rows_in;
    const int m0 = ABL == 7 ? (tm & 1) * TM : tm * TM, n0 = ABL == 7 ? 0 : tn * TN;      // ABL 7: every block reads the same few tiles (cache-hot)
    const int nkt = K / GK;

    Loader<A_KMAJ> la;
    Loader<B_KMAJ> lb;
    la.init(A_KMAJ ? A + (long)m0 * lda : A + m0, lda, A_KMAJ ? ((long)(M - m0 - 1) * lda + K) * 4 : ((long)(K - 1) * lda + M - m0) * 4, tid);
    lb.init(B_KMAJ ? B + (long)n0 * ldb : B + n0, ldb, B_KMAJ ? ((long)(N - n0 - 1) * ldb + K) * 4 : ((long)(K - 1) * ldb + N - n0) * 4, tid);

    // fragment addresses: row = 64 w + 32 i + l32 -> (row >> 3) & 1 = (l32 >> 3) & 1
    const int fsw = (half ^ ((l32 >> 3) & 1)) << 4;
    const char *fa = smem + (wm * 64 + l32) * 32 + fsw;
    const char *fb = smem + 3 * PLANE + (wn * 64 + l32) * 32 + fsw;

    f32x16 acc[2][2] = {}, small[TWO_ACC ? 2 : 1][TWO_ACC ? 2 : 1] = {};
    float ga[8], gb[8], ga2[8], gb2[8];
    const long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    la.load(0, ga);
    lb.load(0, gb);
    la.store(smem, ga);
    lb.store(smem + 3 * PLANE, gb);
    if (nkt > 1) { la.load(1, ga); lb.load(1, gb); }
    if (PF == 2 && nkt > 2) { la.load(2, ga2); lb.load(2, gb2); }
    __syncthreads();
    auto iter = [&](int kt, float (&ga)[8], float (&gb)[8], auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;      // the last K tile: nothing left to split or load
        const int st = (kt & 1) * STAGE;
        u32x4 a[2][3], b[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                a[i][pl] = *reinterpret_cast<const u32x4 *>(fa + st + pl * PLANE + i * 1024);
                b[i][pl] = *reinterpret_cast<const u32x4 *>(fb + st + pl * PLANE + i * 1024);
            }
#define TERM(PA, PB, ACC)                                                           \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                   \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) mfma(a[i][PA], b[j][PB], ACC);
        if (ABL == 5) {      // the same MFMA work as 48 v_mfma_f32_16x16x32_bf16 (operands: whatever the reads returned)
            f32x4 *c16 = reinterpret_cast<f32x4 *>(&acc[0][0]), *s16 = reinterpret_cast<f32x4 *>(&small[0][0]);
#pragma unroll
            for (int term = 0; term < 6; ++term)
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    f32x4 &c = (TWO_ACC && term < 5) ? s16[(2 * t + (term & 1)) & 15] : c16[(2 * t + (term & 1)) & 15];
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[t & 1][term % 3]), __builtin_bit_cast(bf16x8, b[(t >> 1) & 1][term / 2]), c, 0, 0, 0);
                }
        } else
        if (TWO_ACC) {
            TERM(2, 0, small[i][j]) TERM(0, 2, small[i][j]) TERM(1, 1, small[i][j]) TERM(1, 0, small[i][j]) TERM(0, 1, small[i][j])
        } else {
            TERM(2, 0, acc[i][j]) TERM(0, 2, acc[i][j]) TERM(1, 1, acc[i][j]) TERM(1, 0, acc[i][j]) TERM(0, 1, acc[i][j])
        }
        TERM(0, 0, acc[i][j])
#undef TERM
        if (!LAST) {                                            // branch-free body: one scheduling region
            char *nx = smem + ((kt + 1) & 1) * STAGE;
            if (ABL < 1) {
                la.store(nx, ga);
                lb.store(nx + 3 * PLANE, gb);
            } else if (ABL == 4) {
                const Split3 sa = split3(ga), sb = split3(gb);
                unsigned x = 0;
#pragma unroll
                for (int t = 0; t < 4; ++t) x ^= sa.hi[t] ^ sa.mid[t] ^ sa.lo[t] ^ sb.hi[t] ^ sb.mid[t] ^ sb.lo[t];
                if (x == 0x12345u) nx[tid] = 1;
            } else if (ABL == 1 || ABL == 6 || ABL == 7) {
                float s = 0;
#pragma unroll
                for (int t = 0; t < 8; ++t) s += ga[t] + gb[t];
                if (s == 12345.f) nx[tid] = 1;
            }
            if (ABL < 2 || ABL == 4 || ABL == 7) { la.load(kt + 1 + PF, ga); lb.load(kt + 1 + PF, gb); }      // past the end: in-range garbage or the descriptor's zero, never stored
            if (ABL == 6) {
                if (A_KMAJ) la.load_lines(kt + 1 + PF, ga, tid); else la.load(kt + 1 + PF, ga);
                if (B_KMAJ) lb.load_lines(kt + 1 + PF, gb, tid); else lb.load(kt + 1 + PF, gb);
            }
        }
        if (SCHED == 1 && !LAST) {
            // 12 fragment reads first (4 before the first MFMA), then per MFMA four split instructions; the stores
            // and loads follow the split of their operand
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
            for (int g = 0; g < 24; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (g < 4) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                if (g >= 11 && g < 14) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                if (g >= 14 && g < 16) __builtin_amdgcn_sched_group_barrier(0x020, A_KMAJ ? 1 : 4, 0);
                if (g >= 21 && g < 24) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x020, 8, 0);
        }
        if (ABL != 3) __syncthreads();
    };
    if (PF == 1) {
        for (int kt = 0; kt + 1 < nkt; ++kt) iter(kt, ga, gb, std::false_type{});
        iter(nkt - 1, ga, gb, std::true_type{});
    } else {
        for (int kt = 0; kt + 2 < nkt; kt += 2) {       // nkt is even here
            iter(kt, ga, gb, std::false_type{});
            iter(kt + 1, ga2, gb2, std::false_type{});
        }
        iter(nkt - 2, ga, gb, std::false_type{});
        iter(nkt - 1, ga2, gb2, std::true_type{});
    }
    if (clk && tid == 0) {
        clk[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - c0;
        clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
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

template <bool A_KMAJ, bool B_KMAJ, bool TWO_ACC, int ABL = 0, int NB = 2, int PF = 1, int SCHED = 0>
int run(const float *A, const float *B, float *C, int M, int N, int K, int reps, float *ms_out, long long *clk = nullptr) {
    const int grid = (M / TM) * (N / TN);
    const long lda = A_KMAJ ? K : M, ldb = B_KMAJ ? K : N;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    coop_gemm<A_KMAJ, B_KMAJ, TWO_ACC, ABL, NB, PF, SCHED><<<grid, 256>>>(A, B, C, M, N, K, lda, ldb, nullptr);
    CK(hipDeviceSynchronize());
    if (reps) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) coop_gemm<A_KMAJ, B_KMAJ, TWO_ACC, ABL, NB, PF, SCHED><<<grid, 256>>>(A, B, C, M, N, K, lda, ldb, clk);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(ms_out, e0, e1));
        *ms_out /= reps;
    }
    return 0;
}

static float gauss() {
    const double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0);
    return (float)(std::sqrt(-2 * std::log(u)) * std::cos(6.283185307179586 * v));
}

template <bool A_KMAJ, bool B_KMAJ>
int check(const char *name) {
    const int M = 256, N = 384, K = 512;
    std::vector<float> a((size_t)M * K), b((size_t)N * K), c((size_t)M * N);
    for (auto &x : a) x = gauss();
    for (auto &x : b) x = gauss() / 16;
    float *da, *db, *dc;
    CK(hipMalloc(&da, a.size() * 4)); CK(hipMalloc(&db, b.size() * 4)); CK(hipMalloc(&dc, c.size() * 4));
    CK(hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice));
    for (int two = 0; two < 2; ++two) {
        float ms;
        if (two ? run<A_KMAJ, B_KMAJ, true>(da, db, dc, M, N, K, 0, &ms) : run<A_KMAJ, B_KMAJ, false>(da, db, dc, M, N, K, 0, &ms)) return 1;
        CK(hipMemcpy(c.data(), dc, c.size() * 4, hipMemcpyDeviceToHost));
        double se = 0, me = 0, mx = 0;
        for (int i = 0; i < M; ++i)
            for (int j = 0; j < N; ++j) {
                double ref = 0;
                for (int k = 0; k < K; ++k)
                    ref += (double)(A_KMAJ ? a[(size_t)i * K + k] : a[(size_t)k * M + i]) * (B_KMAJ ? b[(size_t)j * K + k] : b[(size_t)k * N + j]);
                const double e = c[(size_t)i * N + j] - ref;
                se += e * e; me += e; mx = std::fmax(mx, std::fabs(ref));
            }
        printf("check %s %s: rms err %.3e, mean err %+.3e, max|ref| %.3g\n", name, two ? "two accumulators" : "one accumulator ", std::sqrt(se / (M * N)), me / (M * N), mx);
    }
    CK(hipFree(da)); CK(hipFree(db)); CK(hipFree(dc));
    return 0;
}

long long *g_clk = nullptr;

template <bool A_KMAJ, bool B_KMAJ, bool TWO, int ABL, int NB = 2, int PF = 1, int SCHED = 0>
int ablate_one(const char *what, int M, int N, int K, const float *A, const float *B, float *C) {
    float ms = 0;
    if (run<A_KMAJ, B_KMAJ, TWO, ABL, NB, PF, SCHED>(A, B, C, M, N, K, 5, &ms, g_clk)) return 1;
    std::vector<long long> h(2 * (size_t)(M / TM) * (N / TN));
    CK(hipMemcpy(h.data(), g_clk, h.size() * 8, hipMemcpyDeviceToHost));
    double cyc = 0, real = 0;
    for (size_t i = 0; i < h.size(); i += 2) { cyc += h[i]; real += h[i + 1]; }
    printf("    %-28s %6.1f TF  clock %.2f GHz  %5.0f cycles per K tile and block\n", what, 2.0 * M * N * K * 1e-9 / ms, cyc / real * 0.1, cyc / (h.size() / 2) / (K / GK));
    return 0;
}

template <bool A_KMAJ, bool B_KMAJ>
int ablate(const char *name, int M, int N, int K, const float *A, const float *B, float *C) {
    printf("%s M=%d N=%d K=%d, two accumulators:\n", name, M, N, K);
    return ablate_one<A_KMAJ, B_KMAJ, true, 0>("full", M, N, K, A, B, C) ||
           ablate_one<A_KMAJ, B_KMAJ, true, 0, 2, 1, 1>("full, interleaved", M, N, K, A, B, C) ||
           ablate_one<A_KMAJ, B_KMAJ, false, 0, 2, 1, 1>("one acc full, interleaved", M, N, K, A, B, C) ||
           ablate_one<A_KMAJ, B_KMAJ, false, 0, 3, 1, 1>("one acc 3/CU, interleaved", M, N, K, A, B, C) ||
           ablate_one<A_KMAJ, B_KMAJ, true, 0, 2, 2>("full, loads 2 tiles ahead", M, N, K, A, B, C) ||
           ablate_one<A_KMAJ, B_KMAJ, true, 1, 2, 2>("no split/stores, 2 ahead", M, N, K, A, B, C) ||
           ablate_one<A_KMAJ, B_KMAJ, false, 0, 2, 2>("one acc full, 2 ahead", M, N, K, A, B, C) ||
           ablate_one<A_KMAJ, B_KMAJ, false, 0, 3, 2>("one acc 3 blocks/CU full, 2 ahead", M, N, K, A, B, C) ||
           ablate_one<A_KMAJ, B_KMAJ, true, 4>("no LDS stores", M, N, K, A, B, C) ||
           ablate_one<A_KMAJ, B_KMAJ, true, 1>("no split, no stores", M, N, K, A, B, C) ||
           ablate_one<A_KMAJ, B_KMAJ, true, 6>("same, whole-line loads", M, N, K, A, B, C) ||
           ablate_one<A_KMAJ, B_KMAJ, true, 7>("same, cache-hot panels", M, N, K, A, B, C) ||
           ablate_one<A_KMAJ, B_KMAJ, true, 2>("+ no global loads", M, N, K, A, B, C) ||
           ablate_one<A_KMAJ, B_KMAJ, true, 3>("+ no barrier", M, N, K, A, B, C) ||
           ablate_one<A_KMAJ, B_KMAJ, true, 5>("no loads, 16x16x32 MFMAs", M, N, K, A, B, C) ||
           ablate_one<A_KMAJ, B_KMAJ, false, 5>("one acc: same", M, N, K, A, B, C) ||
           ablate_one<A_KMAJ, B_KMAJ, false, 0>("one accumulator: full", M, N, K, A, B, C) ||
           ablate_one<A_KMAJ, B_KMAJ, false, 2>("one accumulator: no loads", M, N, K, A, B, C) ||
           ablate_one<A_KMAJ, B_KMAJ, false, 0, 3>("one acc, 3 blocks/CU: full", M, N, K, A, B, C) ||
           ablate_one<A_KMAJ, B_KMAJ, false, 2, 3>("one acc, 3 blocks/CU: no loads", M, N, K, A, B, C) ||
           ablate_one<A_KMAJ, B_KMAJ, false, 5, 3>("one acc, 3 blocks/CU: 16x16x32", M, N, K, A, B, C);
}

template <bool A_KMAJ, bool B_KMAJ>
int bench(const char *name, int M, int N, int K, const float *A, const float *B, float *C) {
    for (int two = 0; two < 2; ++two) {
        float ms = 0;
        if (two ? run<A_KMAJ, B_KMAJ, true>(A, B, C, M, N, K, 5, &ms) : run<A_KMAJ, B_KMAJ, false>(A, B, C, M, N, K, 5, &ms)) return 1;
        printf("%s %s M=%d N=%d K=%d: %.3f ms = %.1f TFLOP/s fp32-equivalent (%.1f %% of the bf16 pipe)\n", name, two ? "two acc" : "one acc", M, N, K, ms,
               2.0 * M * N * K / ms * 1e-9, 100.0 * 12.0 * M * N * K / ms * 1e-9 / 2516.6);
    }
    return 0;
}

int main() {
    srand(1);
    if (check<true, true>("NT") || check<true, false>("NN") || check<false, false>("TN")) return 1;
    // the encoder's FFN shapes: forward (NN), grad_x (NT), grad_w (TN, K = rows)
    const size_t big = (size_t)131072 * 4096;
    float *x, *w, *y;
    CK(hipMalloc(&x, big * 4)); CK(hipMalloc(&w, (size_t)4096 * 4096 * 4)); CK(hipMalloc(&y, big * 4));
    std::vector<float> h((size_t)1 << 22);
    for (auto &v : h) v = (float)((rand() % 2001 - 1000) * 1e-3);
    for (size_t o = 0; o < big; o += h.size()) { CK(hipMemcpy(x + o, h.data(), h.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(y + o, h.data(), h.size() * 4, hipMemcpyHostToDevice)); }
    for (size_t o = 0; o < (size_t)4096 * 4096; o += h.size()) CK(hipMemcpy(w + o, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&g_clk, 2 * 32768 * 8));
    if (ablate<true, true>("NT ffn2 dx   ", 131072, 4096, 1024, x, w, y)) return 1;
    if (ablate<true, false>("NN ffn1 fwd  ", 131072, 4096, 1024, x, w, y)) return 1;
    if (bench<true, false>("NN ffn1 fwd  ", 131072, 4096, 1024, x, w, y)) return 1;
    if (bench<true, false>("NN ffn2 fwd  ", 131072, 1024, 4096, x, w, y)) return 1;
    if (bench<true, true>("NT ffn1 dx   ", 131072, 1024, 4096, x, w, y)) return 1;
    if (bench<true, true>("NT ffn2 dx   ", 131072, 4096, 1024, x, w, y)) return 1;
    if (bench<false, false>("TN dw proxy  ", 4096, 4096, 8192, x, y, w)) return 1;
    if (bench<true, false>("NN proj      ", 131072, 1024, 1024, x, w, y)) return 1;
    return 0;
}
