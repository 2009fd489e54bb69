// Prototype of the NEXT split: fp32 GEMM on the f16 matrix pipe with a TWO-way split and row scaling -- three
// v_mfma_f32_32x32x16_f16 per product instead of the six bf16 MFMAs of the three-way bf16 split.
//   x s = hi + lo   (hi = fp16(x s) rounded to nearest: 11 bits; lo = fp16(x s - hi): the next 11), s = a power of two per
//   ROW of A (and per row of B) that puts the row's largest magnitude in [2^13, 2^14): every element within 2^-17 of it
//   keeps all 22 bits, smaller ones lose at most 2^-39 of the row maximum (lo goes subnormal);
//   a b ~ (hi hi + hi lo + lo hi) / (s_a s_b): the dropped lo lo is below 2^-22 |a b|.
// The error contract changes from elementwise (bf16 x 3: every element carries 24 bits) to row-normwise -- which is
// what a dot product's error is anyway.  The scales need one pass over each operand (row maxima), timed here too.
// Layout NT only (A [M][K], B [N][K]); the split happens once per block on the way to LDS as in coop_split_gemm.hip.
//   hipcc -O3 --offload-arch=gfx950 f16x2_gemm.hip -o f16x2_gemm && ./f16x2_gemm
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int GK = 16, TM = 128, TN = 128;
constexpr int PLANE = TM * GK * 2;                  // bytes per plane tile (4 KB): [128 rows][32 B]
constexpr int STAGE = 4 * PLANE;                    // A hi/lo + B hi/lo (16 KB)

// one wave per row: s = 2^(13 - floor(log2(max |x|))), 1 for an all-zero row
__global__ void row_scales(const float *__restrict__ x, float *__restrict__ scale, float *__restrict__ inv, int rows, int k) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    float m = 0.f;
    for (int c = lane * 4; c < k; c += 256) {
        const float4 v = *reinterpret_cast<const float4 *>(x + (size_t)row * k + c);
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if (lane == 0) {
        int e = 0;
        if (m > 0.f) (void)frexpf(m, &e);            // m = f 2^e, f in [0.5, 1)  ->  m 2^(14 - e) in [2^13, 2^14)
        const float s = m > 0.f ? ldexpf(1.f, 14 - e) : 1.f;
        scale[row] = s;
        inv[row] = 1.f / s;
    }
}

struct Split2 { u32x4 hi, lo; };

__device__ __forceinline__ Split2 split2(const float (&v)[8], float s) {
    Split2 out;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const float x0 = v[2 * t] * s, x1 = v[2 * t + 1] * s;
        const f16x2 h = __builtin_convertvector(f32x2{x0, x1}, f16x2);           // v_cvt_pk_f16_f32, round to nearest
        const f16x2 l = __builtin_convertvector(f32x2{x0 - (float)h.x, x1 - (float)h.y}, f16x2);
        out.hi[t] = __builtin_bit_cast(unsigned, h);
        out.lo[t] = __builtin_bit_cast(unsigned, l);
    }
    return out;
}

__device__ __forceinline__ void mfma(const u32x4 &a, const u32x4 &b, f32x16 &c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// one thread's share of a K-major operand tile (128 rows x 16 k): the 8 consecutive k of one row
struct Loader {
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned voff, lds_off;
    float s;
    __device__ __forceinline__ void init(const float *panel, long ld, long bytes_left, const float *scale, int tid) {
        rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)panel, 0, (int)(bytes_left > 0xFFFFFFFFL ? 0xFFFFFFFFL : bytes_left), 0x00020000);
        const int row = tid >> 1, h = tid & 1;
        voff = (unsigned)(row * ld * 4 + h * 32);
        lds_off = row * 32 + ((h ^ ((row >> 3) & 1)) << 4);
        s = scale[row];
    }
    __device__ __forceinline__ void load(int kt, float (&v)[8]) const {
        const unsigned so = kt * GK * 4u;
        const u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, so, 0);
        const u32x4 y = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + 16, so, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) { v[t] = __uint_as_float(x[t]); v[4 + t] = __uint_as_float(y[t]); }
    }
    __device__ __forceinline__ void store(char *planes, const float (&v)[8]) const {
        const Split2 sp = split2(v, s);
        *reinterpret_cast<u32x4 *>(planes + lds_off) = sp.hi;
        *reinterpret_cast<u32x4 *>(planes + PLANE + lds_off) = sp.lo;
    }
};

// C[M, N] = A[M, K] B[N, K]^T.  TWO_ACC: the two cross terms accumulate apart from hi hi (as the bf16 split's small terms do).
template <bool TWO_ACC, int NB>
__global__ void __launch_bounds__(256, NB)
f16x2_gemm(const float *__restrict__ A, const float *__restrict__ B, float *__restrict__ C, const float *__restrict__ sa,
           const float *__restrict__ sb, const float *__restrict__ inv_sa, const float *__restrict__ inv_sb, int M, int N, int K) {
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
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

    Loader la, lb;
    la.init(A + (long)m0 * K, K, ((long)(M - m0 - 1) * K + K) * 4, sa + m0, tid);
    lb.init(B + (long)n0 * K, K, ((long)(N - n0 - 1) * K + K) * 4, sb + n0, tid);
    const int fsw = (half ^ ((l32 >> 3) & 1)) << 4;
    const char *fa = smem + (wm * 64 + l32) * 32 + fsw;
    const char *fb = smem + 2 * PLANE + (wn * 64 + l32) * 32 + fsw;

    f32x16 acc[2][2] = {}, small[TWO_ACC ? 2 : 1][TWO_ACC ? 2 : 1] = {};
    float ga[8], gb[8];
    la.load(0, ga);
    lb.load(0, gb);
    la.store(smem, ga);
    lb.store(smem + 2 * PLANE, gb);
    la.load(1, ga);
    lb.load(1, gb);
    __syncthreads();
    auto iter = [&](int kt, bool last) {
        const int st = (kt & 1) * STAGE;
        u32x4 a[2][2], b[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                a[i][pl] = *reinterpret_cast<const u32x4 *>(fa + st + pl * PLANE + i * 1024);
                b[i][pl] = *reinterpret_cast<const u32x4 *>(fb + st + pl * PLANE + i * 1024);
            }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x16 &sm = TWO_ACC ? small[i][j] : acc[i][j];
                mfma(a[i][1], b[j][0], sm);
                mfma(a[i][0], b[j][1], sm);
            }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) mfma(a[i][0], b[j][0], acc[i][j]);
        if (!last) {
            char *nx = smem + ((kt + 1) & 1) * STAGE;
            la.store(nx, ga);
            lb.store(nx + 2 * PLANE, gb);
            la.load(kt + 2, ga);                      // past the end: the descriptor's zero or in-range garbage, never stored
            lb.load(kt + 2, gb);
        }
        __syncthreads();
    };
    for (int kt = 0; kt + 1 < nkt; ++kt) iter(kt, false);
    iter(nkt - 1, true);

    float *cbase = C + (size_t)(m0 + wm * 64) * N + n0 + wn * 64;
    const auto rc = __builtin_amdgcn_make_buffer_rsrc((void *)cbase, 0, (int)(64 * (size_t)N * 4), 0x00020000);
    const int vo = (4 * half * N + l32) * 4;
    const float cb[2] = {inv_sb[n0 + wn * 64 + l32], inv_sb[n0 + wn * 64 + 32 + l32]};
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const int row = i * 32 + (rr & 3) + 8 * (rr >> 2);
            const float ra = inv_sa[m0 + wm * 64 + row + 4 * half];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float v = acc[i][j][rr];
                if (TWO_ACC) v += small[i][j][rr];
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v * ra * cb[j]), rc, vo + j * 128, row * N * 4, 0);
            }
        }
}

static float gauss() {
    const double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0);
    return (float)(std::sqrt(-2 * std::log(u)) * std::cos(6.283185307179586 * v));
}

template <bool TWO, int NB>
int launch(const float *A, const float *B, float *C, float *sa, float *sb, float *ia, float *ib, int M, int N, int K) {
    f16x2_gemm<TWO, NB><<<(M / TM) * (N / TN), 256>>>(A, B, C, sa, sb, ia, ib, M, N, K);
    return 0;
}

int check(const char *what, float row_spread, float elem_spread) {
    // row_spread: rows of A differ in magnitude by up to 2^row_spread; elem_spread: elements inside a row by 2^elem_spread
    const int M = 256, N = 256, K = 4096;
    std::vector<float> a((size_t)M * K), b((size_t)N * K), c((size_t)M * N);
    for (int i = 0; i < M; ++i) {
        const float rs = std::ldexp(1.f, (int)(row_spread * ((i * 37) % M) / M));
        for (int k = 0; k < K; ++k) a[(size_t)i * K + k] = gauss() * rs * std::ldexp(1.f, -(int)(elem_spread * ((k * 13) % 64) / 64));
    }
    for (auto &x : b) x = gauss() / 64;
    float *da, *db, *dc, *sa, *sb, *ia, *ib;
    CK(hipMalloc(&da, a.size() * 4)); CK(hipMalloc(&db, b.size() * 4)); CK(hipMalloc(&dc, c.size() * 4));
    CK(hipMalloc(&sa, M * 4)); CK(hipMalloc(&sb, N * 4)); CK(hipMalloc(&ia, M * 4)); CK(hipMalloc(&ib, N * 4));
    CK(hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice));
    row_scales<<<M / 4, 256>>>(da, sa, ia, M, K);
    row_scales<<<N / 4, 256>>>(db, sb, ib, N, K);
    for (int two = 0; two < 2; ++two) {
        if (two) launch<true, 2>(da, db, dc, sa, sb, ia, ib, M, N, K); else launch<false, 2>(da, db, dc, sa, sb, ia, ib, M, N, K);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(c.data(), dc, c.size() * 4, hipMemcpyDeviceToHost));
        // error of every element relative to the size of its own ROW of C (row-normwise), against fp64; and the same
        // for a plain fp32 dot product in k order (what the exact-f32 MFMA computes)
        double worst = 0, worst32 = 0, rms = 0, rms32 = 0;
        for (int i = 0; i < M; ++i) {
            std::vector<double> ref(N);
            std::vector<float> f32(N);
            double rowmax = 0;
            for (int j = 0; j < N; ++j) {
                double s = 0; float f = 0;
                for (int k = 0; k < K; ++k) { s += (double)a[(size_t)i * K + k] * b[(size_t)j * K + k]; f = fmaf(a[(size_t)i * K + k], b[(size_t)j * K + k], f); }
                ref[j] = s; f32[j] = f; rowmax = std::fmax(rowmax, std::fabs(s));
            }
            for (int j = 0; j < N; ++j) {
                const double e = (c[(size_t)i * N + j] - ref[j]) / rowmax, e32 = (f32[j] - ref[j]) / rowmax;
                worst = std::fmax(worst, std::fabs(e)); worst32 = std::fmax(worst32, std::fabs(e32)); rms += e * e; rms32 += e32 * e32;
            }
        }
        printf("check %-28s %s: error / row max of C: rms %.2e worst %.2e   (fp32 fma chain: rms %.2e worst %.2e)\n", what,
               two ? "two accumulators" : "one accumulator ", std::sqrt(rms / (M * N)), worst, std::sqrt(rms32 / (M * N)), worst32);
    }
    CK(hipFree(da)); CK(hipFree(db)); CK(hipFree(dc)); CK(hipFree(sa)); CK(hipFree(sb)); CK(hipFree(ia)); CK(hipFree(ib));
    return 0;
}

template <bool TWO, int NB>
int bench(const char *name, int M, int N, int K, const float *A, const float *B, float *C, float *sa, float *sb, float *ia, float *ib) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch<TWO, NB>(A, B, C, sa, sb, ia, ib, M, N, K);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < 10; ++i) launch<TWO, NB>(A, B, C, sa, sb, ia, ib, M, N, K);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 10;
    CK(hipEventRecord(e0));
    for (int i = 0; i < 10; ++i) row_scales<<<M / 4, 256>>>(A, sa, ia, M, K);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float sms = 0;
    CK(hipEventElapsedTime(&sms, e0, e1));
    sms /= 10;
    printf("%s %s, %d blocks/CU  M=%d N=%d K=%d: %.3f ms = %.1f TFLOP/s fp32-equivalent (%.1f %% of the f16 pipe executed); row-maximum pass over A %.3f ms (%.0f GB/s)\n",
           name, TWO ? "two acc" : "one acc", NB, M, N, K, ms, 2.0 * M * N * K / ms * 1e-9, 100.0 * 6.0 * M * N * K / ms * 1e-9 / 2516.6, sms, 4.0 * M * K / sms * 1e-6);
    return 0;
}

int main() {
    srand(1);
    if (check("N(0,1) rows", 0, 0) || check("rows spread over 2^40", 40, 0) || check("elements spread over 2^12", 0, 12) ||
        check("elements spread over 2^30", 0, 30)) return 1;
    const size_t big = (size_t)131072 * 4096;
    float *x, *w, *y, *sa, *sb, *ia, *ib;
    CK(hipMalloc(&x, big * 4)); CK(hipMalloc(&w, (size_t)4096 * 4096 * 4)); CK(hipMalloc(&y, big * 4));
    CK(hipMalloc(&sa, 131072 * 4)); CK(hipMalloc(&sb, 4096 * 4)); CK(hipMalloc(&ia, 131072 * 4)); CK(hipMalloc(&ib, 4096 * 4));
    std::vector<float> h((size_t)1 << 22);
    for (auto &v : h) v = (float)((rand() % 2001 - 1000) * 1e-3);
    for (size_t o = 0; o < big; o += h.size()) CK(hipMemcpy(x + o, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    for (size_t o = 0; o < (size_t)4096 * 4096; o += h.size()) CK(hipMemcpy(w + o, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    const int shapes[3][3] = {{131072, 4096, 1024}, {131072, 1024, 4096}, {131072, 1024, 1024}};
    for (auto &s : shapes) {
        const int M = s[0], N = s[1], K = s[2];
        row_scales<<<M / 4, 256>>>(x, sa, ia, M, K);
        row_scales<<<N / 4, 256>>>(w, sb, ib, N, K);
        CK(hipDeviceSynchronize());
        if (bench<false, 2>("NT", M, N, K, x, w, y, sa, sb, ia, ib) || bench<true, 2>("NT", M, N, K, x, w, y, sa, sb, ia, ib) ||
            bench<false, 3>("NT", M, N, K, x, w, y, sa, sb, ia, ib) || bench<true, 3>("NT", M, N, K, x, w, y, sa, sb, ia, ib) ||
            bench<false, 4>("NT", M, N, K, x, w, y, sa, sb, ia, ib)) return 1;
    }
    return 0;
}
