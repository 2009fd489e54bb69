// Microbenchmark: issue rate of v_mfma_f32_16x16x4_f32 (8 passes: 32 cycles by the book, the same 64 FLOP per cycle and
// SIMD as v_mfma_f32_32x32x2_f32) with W waves per SIMD, NACC accumulators used round-robin and F vector-ALU fillers after
// every MFMA.  Prints cycles per MFMA (s_memtime, median over waves) per wave and per SIMD; the floor per SIMD is 32.
//   hipcc -O3 --offload-arch=gfx950 mfma_f32_16x16.hip -o mfma_f32_16x16 && ./mfma_f32_16x16
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MFMA16(ACC) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(ACC) : "v"(a), "v"(b))
#define MFMA32(ACC) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(ACC) : "v"(a), "v"(b))
#define FILL(I)                                                                                   \
    do {                                                                                          \
        if ((I) % 2 == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x0) : "v"(a));               \
        else asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x1) : "v"(b));                         \
    } while (0)

// other kinds of filler: KIND 1 scalar ALU, 2 LDS read (waited for once per 16 MFMAs), 3 s_nop, 4 s_waitcnt with nothing outstanding
template <int KIND, int F, bool BIG, int THREADS>
__global__ void __launch_bounds__(THREADS) kern_kind(long long *out, int trips) {
    __shared__ float lds[1024];
    float a = threadIdx.x * 0.001f, b = 1.0f - a, y = 0.f;
    lds[threadIdx.x & 1023] = a;
    __syncthreads();
    unsigned sc = 1;
    f32x4 acc[8] = {};
    f32x16 big[2] = {};
    const unsigned la = (threadIdx.x & 63) * 4;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < trips; ++t) {
        float v = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            if (BIG) MFMA32(big[g % 2]);
            else MFMA16(acc[g % 8]);
#pragma unroll
            for (int i = 0; i < F; ++i) {
                if (KIND == 1) asm volatile("s_add_u32 %0, %0, 3" : "+s"(sc) : : "scc");
                else if (KIND == 2) asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(la));
                else if (KIND == 3) asm volatile("s_nop 0");
                else asm volatile("s_waitcnt lgkmcnt(0)");
            }
        }
        if (KIND == 2) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); y += v; }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6)] = t1 - t0;
    float s = y + (float)sc;
    for (int i = 0; i < 8; ++i) s += acc[i][0];
    if (s + big[0][0] + big[1][3] == 1234.5f) out[0] = 0;
}

template <int NACC, int F, bool BIG, int THREADS>
__global__ void __launch_bounds__(THREADS) kern(long long *out, int trips) {
    float a = threadIdx.x * 0.001f, b = 1.0f - a, x0 = a, x1 = b;
    f32x4 acc[8] = {};
    f32x16 big[2] = {};
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            if (BIG) MFMA32(big[g % (NACC > 2 ? 2 : NACC)]);
            else MFMA16(acc[g % NACC]);
#pragma unroll
            for (int i = 0; i < F; ++i) FILL(i + g);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6)] = t1 - t0;
    float s = x0 + x1;
    for (int i = 0; i < 8; ++i) s += acc[i][0];
    if (s + big[0][0] + big[1][3] == 1234.5f) out[0] = 0;
}

template <int NACC, int F, bool BIG, int WAVES_PER_SIMD>
int run(long long *dev) {
    constexpr int THREADS = 256 * (WAVES_PER_SIMD > 2 ? 2 : WAVES_PER_SIMD);           // one block of 4 or 8 waves ...
    const int per_cu = WAVES_PER_SIMD > 2 ? WAVES_PER_SIMD / 2 : 1;                     // ... times blocks per CU
    const int trips = 500, blocks = 256 * per_cu;
    hipLaunchKernelGGL((kern<NACC, F, BIG, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, dev, trips);
    CK(hipDeviceSynchronize());
    std::vector<long long> h(blocks * (THREADS / 64));
    CK(hipMemcpy(h.data(), dev, h.size() * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    const double per_wave = (double)h[h.size() / 2] / (trips * 16.0);
    printf("%s  waves/SIMD %d  accumulators %d  fillers/MFMA %2d : %6.1f cycles per MFMA per wave = %6.1f per SIMD (floor %d)\n",
           BIG ? "32x32x2 " : "16x16x4", WAVES_PER_SIMD, NACC, F, per_wave, per_wave / WAVES_PER_SIMD, BIG ? 64 : 32);
    return 0;
}

template <int KIND, int F, bool BIG, int WAVES_PER_SIMD>
int run_kind(long long *dev) {
    constexpr int THREADS = 256 * WAVES_PER_SIMD;
    const int trips = 500, blocks = 256;
    hipLaunchKernelGGL((kern_kind<KIND, F, BIG, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, dev, trips);
    CK(hipDeviceSynchronize());
    std::vector<long long> h(blocks * (THREADS / 64));
    CK(hipMemcpy(h.data(), dev, h.size() * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    const double per_wave = (double)h[h.size() / 2] / (trips * 16.0);
    const char *names[] = {"", "scalar ALU", "LDS read", "s_nop", "s_waitcnt"};
    printf("%s  waves/SIMD %d  %-10s fillers/MFMA %2d : %6.1f cycles per MFMA per wave = %6.1f per SIMD (floor %d)\n",
           BIG ? "32x32x2 " : "16x16x4", WAVES_PER_SIMD, names[KIND], F, per_wave, per_wave / WAVES_PER_SIMD, BIG ? 64 : 32);
    return 0;
}

int main() {
    long long *dev;
    CK(hipMalloc(&dev, 256 * 4 * 8 * 8 * 2));
    run<1, 0, false, 1>(dev); run<2, 0, false, 1>(dev); run<4, 0, false, 1>(dev); run<8, 0, false, 1>(dev);
    run<8, 1, false, 1>(dev); run<8, 2, false, 1>(dev); run<8, 4, false, 1>(dev);
    run<1, 0, false, 2>(dev); run<2, 0, false, 2>(dev); run<8, 0, false, 2>(dev);
    run<8, 1, false, 2>(dev); run<8, 2, false, 2>(dev); run<8, 4, false, 2>(dev); run<2, 1, false, 2>(dev);
    run<8, 0, false, 4>(dev); run<8, 1, false, 4>(dev);
    run<1, 0, true, 1>(dev); run<2, 0, true, 1>(dev); run<2, 1, true, 1>(dev); run<2, 2, true, 1>(dev);
    run<2, 0, true, 2>(dev); run<2, 1, true, 2>(dev); run<2, 2, true, 2>(dev); run<2, 4, true, 2>(dev);
    run_kind<1, 1, false, 2>(dev); run_kind<1, 2, false, 2>(dev); run_kind<1, 4, false, 2>(dev);
    run_kind<2, 1, false, 2>(dev); run_kind<2, 2, false, 2>(dev);
    run_kind<3, 1, false, 2>(dev); run_kind<3, 2, false, 2>(dev);
    run_kind<4, 1, false, 2>(dev); run_kind<4, 2, false, 2>(dev);
    run_kind<1, 1, true, 2>(dev); run_kind<1, 2, true, 2>(dev); run_kind<1, 4, true, 2>(dev);
    run_kind<2, 1, true, 2>(dev); run_kind<2, 2, true, 2>(dev);
    run_kind<3, 1, true, 2>(dev); run_kind<4, 1, true, 2>(dev);
    run_kind<1, 1, true, 1>(dev); run_kind<2, 1, true, 1>(dev); run_kind<3, 1, true, 1>(dev); run_kind<4, 1, true, 1>(dev);
    return 0;
}
