// Microbenchmark: what one wave per SIMD can hide beside v_mfma_f32_32x32x2_f32 (64 cycles per instruction).
//   NACC accumulators used round-robin (1 = a dependent chain, as in an attention tile), F vector-ALU fillers or one
//   LDS read after every MFMA.  Prints cycles per MFMA (s_memtime, median over waves); the floor is 64.
//   hipcc -O3 --offload-arch=gfx950 mfma_f32_chain.hip -o mfma_f32_chain && ./mfma_f32_chain
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MFMA(ACC) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(ACC) : "v"(a), "v"(b))
#define FILL(I)                                                                                   \
    do {                                                                                          \
        if ((I) % 2 == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x0) : "v"(a));               \
        else asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x1) : "v"(b));                         \
    } while (0)

template <int NACC, int F, int LDSR>
__global__ void __launch_bounds__(256) kern(long long *out, int trips) {
    __shared__ float lds[4096];
    float a = threadIdx.x * 0.001f, b = 1.0f - a, x0 = a, x1 = b, y = 0.f;
    lds[threadIdx.x] = a;
    __syncthreads();
    f32x16 acc[4] = {};
    const float *lp = lds + (threadIdx.x & 63);
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            MFMA(acc[g % NACC]);
            if (LDSR) {
                float v;
                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"((unsigned)(size_t)lp * 0 + (threadIdx.x & 63) * 4), "n"(256 * (g % 8)));
                if (g % 4 == 3) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); y += v; }
            }
#pragma unroll
            for (int i = 0; i < F; ++i) FILL(i + g);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    if (acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + x0 + x1 + y == 1234.5f) out[0] = 0;
}

template <int NACC, int F, int LDSR>
int run(long long *dev, int per_cu = 1) {
    const int trips = 500, blocks = 256 * per_cu;
    hipLaunchKernelGGL((kern<NACC, F, LDSR>), dim3(blocks), dim3(256), 0, 0, dev, trips);
    CK(hipDeviceSynchronize());
    std::vector<long long> h(blocks * 4);
    CK(hipMemcpy(h.data(), dev, h.size() * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    printf("waves/SIMD %d  accumulators %d  fillers/MFMA %2d  lds reads/MFMA %d : %6.1f cycles per MFMA per wave = %6.1f per SIMD\n", per_cu, NACC, F, LDSR,
           (double)h[h.size() / 2] / (trips * 16.0), (double)h[h.size() / 2] / (trips * 16.0) / per_cu);
    return 0;
}

int main() {
    long long *dev;
    CK(hipMalloc(&dev, 256 * 4 * 4 * 8));
    run<1, 0, 0>(dev); run<1, 2, 0>(dev); run<1, 4, 0>(dev); run<1, 8, 0>(dev); run<1, 12, 0>(dev); run<1, 16, 0>(dev);
    run<2, 0, 0>(dev); run<2, 4, 0>(dev); run<2, 8, 0>(dev); run<2, 12, 0>(dev);
    run<4, 0, 0>(dev); run<4, 4, 0>(dev); run<4, 8, 0>(dev); run<4, 12, 0>(dev);
    run<1, 0, 1>(dev); run<1, 4, 1>(dev); run<4, 4, 1>(dev);
    for (int w = 2; w <= 4; ++w) { run<1, 0, 0>(dev, w); run<1, 4, 0>(dev, w); run<1, 8, 0>(dev, w); run<1, 16, 0>(dev, w); run<1, 4, 1>(dev, w); }
    return 0;
}
