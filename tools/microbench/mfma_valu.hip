// Microbenchmark: cost of vector-ALU fillers beside v_mfma_f32_32x32x16_bf16 on one SIMD.
//   interleaved: [1 MFMA, F fillers] x 24 per loop trip        (fillers inside the wave's own MFMA gaps)
//   phased:      [24 F fillers] then [24 MFMA] per loop trip     (what a compiler emits for split-then-multiply)
// with 1..4 waves per SIMD (1..4 blocks of 4 waves per CU).  Prints ns per MFMA per SIMD; the MFMA-only floor is
// 32 cycles.   hipcc -O3 --offload-arch=gfx950 mfma_valu.hip -o mfma_valu && ./mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define MFMA(ACC) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(a), "v"(b))
// one filler = the op mix of the operand split (and / sub / perm / cvt), on registers no MFMA touches
#define FILL(I)                                                                                              \
    do {                                                                                                     \
        if ((I) % 4 == 0) asm volatile("v_and_b32 %0, %1, %0" : "+v"(x0) : "v"(msk));                        \
        else if ((I) % 4 == 1) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x1) : "v"(x0));                    \
        else if ((I) % 4 == 2) asm volatile("v_perm_b32 %0, %1, %0, %2" : "+v"(x2) : "v"(x1), "v"(sel));     \
        else asm volatile("v_and_b32 %0, %1, %0" : "+v"(x3) : "v"(msk));                                     \
    } while (0)

template <int F, bool PHASED>
__global__ void __launch_bounds__(256) kern(float *out, int trips) {
    u32x4 a = {threadIdx.x, 1u, 2u, 3u}, b = {4u, 5u, 6u, threadIdx.x};
    f32x16 acc0 = {}, acc1 = {}, acc2 = {}, acc3 = {};
    unsigned x0 = threadIdx.x, x2 = 7u, x3 = 9u, msk = 0xffff0000u, sel = 0x07060302u;
    float x1 = 1.5f;
    for (int t = 0; t < trips; ++t) {
        if (PHASED) {
#pragma unroll
            for (int i = 0; i < 24 * F; ++i) FILL(i);
#pragma unroll
            for (int g = 0; g < 6; ++g) { MFMA(acc0); MFMA(acc1); MFMA(acc2); MFMA(acc3); }
        } else {
#pragma unroll
            for (int g = 0; g < 6; ++g) {
                MFMA(acc0);
#pragma unroll
                for (int i = 0; i < F; ++i) FILL(i);
                MFMA(acc1);
#pragma unroll
                for (int i = 0; i < F; ++i) FILL(i + 1);
                MFMA(acc2);
#pragma unroll
                for (int i = 0; i < F; ++i) FILL(i + 2);
                MFMA(acc3);
#pragma unroll
                for (int i = 0; i < F; ++i) FILL(i + 3);
            }
        }
    }
    if (acc0[0] + acc1[1] + acc2[2] + acc3[3] + x1 + (float)(x0 + x2 + x3) == 1234.5f) out[threadIdx.x] = 1.f;
}

template <int F, bool PHASED>
int run(int blocks_per_cu, float *out) {
    const int trips = 2000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    kern<F, PHASED><<<256 * blocks_per_cu, 256>>>(out, 200);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    kern<F, PHASED><<<256 * blocks_per_cu, 256>>>(out, trips);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double mfma_per_simd = (double)trips * 24 * blocks_per_cu;        // one wave of every block per SIMD
    const double ns = ms * 1e6 / mfma_per_simd;
    printf("%-11s fillers/MFMA=%d waves/SIMD=%d : %6.2f ns per MFMA per SIMD (%5.1f cycles at 2.4 GHz), %6.1f TF bf16\n",
           PHASED ? "phased" : "interleaved", F, blocks_per_cu, ns, ns * 2.4, 32768.0 * 1024 / ns * 1e-3);
    return 0;
}

int main() {
    float *out;
    CK(hipMalloc(&out, 4096));
    for (int w = 1; w <= 4; ++w) {
        if (run<0, false>(w, out)) return 1;
        if (run<3, false>(w, out)) return 1;
        if (run<5, false>(w, out)) return 1;
        if (run<7, false>(w, out)) return 1;
        if (run<9, false>(w, out)) return 1;
        if (run<7, true>(w, out)) return 1;
    }
    return 0;
}
