// Microbenchmark: how long does a wave take to ISSUE n dword stores of the MFMA-epilogue shape
// (two 128-B row segments per instruction), with 1..4 blocks of 4 waves per CU, alone or beside
// waves that stream LDS-DMA loads?  Prints per-wave issue time (s_memrealtime, 100 MHz) medians.
//   hipcc -O3 --offload-arch=gfx950 store_issue.hip -o store_issue && ./store_issue
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef __attribute__((address_space(3))) void lds_void;

template <int NST, bool WIDE>
__global__ void __launch_bounds__(256, 4) store_kernel(float *out, long ld, long long *stamps, int reps, const float *src, int loaders, int mode) {
    __shared__ __attribute__((aligned(16))) float smem[8192];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l32 = lane & 31, half = lane >> 5;
    if ((int)(blockIdx.x & 3) < loaders && mode != 0) {
        // neighbour block shaped like the GEMM main loop: per K tile 32 MFMAs per wave, 4 DMA pieces, 12 ds_read_b128
        typedef float f32x16 __attribute__((ext_vector_type(16)));
        const auto rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, 0x7FFFFFFF, 0x00020000);
        f32x16 acc[4] = {};
        float a = tid, b = tid * 0.25f;
        unsigned off0 = (blockIdx.x * 4 + wave) * 65536u + lane * 16;
        for (int r = 0; r < reps * 24; ++r) {
            const unsigned off = off0 + (r & 7) * 4096;
            if (mode & 2) {
                __syncthreads();
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void *)(smem + wave * 1024 + i * 256), 16, off + i * 1024, 0, 0, 0);
            }
            if (mode & 4) {
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    const float4 t = *reinterpret_cast<const float4 *>(smem + ((wave * 1024 + i * 64 + lane * 4) & 8191));
                    a += t.x; b += t.w;
                }
            }
            if (mode & 1) {
#pragma unroll
                for (int i = 0; i < 32; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i & 3], 0, 0, 0);
            } else {
                for (int k = 0; k < 16; ++k) asm volatile("s_sleep 8");
            }
        }
        if (acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] == 1234.5f) out[tid] = a + b;
        return;
    }
    if ((int)(blockIdx.x & 3) < loaders) {
        // loader block: stream LDS-DMA loads for the kernel's lifetime (about reps * 8 KB per wave)
        const auto rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, 0x7FFFFFFF, 0x00020000);
        unsigned off = (blockIdx.x * 4 + wave) * 65536u + lane * 16;
        for (int r = 0; r < reps * 64; ++r) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void *)(smem + wave * 2048), 16, off, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void *)(smem + wave * 2048 + 256), 16, off + 1024, 0, 0, 0);
            off += 2048;
            if ((r & 7) == 7) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); off = (blockIdx.x * 4 + wave) * 65536u + lane * 16; }
        }
        return;
    }
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = tid * 0.5f + i;
    long long t_issue = 0, t_done = 0;
    for (int rep = 0; rep < reps; ++rep) {
        float *base = out + ((long)(blockIdx.x * reps + rep) * 128 + (wave >> 1) * 64) * ld + (wave & 1) * 64;
        const auto rc = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)(64 * ld * 4), 0x00020000);
        const long long t0 = __builtin_amdgcn_s_memrealtime();
        if (WIDE) {
            const int voff = ((lane >> 4) * (int)ld + (lane & 15) * 4) * 4;
#pragma unroll
            for (int q = 0; q < NST / 4; ++q) {
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 d = {__float_as_uint(v[q & 15]), __float_as_uint(v[(q + 1) & 15]), __float_as_uint(v[(q + 2) & 15]), __float_as_uint(v[(q + 3) & 15])};
                __builtin_amdgcn_raw_buffer_store_b128(d, rc, voff, q * 4 * (int)ld * 4, 0);
            }
        } else {
            const int voff = (4 * half * (int)ld + l32) * 4;
#pragma unroll
            for (int s = 0; s < NST; ++s) {
                const int r = s >> 1, j = s & 1;
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[s & 15]), rc, voff + j * 128,
                                                      (((r & 3) + 8 * ((r >> 2) & 3)) + 32 * (r >> 4)) * (int)ld * 4, 0);
            }
        }
        const long long t1 = __builtin_amdgcn_s_memrealtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const long long t2 = __builtin_amdgcn_s_memrealtime();
        t_issue += t1 - t0;
        t_done += t2 - t0;
        // pretend to compute for a while so stores of different waves do not stay in lockstep
        for (int k = 0; k < 64; ++k) asm volatile("s_sleep 8");
    }
    if (lane == 0) {
        stamps[(blockIdx.x * 4 + wave) * 2 + 0] = t_issue;
        stamps[(blockIdx.x * 4 + wave) * 2 + 1] = t_done;
    }
}

template <int NST, bool WIDE>
int run(int blocks_per_cu, int loaders, float *out, long ld, long long *stamps, const float *src, int mode = 0) {
    const int cus = 256, reps = 8;
    const int grid = cus * blocks_per_cu;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    store_kernel<NST, WIDE><<<grid, 256>>>(out, ld, stamps, reps, src, loaders, mode);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    store_kernel<NST, WIDE><<<grid, 256>>>(out, ld, stamps, reps, src, loaders, mode);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> h(grid * 8);
    CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> iss, don;
    for (int b = 0; b < grid; ++b) {
        if ((b & 3) < loaders) continue;
        for (int w = 0; w < 4; ++w) { iss.push_back(h[(b * 4 + w) * 2] / (double)reps); don.push_back(h[(b * 4 + w) * 2 + 1] / (double)reps); }
    }
    std::sort(iss.begin(), iss.end()); std::sort(don.begin(), don.end());
    const double storers = grid * (4 - loaders) / 4.0;
    const double bytes = storers * reps * 4.0 * NST * 256.0;
    printf("%s stores/wave=%3d blocks/CU=%d neighbours/4=%d mode=%d : issue median %7.2f us  p90 %7.2f | drained median %7.2f us | kernel %.3f ms  %.2f TB/s written\n",
           WIDE ? "x4" : "x1", WIDE ? NST / 4 : NST, blocks_per_cu, loaders, mode, iss[iss.size() / 2] * 0.01, iss[iss.size() * 9 / 10] * 0.01,
           don[don.size() / 2] * 0.01, ms, bytes / ms * 1e-9);
    return 0;
}

int main() {
    const long ld = 4096;
    float *out; long long *stamps; float *src;
    const size_t out_bytes = (size_t)1024 * 8 * 128 * ld * 4;       // grid * reps * 128 rows
    CK(hipMalloc(&out, out_bytes));
    CK(hipMalloc(&stamps, 1024 * 8 * 8));
    CK(hipMalloc(&src, (size_t)1 << 30));
    CK(hipMemset(src, 0, (size_t)1 << 30));
    for (int bpc = 1; bpc <= 4; bpc *= 2) {
        if (run<16, false>(bpc, 0, out, ld, stamps, src)) return 1;
        if (run<32, false>(bpc, 0, out, ld, stamps, src)) return 1;
        if (run<64, false>(bpc, 0, out, ld, stamps, src)) return 1;
        if (run<128, false>(bpc, 0, out, ld, stamps, src)) return 1;
        if (run<64, true>(bpc, 0, out, ld, stamps, src)) return 1;
    }
    for (int loaders = 1; loaders <= 3; ++loaders) {
        if (run<32, false>(4, loaders, out, ld, stamps, src)) return 1;
        if (run<64, false>(4, loaders, out, ld, stamps, src)) return 1;
        if (run<64, true>(4, loaders, out, ld, stamps, src)) return 1;
    }
    // neighbours shaped like GEMM main loops: 1 MFMA, 2 DMA (4 KB per wave and K tile), 4 LDS reads
    for (int mode : {1, 2, 3, 4, 5, 7}) {
        if (run<64, false>(4, 3, out, ld, stamps, src, mode)) return 1;
        if (run<64, true>(4, 3, out, ld, stamps, src, mode)) return 1;
    }
    return 0;
}
