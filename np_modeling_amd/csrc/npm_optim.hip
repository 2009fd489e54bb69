// "Next" rows of the scope table: the pieces around the layer hot path
// that keep a training step on the device -- Adam with the reference's numerics, the losses and
// the dropout mask application.  All HBM-bound, float4 / grid-stride, one pass over the data.
#include <algorithm>

#include "npm_internal.h"

namespace {

inline int grid_for(size_t n, int cap = 4096) {
    return (int)std::max<size_t>(1, std::min<size_t>((n + 255) / 256, (size_t)cap));
}

// reference optimizer.py:53-67 -- fp64 moments, bias correction, epsilon INSIDE the square root,
// the subtraction itself in fp64 (NumPy's `f32 -= f64` runs the fp64 loop) and ONE rounding to the fp32 parameter.
__global__ void __launch_bounds__(256)
adam_kernel(float *__restrict__ var, const float *__restrict__ grad, double *__restrict__ m, double *__restrict__ v,
            size_t n, double lr, double beta1, double beta2, double eps, double corr1, double corr2) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const double g = (double)grad[i];
        const double nm = beta1 * m[i] + (1.0 - beta1) * g;
        const double nv = beta2 * v[i] + (1.0 - beta2) * g * g;
        m[i] = nm;
        v[i] = nv;
        const double step = lr * ((nm / corr1) / sqrt(nv / corr2 + eps));
        var[i] = (float)((double)var[i] - step);  // numpy: float32 -= float64 array runs the fp64 loop and rounds ONCE
    }
}

// Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11): counter (c0..c3), key (k0, k1).
__device__ __forceinline__ void philox4x32_10(unsigned (&c)[4], unsigned k0, unsigned k1) {
#pragma unroll
    for (int round = 0; round < 10; ++round) {
        const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1;
        c[1] = (unsigned)p1;
        c[3] = (unsigned)p0;
        c[0] = n0;
        c[2] = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

// DropOut with the mask drawn on the device (reference normalizations.py:14-30 draws it with np.random.binomial on the
// host): elements 4 g .. 4 g + 3 take the four words of Philox4x32-10(counter = (g_lo, g_hi, offset_lo, offset_hi),
// key = seed); keep when word < threshold; y = keep ? x / keep_prob : 0, mask = keep as a byte.
__global__ void __launch_bounds__(256)
dropout_philox_kernel(const float *__restrict__ x, float *__restrict__ y, unsigned char *__restrict__ mask, size_t n,
                      float keep_prob, unsigned long long threshold, unsigned k0, unsigned k1, unsigned long long offset) {
    const size_t groups = (n + 3) / 4;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += (size_t)gridDim.x * blockDim.x) {
        unsigned c[4] = {(unsigned)g, (unsigned)(g >> 32), (unsigned)offset, (unsigned)(offset >> 32)};
        philox4x32_10(c, k0, k1);
        const size_t i = 4 * g;
        if (y == nullptr) {                  // the mask alone (its consumer applies it: npm_layernorm_dropout_fwd / _bwd)
            if (i + 3 < n && (((uintptr_t)mask) & 3) == 0) {
                *reinterpret_cast<unsigned *>(mask + i) = (unsigned)(c[0] < threshold) | ((unsigned)(c[1] < threshold) << 8) |
                                                          ((unsigned)(c[2] < threshold) << 16) | ((unsigned)(c[3] < threshold) << 24);
            } else {
                for (int e = 0; e < 4 && i + e < n; ++e) mask[i + e] = c[e] < threshold;
            }
        } else if (i + 3 < n && ((((uintptr_t)x) | ((uintptr_t)y)) & 15) == 0 && (((uintptr_t)mask) & 3) == 0) {
            const float4 v = *reinterpret_cast<const float4 *>(x + i);
            const bool b0 = c[0] < threshold, b1 = c[1] < threshold, b2 = c[2] < threshold, b3 = c[3] < threshold;
            *reinterpret_cast<float4 *>(y + i) = make_float4(b0 ? v.x / keep_prob : 0.f, b1 ? v.y / keep_prob : 0.f,
                                                              b2 ? v.z / keep_prob : 0.f, b3 ? v.w / keep_prob : 0.f);
            *reinterpret_cast<unsigned *>(mask + i) = (unsigned)b0 | ((unsigned)b1 << 8) | ((unsigned)b2 << 16) | ((unsigned)b3 << 24);
        } else {
            for (int e = 0; e < 4 && i + e < n; ++e) {
                const bool keep = c[e] < threshold;
                y[i + e] = keep ? x[i + e] / keep_prob : 0.f;
                mask[i + e] = keep;
            }
        }
    }
}

// per-block fp64 partial sums of f(a, b); a second launch of the same kernel reduces the partials
template <int MODE>   // 0: (a-b)^2   1: -b*log(a)
__global__ void __launch_bounds__(256)
pair_sum_kernel(const float *__restrict__ a, const float *__restrict__ b, size_t n, double *__restrict__ out) {
    __shared__ double red[256];
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        if (MODE == 0) { const double d = (double)a[i] - (double)b[i]; s += d * d; }
        else s -= (double)b[i] * log((double)a[i]);
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = red[0];
}

__global__ void __launch_bounds__(256)
sum_f64_kernel(const double *__restrict__ in, int n, double *__restrict__ out) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += in[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0];
}

__global__ void __launch_bounds__(256)
mse_bwd_kernel(const float *__restrict__ y, const float *__restrict__ t, float *__restrict__ dy, size_t n, float scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dy[i] = scale * (y[i] - t[i]);
}

__global__ void __launch_bounds__(256)
xent_bwd_kernel(const float *__restrict__ y, const float *__restrict__ t, float *__restrict__ dy, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dy[i] = -t[i] / y[i];
}

__global__ void __launch_bounds__(256)
mask_scale_kernel(const float *__restrict__ x, const unsigned char *__restrict__ mask, float *__restrict__ y,
                  size_t n, float keep) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        y[i] = mask[i] ? x[i] / keep : 0.f;
}

template <int MODE>
int pair_sum(const float *a, const float *b, size_t n, double *host_out) {
    hipStream_t s = npm::ctx().stream;
    const int blocks = grid_for(n, 1024);
    npm::Scratch part;
    int rc = part.alloc(sizeof(double) * (blocks + 1));
    if (rc) return rc;
    double *p = (double *)part.ptr;
    hipLaunchKernelGGL(pair_sum_kernel<MODE>, dim3(blocks), dim3(256), 0, s, a, b, n, p + 1);
    NPM_CHECK_LAUNCH();
    hipLaunchKernelGGL(sum_f64_kernel, dim3(1), dim3(256), 0, s, (const double *)(p + 1), blocks, p);
    NPM_CHECK_LAUNCH();
    return npm_d2h(host_out, p, sizeof(double));
}

}  // namespace

extern "C" {

int npm_adam_step(float *var, const float *grad, double *m, double *v, size_t n, double lr, double beta1,
                  double beta2, double eps, int step) {
    NPM_REQUIRE_INIT();
    NPM_ARG(step >= 1);
    if (n == 0) return NPM_OK;
    NPM_ARG(var && grad && m && v);
    const double corr1 = 1.0 - pow(beta1, (double)step), corr2 = 1.0 - pow(beta2, (double)step);
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, npm::ctx().stream, var, grad, m, v, n, lr, beta1,
                       beta2, eps, corr1, corr2);
    NPM_CHECK_LAUNCH();
    return NPM_OK;
}

int npm_fill_f64(double *dst, double value, size_t n) {
    NPM_REQUIRE_INIT();
    if (n == 0) return NPM_OK;
    NPM_ARG(dst != nullptr);
    if (value == 0.0) {
        NPM_HIP(hipMemsetAsync(dst, 0, n * sizeof(double), npm::ctx().stream));
        return NPM_OK;
    }
    return npm::fail(NPM_E_UNSUPPORTED, "npm_fill_f64: only 0.0 is supported");
}

int npm_mse_fwd(const float *y, const float *targets, size_t n, double *loss) {
    NPM_REQUIRE_INIT();
    NPM_ARG(loss != nullptr && n > 0 && y && targets);
    int rc = pair_sum<0>(y, targets, n, loss);
    if (rc) return rc;
    *loss /= (double)n;
    return NPM_OK;
}

int npm_mse_bwd(const float *y, const float *targets, float *dy, size_t n) {
    NPM_REQUIRE_INIT();
    if (n == 0) return NPM_OK;
    NPM_ARG(y && targets && dy);
    hipLaunchKernelGGL(mse_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, npm::ctx().stream, y, targets, dy, n,
                       (float)(2.0 / (double)n));
    NPM_CHECK_LAUNCH();
    return NPM_OK;
}

int npm_xent_fwd(const float *y, const float *targets, size_t n, double *loss) {
    NPM_REQUIRE_INIT();
    NPM_ARG(loss != nullptr && n > 0 && y && targets);
    return pair_sum<1>(y, targets, n, loss);
}

int npm_xent_bwd(const float *y, const float *targets, float *dy, size_t n) {
    NPM_REQUIRE_INIT();
    if (n == 0) return NPM_OK;
    NPM_ARG(y && targets && dy);
    hipLaunchKernelGGL(xent_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, npm::ctx().stream, y, targets, dy, n);
    NPM_CHECK_LAUNCH();
    return NPM_OK;
}

int npm_dropout_philox(const float *x, float *y, unsigned char *mask, size_t n, float keep_prob, uint64_t seed, uint64_t offset) {
    NPM_REQUIRE_INIT();
    if (n == 0) return NPM_OK;
    NPM_ARG(mask && ((x && y) || (!x && !y)) && keep_prob > 0.f && keep_prob <= 1.f);      // x = y = NULL: draw the mask only
    const size_t groups = (n + 3) / 4;
    // keep an element when its 32 random bits are below keep_prob * 2^32 (integer compare: reproducible anywhere)
    const double scaled = (double)keep_prob * 4294967296.0;
    const uint64_t threshold = scaled >= 4294967296.0 ? 4294967296ull : (uint64_t)scaled;
    hipLaunchKernelGGL(dropout_philox_kernel, dim3(grid_for(groups, 1 << 20)), dim3(256), 0, npm::ctx().stream, x, y, mask, n,
                       keep_prob, threshold, (unsigned)seed, (unsigned)(seed >> 32), offset);
    NPM_CHECK_LAUNCH();
    return NPM_OK;
}

int npm_mask_scale(const float *x, const unsigned char *mask, float *y, size_t n, float keep_prob) {
    NPM_REQUIRE_INIT();
    if (n == 0) return NPM_OK;
    NPM_ARG(x && mask && y && keep_prob > 0.f);
    hipLaunchKernelGGL(mask_scale_kernel, dim3(grid_for(n)), dim3(256), 0, npm::ctx().stream, x, mask, y, n, keep_prob);
    NPM_CHECK_LAUNCH();
    return NPM_OK;
}

}  // extern "C"
