// Conv2D (NHWC x HWIO, SAME, stride 1, odd k) -- implicit-im2col GEMM.  Placeholder
// until the implicit-GEMM kernels land: reports NPM_E_UNSUPPORTED (never a CPU fallback).
#include "npm_internal.h"

extern "C" {

int npm_conv2d_fwd(const npm_conv2d *c) {
    NPM_REQUIRE_INIT();
    (void)c;
    return npm::fail(NPM_E_UNSUPPORTED, "npm_conv2d_fwd: not built yet");
}

int npm_conv2d_bwd_x(const float *, const float *, float *, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t) {
    NPM_REQUIRE_INIT();
    return npm::fail(NPM_E_UNSUPPORTED, "npm_conv2d_bwd_x: not built yet");
}

int npm_conv2d_bwd_w(const float *, const float *, float *, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t) {
    NPM_REQUIRE_INIT();
    return npm::fail(NPM_E_UNSUPPORTED, "npm_conv2d_bwd_w: not built yet");
}

}  // extern "C"
