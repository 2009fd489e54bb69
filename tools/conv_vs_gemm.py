#!/usr/bin/env python3
"""What the gather costs: the plain GEMMs with C3's implicit-GEMM shapes (forward [M, 576] x [576, 128] with bias + ReLU + saved
pre-activation; grad_x [M, 1152] x [1152, 64]) on a quarter of C3's pixels, beside the convolution kernels on the same pixels."""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from np_modeling_amd import device as D, _C
lib = _C.lib()
NB, H, W, C0, C1, KS = 64, 224, 224, 64, 128, 3
M = NB * H * W
def run(fn, flops, label):
    fn(); fn(); D.synchronize()
    e0 = D.Event().record()
    for _ in range(5): fn()
    e1 = D.Event().record(); D.synchronize()
    ms = e0.elapsed_ms(e1) / 5
    print(f'{label:64s} {ms:8.3f} ms {flops / ms / 1e9:7.1f} TF', flush=True)
a = D.empty([M * KS * KS * C1]); _C.check(lib.npm_fill_f32(a.ptr, 0.5, a.size))
w = D.empty([KS * KS * C1 * C1]); _C.check(lib.npm_fill_f32(w.ptr, 0.25, w.size))
bias = D.empty([C1]); _C.check(lib.npm_fill_f32(bias.ptr, 0.1, bias.size))
y = D.empty([M * C1]); pre = D.empty([M * C1])
K = KS * KS * C0
run(lambda: D.gemm(M, C1, K, D.Mat(a, K), D.Mat(w, C1), D.Mat(y, C1)), 2.0 * M * C1 * K, f'gemm NN M={M} N={C1} K={K}')
run(lambda: D.gemm(M, C1, K, D.Mat(a, K), D.Mat(w, C1), D.Mat(y, C1), bias=bias, relu_save=D.Mat(pre, C1)), 2.0 * M * C1 * K, '  + bias, ReLU, saved pre-activation')
K2 = KS * KS * C1
run(lambda: D.gemm(M, C0, K2, D.Mat(a, K2), D.Mat(w, C0), D.Mat(y, C0)), 2.0 * M * C0 * K2, f'gemm NN M={M} N={C0} K={K2}')
for k in (256, 1024, 4096):
    run(lambda k=k: D.gemm(M // 8, C1, k, D.Mat(a, k), D.Mat(w, C1), D.Mat(y, C1)), 2.0 * (M // 8) * C1 * k, f'gemm NN M={M // 8} N={C1} K={k}')
x = D.empty([NB, H, W, C0]); _C.check(lib.npm_fill_f32(x.ptr, 0.5, x.size))
filt = D.empty([KS, KS, C0, C1]); _C.check(lib.npm_fill_f32(filt.ptr, 0.25, filt.size))
import ctypes as C
def fwd(relu):
    desc = _C.npm_conv2d(n=NB, h=H, w=W, c_in=C0, c_out=C1, ksize=KS, x=x.ptr, filt=filt.ptr, bias=bias.ptr, y=y.ptr,
                         pre=pre.ptr if relu else None, relu=int(relu))
    _C.check(lib.npm_conv2d_fwd(C.byref(desc)))
run(lambda: fwd(True), 2.0 * M * C1 * K, 'conv2d_fwd + bias, ReLU, saved pre-activation')
run(lambda: fwd(False), 2.0 * M * C1 * K, 'conv2d_fwd + bias')
dx = D.empty([NB, H, W, C0])
run(lambda: _C.check(lib.npm_conv2d_bwd_x(y.ptr, filt.ptr, dx.ptr, NB, H, W, C0, C1, KS)), 2.0 * M * C0 * K2, 'conv2d_bwd_x')
