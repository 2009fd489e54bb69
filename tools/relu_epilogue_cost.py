#!/usr/bin/env python3
"""What the saved pre-activation costs in the FFN GEMMs of the C5 step: dense1 forward with and without the second output
(NPM_EPI_RELU_SAVE vs NPM_EPI_RELU) and the dx GEMM with and without the mask read (NPM_EPI_RELU_MASK)."""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from np_modeling_amd import device as D, _C
lib = _C.lib()
M, F, U = 131072, 1024, 4096
x = D.empty([M * F]); _C.check(lib.npm_fill_f32(x.ptr, 0.5, x.size))
w1 = D.empty([F * U]); _C.check(lib.npm_fill_f32(w1.ptr, 0.01, w1.size))
b1 = D.empty([U]); _C.check(lib.npm_fill_f32(b1.ptr, -1.0, b1.size))
h = D.empty([M * U]); pre = D.empty([M * U]); dh = D.empty([M * U])
dy = D.empty([M * F]); _C.check(lib.npm_fill_f32(dy.ptr, 0.25, dy.size))
def gemm(epi, aux=None, **kw):
    g = _C.npm_gemm()
    for k, v in kw.items(): setattr(g, k, v)
    g.batch0 = g.batch1 = 1; g.alpha = 1.0; g.epilogue = epi
    if aux is not None: g.aux, g.ldaux = aux.ptr, U
    _C.check(lib.npm_sgemm(_C.C.byref(g)))
def run(fn, flops, label, reps=20):
    fn(); fn(); D.synchronize()
    e0 = D.Event().record()
    for _ in range(reps): fn()
    e1 = D.Event().record(); D.synchronize()
    ms = e0.elapsed_ms(e1) / reps
    print(f'{label:72s} {ms:8.3f} ms {flops / ms / 1e9:7.1f} TF', flush=True)
fl = 2.0 * M * F * U
fwd = dict(trans_a=0, trans_b=0, m=M, n=U, k=F, a=x.ptr, lda=F, b=w1.ptr, ldb=U, c=h.ptr, ldc=U, bias=b1.ptr)
bwd = dict(trans_a=0, trans_b=1, m=M, n=U, k=F, a=dy.ptr, lda=F, b=w1.ptr, ldb=F, c=dh.ptr, ldc=U)
for _ in range(2):
    run(lambda: gemm(_C.EPI_BIAS, **fwd), fl, 'dense1 forward: bias only')
    run(lambda: gemm(_C.EPI_BIAS | _C.EPI_RELU, **fwd), fl, 'dense1 forward: bias + ReLU')
    run(lambda: gemm(_C.EPI_BIAS | _C.EPI_RELU_SAVE, pre, **fwd), fl, 'dense1 forward: bias + ReLU + saved pre-activation (the step)')
    run(lambda: gemm(0, **bwd), fl, 'dh = dy @ w2^T: plain')
    run(lambda: gemm(_C.EPI_RELU_MASK, pre, **bwd), fl, 'dh = dy @ w2^T: masked by the saved pre-activation (the step)')
