#!/usr/bin/env python3
"""Time vs K at fixed M, N (fixed output tiles): slope = main-loop cost, intercept = per-block fixed cost."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from np_modeling_amd import device as D, _C
for kv in filter(None, (sys.argv[1] if len(sys.argv) > 1 else '').split(',')):
    a, b = kv.split('='); _C.check(_C.lib().npm_set_tuning(int(a), int(b)))
M, N = 131072, 1024
rng = np.random.default_rng(0)
a = D.from_host(rng.standard_normal(M * 2048, dtype=np.float32))
b = D.from_host(rng.standard_normal(N * 2048, dtype=np.float32))
c = D.empty([M * N])
def run(fn, flops, label):
    fn(); fn(); D.synchronize()
    e0 = D.Event().record()
    for _ in range(5): fn()
    e1 = D.Event().record(); D.synchronize()
    ms = e0.elapsed_ms(e1) / 5
    print(f'{label:40s} {ms:8.3f} ms {flops / ms / 1e9:7.1f} TF', flush=True)
    return ms
for layout in ('NT', 'NN'):
    for K in (0, 16, 32, 64, 128, 256, 512, 1024, 2048):
        if layout == 'NT':
            fn = lambda K=K: D.gemm(M, N, K, D.Mat(a, 2048), D.Mat(b, 2048), D.Mat(c, N), trans_b=True)
        else:
            fn = lambda K=K: D.gemm(M, N, K, D.Mat(a, 2048), D.Mat(b, N), D.Mat(c, N))
        run(fn, 2.0 * M * N * K, f'{layout} M={M} N={N} K={K}')
