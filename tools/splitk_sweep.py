#!/usr/bin/env python3
"""Tall-K weight-gradient GEMMs (TN, K = 131072 tokens) over the number of K splits: the default (pick_splits: three resident blocks
per CU in one generation) against more, shorter K ranges per block."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from np_modeling_amd import device as D, _C
Mat = D.Mat
M, F, U = 131072, 1024, 4096
rng = np.random.default_rng(0)
x = D.from_host(rng.standard_normal(M * F, dtype=np.float32))
hbuf = D.from_host(rng.standard_normal(M * U, dtype=np.float32))
dw = D.empty([F * U]); db = D.empty([U])
shapes = {
    'qkv  M=3072 N=1024 (lda 3072)': (3 * F, F, Mat(hbuf, 3 * F), Mat(x, F), F, (0, 4, 8, 12, 16, 20, 24, 32, 0, 16)),
    'proj M=1024 N=1024 (lda 3072)': (F, F, Mat(hbuf, 3 * F), Mat(x, F), F, (0, 12, 16, 20, 28, 0, 16)),
    'ffn  M=1024 N=4096': (F, U, Mat(x, F), Mat(hbuf, U), U, (0, 3, 4, 8, 12, 16, 0, 4)),
    'ffn  M=4096 N=1024': (U, F, Mat(hbuf, U), Mat(x, F), F, (0, 3, 4, 8, 12, 16, 0, 12)),
}
for _ in range(3):
    D.gemm(F, U, M, Mat(x, F), Mat(hbuf, U), Mat(dw, U), trans_a=True)
for name, (m, n, a, b, ldc, splits) in shapes.items():
    for s in splits:
        fn = lambda: D.gemm(m, n, M, a, b, Mat(dw, ldc), trans_a=True, asum_out=db.flat_view(0, [m]), split_k=s)
        fn(); D.synchronize()
        e0 = D.Event().record()
        for _ in range(12):
            fn()
        e1 = D.Event().record(); D.synchronize()
        ms = e0.elapsed_ms(e1) / 12
        fl = 2.0 * m * n * M
        print(f'{name:32s} split_k {s:3d}: {ms:7.3f} ms  {fl / ms / 1e9:6.1f} TF ({fl / ms / 1e9 / 157.3:.1%})', flush=True)
