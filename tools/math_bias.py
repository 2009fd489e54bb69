#!/usr/bin/env python3
"""Signed error of the two GEMM math modes (f32 MFMA / bf16 three-way split) against fp64: mean and rms of (C - ref)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from np_modeling_amd import device as D, _C

def run(a, b, math):
    _C.check(_C.lib().npm_set_tuning(10, math))
    m, k = a.shape
    n = b.shape[1]
    c = D.empty([m, n])
    D.gemm(m, n, k, D.Mat(D.from_host(a), k), D.Mat(D.from_host(b), n), D.Mat(c, n))
    return c.numpy().astype(np.float64)

rng = np.random.default_rng(0)
m, n, k = 512, 512, 4096
cases = {}
a = rng.standard_normal((m, k), dtype=np.float32); b = (rng.standard_normal((k, n), dtype=np.float32) / 64).astype(np.float32)
cases['gaussian'] = (a, b)
tb = lambda x: (x.view(np.uint32) & np.uint32(0xffff0000)).view(np.float32)
cases['bf16-exact inputs'] = (tb(a.copy()), tb(b.copy()))
cases['positive'] = (np.abs(a), np.abs(b))
for name, (a, b) in cases.items():
    ref = a.astype(np.float64) @ b.astype(np.float64)
    for math in (0, 1, 2):
        err = run(a, b, math) - ref
        print(f'{name:20s} math={math}: mean err {err.mean():+.3e}  rms {np.sqrt((err**2).mean()):.3e}  max|ref| {np.abs(ref).max():.3g}  mean err / ulp(1) {err.mean() / 2**-23:+.3f}')
