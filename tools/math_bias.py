#!/usr/bin/env python3
"""Signed error of the GEMM math modes (0 f32 MFMA, 1 / 2 bf16 three-way split with one / two accumulators, 3 scaled
two-way fp16 split) against fp64: mean and rms of (C - ref), plus rows and elements of widely different magnitude."""
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
cases['rows of A 2^40 apart'] = ((a * np.exp2(rng.integers(-20, 21, size=(m, 1)))).astype(np.float32), b)
cases['elements 2^30 apart'] = ((a * np.exp2(-rng.integers(0, 31, size=(1, k)))).astype(np.float32), b)
for name, (a, b) in cases.items():
    ref = a.astype(np.float64) @ b.astype(np.float64)
    rowmax = np.abs(ref).max(axis=1, keepdims=True)
    for math in (0, 1, 2, 3):
        err = run(a, b, math) - ref
        print(f'{name:24s} math={math}: mean err {err.mean():+.3e}  rms {np.sqrt((err**2).mean()):.3e}  max|ref| {np.abs(ref).max():.3g}  mean err / ulp(1) {err.mean() / 2**-23:+.3f}'
              f'  | relative to the output row maximum: rms {np.sqrt(((err / rowmax)**2).mean()):.3e} worst {np.abs(err / rowmax).max():.3e}')
