#!/usr/bin/env python3
"""Achieved HBM GB/s of the HBM-bound kernels at the C4/C5 tensor sizes (algorithmic bytes / HIP-event time)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from np_modeling_amd import device as D, _C
for kv in filter(None, (sys.argv[1] if len(sys.argv) > 1 else '').split(',')):
    a, b = kv.split('='); _C.check(_C.lib().npm_set_tuning(int(a), int(b)))
lib = _C.lib()
rng = np.random.default_rng(0)
M, F, U, S, BH = 131072, 1024, 4096, 512, 2048
x = D.from_host(rng.standard_normal(M * F, dtype=np.float32)).reshape(M, F)
dz = D.from_host(rng.standard_normal(M * F, dtype=np.float32)).reshape(M, F)
gamma = D.from_host(rng.standard_normal(F, dtype=np.float32)); beta = D.from_host(rng.standard_normal(F, dtype=np.float32))
big = D.from_host(rng.standard_normal(M * U // 2, dtype=np.float32))          # 1 GB
big2 = D.empty([M * U // 2]); big3 = D.empty([M * U // 2])
scores = D.empty([BH * S, S]); _C.check(lib.npm_fill_f32(scores.ptr, 0.5, scores.size))      # 2-D: rows x n
scores2 = D.empty([BH * S, S])
def run(name, nbytes, fn):
    fn(); D.synchronize(); e0 = D.Event().record()
    for _ in range(5): fn()
    e1 = D.Event().record(); D.synchronize(); ms = e0.elapsed_ms(e1) / 5
    print(f'{name:34s} {ms:7.3f} ms  {nbytes / ms / 1e6:7.0f} GB/s  ({100 * nbytes / ms / 1e6 / 8000:4.1f} % of 8 TB/s)')
z, mean, rstd = D.layernorm_fwd(x, gamma, beta, 1e-3)
dg, db = D.empty([F]), D.empty([F])
run('layernorm_fwd 131072x1024', 8.0 * M * F, lambda: D.layernorm_fwd(x, gamma, beta, 1e-3))
run('layernorm_bwd 131072x1024', 12.0 * M * F, lambda: D.layernorm_bwd(dz, x, mean, rstd, gamma, dg, db))
run('layernorm_bwd + residual', 16.0 * M * F, lambda: D.layernorm_bwd(dz, x, mean, rstd, gamma, dg, db, residual=z))
run('softmax_fwd 1M x 512', 8.0 * scores.size, lambda: D.softmax_fwd(scores, 0.125, out=scores2))
run('softmax_bwd 1M x 512', 12.0 * scores.size, lambda: D.softmax_bwd(scores2, scores, 0.125, out=scores))
n = big.size
run('relu_fwd 268M', 8.0 * n, lambda: D.relu_fwd(big, out=big2))
run('relu_bwd 268M', 12.0 * n, lambda: D.relu_bwd(big, big2, out=big3))
run('add 268M', 12.0 * n, lambda: D.add(big, big2, out=big3))
run('colsum 131072x4096', 4.0 * M * U / 2 * 1, lambda: D.colsum(big, M // 2, U))
run('attn_rowdot [256,512,8,128]', 8.0 * M * F, lambda: D.attn_rowdot(x.reshape(256, 512, 8, 128), dz.reshape(256, 512, 8, 128)))
