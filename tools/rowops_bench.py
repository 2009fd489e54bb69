#!/usr/bin/env python3
"""Achieved HBM GB/s of the HBM-bound kernels at the C4/C5 tensor sizes (algorithmic bytes / HIP-event time), from COLD
caches (a 1 GB fill between repetitions: inside a step these tensors are 0.5-2 GB and never cache-resident), with the
default cache policy and with the nontemporal hint the product uses for tensors of >= 32 MB (NPM_TUNE_STREAM_NT)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from np_modeling_amd import device as D, _C
lib = _C.lib()
rng = np.random.default_rng(0)
M, F, U, S, BH = 131072, 1024, 4096, 512, 2048
x = D.from_host(rng.standard_normal(M * F, dtype=np.float32)).reshape(M, F)
dz = D.from_host(rng.standard_normal(M * F, dtype=np.float32)).reshape(M, F)
gamma = D.from_host(rng.standard_normal(F, dtype=np.float32)); beta = D.from_host(rng.standard_normal(F, dtype=np.float32))
z, mean, rstd = D.layernorm_fwd(x, gamma, beta, 1e-3)
dg, db = D.empty([F]), D.empty([F])
big = D.from_host(rng.standard_normal(M * U // 2, dtype=np.float32)); big2 = D.empty([M * U // 2]); big3 = D.empty([M * U // 2])
scores = D.empty([BH * S, S]); _C.check(lib.npm_fill_f32(scores.ptr, 0.5, scores.size)); scores2 = D.empty([BH * S, S])
flush = D.empty([M * 2048])
def run(fn, reps=5):
    ts = []
    for _ in range(reps):
        _C.check(lib.npm_set_tuning(12, 0)); _C.check(lib.npm_fill_f32(flush.ptr, 1.0, flush.size)); _C.check(lib.npm_set_tuning(12, cur)); D.synchronize()
        e0 = D.Event().record(); fn(); e1 = D.Event().record(); D.synchronize(); ts.append(e0.elapsed_ms(e1))
    return float(np.median(ts))
cases = [
 ('layernorm_fwd 131072x1024', 8.0 * M * F, lambda: D.layernorm_fwd(x, gamma, beta, 1e-3)),
 ('layernorm_bwd 131072x1024', 12.0 * M * F, lambda: D.layernorm_bwd(dz, x, mean, rstd, gamma, dg, db)),
 ('layernorm_bwd + residual', 16.0 * M * F, lambda: D.layernorm_bwd(dz, x, mean, rstd, gamma, dg, db, residual=z)),
 ('softmax_fwd 1M x 512', 8.0 * scores.size, lambda: D.softmax_fwd(scores, 0.125, out=scores2)),
 ('softmax_bwd 1M x 512', 12.0 * scores.size, lambda: D.softmax_bwd(scores2, scores, 0.125, out=scores)),
 ('relu_fwd 268M', 8.0 * big.size, lambda: D.relu_fwd(big, out=big2)),
 ('relu_bwd 268M', 12.0 * big.size, lambda: D.relu_bwd(big, big2, out=big3)),
 ('add 268M', 12.0 * big.size, lambda: D.add(big, big2, out=big3)),
 ('colsum 65536x4096', 4.0 * big.size, lambda: D.colsum(big, M // 2, U)),
 ('attn_rowdot [256,512,8,128]', 8.0 * M * F, lambda: D.attn_rowdot(x.reshape(256, 512, 8, 128), dz.reshape(256, 512, 8, 128))),
]
print(f'{"kernel (cold caches)":32s}   default policy        nontemporal')
for name, nbytes, fn in cases:
    out = []
    for cur in (0, 1):
        ms = run(fn); out.append(f'{ms:6.3f} ms {nbytes / ms / 1e6:6.0f} GB/s')
    print(f'{name:32s} {out[0]}   {out[1]}', flush=True)
