#!/usr/bin/env python3
"""LayerNorm backward at the C5 size from cold caches, over the grid size (NPM_TUNE_LN_BWD_BLOCKS = blocks per CU): is the kernel
limited by bytes in flight?  (profiles/r03_ln_bwd_grid_sweep.log: flat, 5.4-5.6 TB/s from 2 to 32 blocks per CU.)"""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from np_modeling_amd import device as D, _C
lib = _C.lib()
rng = np.random.default_rng(0)
M, F = 131072, 1024
x = D.from_host(rng.standard_normal(M * F, dtype=np.float32)).reshape(M, F)
dz = D.from_host(rng.standard_normal(M * F, dtype=np.float32)).reshape(M, F)
gamma = D.from_host(rng.standard_normal(F, dtype=np.float32)); beta = D.from_host(rng.standard_normal(F, dtype=np.float32))
z, mean, rstd = D.layernorm_fwd(x, gamma, beta, 1e-3)
dg, db = D.empty([2 * F]).flat_view(0, [F]), D.empty([F])
flush = D.empty([M * 2048])
def run(fn, reps=7):
    ts = []
    for _ in range(reps):
        _C.check(lib.npm_fill_f32(flush.ptr, 1.0, flush.size)); D.synchronize()
        e0 = D.Event().record(); fn(); e1 = D.Event().record(); D.synchronize(); ts.append(e0.elapsed_ms(e1))
    return float(np.median(ts))
for blocks in (2, 3, 4, 6, 8, 12, 16, 32):
    _C.check(lib.npm_set_tuning(6, blocks))
    a = run(lambda: D.layernorm_bwd(dz, x, mean, rstd, gamma, dg, db))
    b = run(lambda: D.layernorm_bwd(dz, x, mean, rstd, gamma, dg, db, residual=z))
    print(f'blocks/CU {blocks:2d}: bwd {a:.3f} ms {12.0*M*F/a/1e6:6.0f} GB/s   bwd+res {b:.3f} ms {16.0*M*F/b/1e6:6.0f} GB/s', flush=True)
