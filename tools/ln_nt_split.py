#!/usr/bin/env python3
"""LayerNorm backward (+ fused residual, the form the pre-norm encoder step runs: 16 B per element) at the C5 size, by where the
nontemporal hint sits (NPM_TUNE_LN_NT_SPLIT: 0 loads and stores, 1 loads only, 2 stores only; NPM_TUNE_STREAM_NT=0: nowhere), from
cold caches and right behind the kernel that PRODUCED dz (in the step dz is what the dx GEMM before it has just written)."""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from np_modeling_amd import device as D, _C
lib = _C.lib()
rng = np.random.default_rng(0)
M, F = 131072, 1024
x = D.from_host(rng.standard_normal(M * F, dtype=np.float32)).reshape(M, F)
dz = D.from_host(rng.standard_normal(M * F, dtype=np.float32)).reshape(M, F)
a_, b_ = D.from_host(rng.standard_normal(M * F, dtype=np.float32)).reshape(M, F), D.zeros([M, F])
gamma = D.from_host(rng.standard_normal(F, dtype=np.float32)); beta = D.from_host(rng.standard_normal(F, dtype=np.float32))
z, mean, rstd = D.layernorm_fwd(x, gamma, beta, 1e-3)
dg = D.empty([2 * F]); db = dg.flat_view(F, [F]); dg = dg.flat_view(0, [F])
flush = D.empty([M * 2048])

def run(fn, prep, reps=9):
    ts = []
    for _ in range(reps):
        prep(); e0 = D.Event().record(); fn(); e1 = D.Event().record(); D.synchronize(); ts.append(e0.elapsed_ms(e1))
    return float(np.median(ts))

def cold():
    _C.check(lib.npm_fill_f32(flush.ptr, 1.0, flush.size)); D.synchronize()

def produced():          # dz written by the kernel right before (default cache policy for a 512 MB tensor: nontemporal stores ...)
    _C.check(lib.npm_fill_f32(flush.ptr, 1.0, flush.size))
    _C.check(lib.npm_add(a_.ptr, b_.ptr, dz.ptr, dz.size))

print(f'layernorm_bwd + residual, {M} x {F}: 16 B / element')
for name, nt, split in (('hint on loads and stores (shipped)', 1, 0), ('loads only', 1, 1), ('stores only', 1, 2), ('no hint', 0, 0)):
    _C.check(lib.npm_set_tuning(12, nt)); _C.check(lib.npm_set_tuning(19, split))
    fn = lambda: D.layernorm_bwd(dz, x, mean, rstd, gamma, dg, db, residual=z)
    a = run(fn, cold); b = run(fn, produced)
    print(f'{name:36s} cold {a:.3f} ms {16.0*M*F/a/1e6:6.0f} GB/s   behind its producer {b:.3f} ms {16.0*M*F/b/1e6:6.0f} GB/s', flush=True)
_C.check(lib.npm_set_tuning(12, 1)); _C.check(lib.npm_set_tuning(19, 0))
