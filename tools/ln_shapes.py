#!/usr/bin/env python3
"""LayerNorm forward / backward from cold caches over the row length at a fixed element count (128 Mi): does the rate depend on the
registers a row costs (VPL = d / 256 float4 per lane)?"""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from np_modeling_amd import device as D, _C
lib = _C.lib()
rng = np.random.default_rng(0)
TOTAL = 131072 * 1024
flush = D.empty([TOTAL * 2])
def run(fn, reps=7):
    ts = []
    for _ in range(reps):
        _C.check(lib.npm_fill_f32(flush.ptr, 1.0, flush.size)); D.synchronize()
        e0 = D.Event().record(); fn(); e1 = D.Event().record(); D.synchronize(); ts.append(e0.elapsed_ms(e1))
    return float(np.median(ts))
xs = D.from_host(rng.standard_normal(TOTAL, dtype=np.float32)); dzs = D.from_host(rng.standard_normal(TOTAL, dtype=np.float32))
for F in (256, 512, 1024, 2048, 4096):
    M = TOTAL // F
    x, dz = xs.reshape(M, F), dzs.reshape(M, F)
    gamma = D.from_host(rng.standard_normal(F, dtype=np.float32)); beta = D.from_host(rng.standard_normal(F, dtype=np.float32))
    z, mean, rstd = D.layernorm_fwd(x, gamma, beta, 1e-3)
    dg, db = D.empty([2 * F]).flat_view(0, [F]), D.empty([F])
    f = run(lambda: D.layernorm_fwd(x, gamma, beta, 1e-3))
    a = run(lambda: D.layernorm_bwd(dz, x, mean, rstd, gamma, dg, db))
    b = run(lambda: D.layernorm_bwd(dz, x, mean, rstd, gamma, dg, db, residual=z))
    print(f'd {F:5d} rows {M:7d}: fwd {f:.3f} ms {8.0*TOTAL/f/1e6:6.0f} GB/s  bwd {a:.3f} ms {12.0*TOTAL/a/1e6:6.0f} GB/s   bwd+res {b:.3f} ms {16.0*TOTAL/b/1e6:6.0f} GB/s', flush=True)
