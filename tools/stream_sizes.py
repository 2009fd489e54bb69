#!/usr/bin/env python3
"""The elementwise `add` (2 reads + 1 write, nontemporal, one float4 per thread) from cold caches over the tensor size: what a
streaming kernel with LayerNorm backward's traffic mix reaches at LayerNorm backward's size (134 M elements)."""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from np_modeling_amd import device as D, _C
lib = _C.lib()
N = 131072 * 4096
a = D.empty([N]); b = D.empty([N]); c = D.empty([N])
for t in (a, b): _C.check(lib.npm_fill_f32(t.ptr, 0.5, t.size))
flush = D.empty([131072 * 2048])
def run(fn, reps=7):
    ts = []
    for _ in range(reps):
        _C.check(lib.npm_fill_f32(flush.ptr, 1.0, flush.size)); D.synchronize()
        e0 = D.Event().record(); fn(); e1 = D.Event().record(); D.synchronize(); ts.append(e0.elapsed_ms(e1))
    return float(np.median(ts))
for mi in (32, 64, 128, 256, 512):
    n = mi << 20
    ms = run(lambda: D.add(a.flat_view(0, [n]), b.flat_view(0, [n]), out=c.flat_view(0, [n])))
    ms2 = run(lambda: D.relu_fwd(a.flat_view(0, [n]), out=c.flat_view(0, [n])))
    print(f'{mi:4d} Mi elements: add {ms:.3f} ms {12.0*n/ms/1e6:6.0f} GB/s   relu_fwd {ms2:.3f} ms {8.0*n/ms2/1e6:6.0f} GB/s', flush=True)
