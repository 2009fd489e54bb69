#!/usr/bin/env python3
"""Which XCD does block b of a launch run on?  The tile order of the GEMM / convolution kernels (xcd_remap, npm_mfma_tile.h) gives
each XCD a contiguous run of tiles ASSUMING block b runs on XCD b mod 8.  Reads HW_REG_XCC_ID of every block of one traced GEMM launch
(npm_debug_gemm_trace) and prints how often that holds, per generation of blocks."""
import os, sys, collections
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from np_modeling_amd import device as D, _C
M, N, K = 131072, 1024, 1024
a, b, c = D.empty([M * K]), D.empty([K * N]), D.empty([M * N])
_C.check(_C.lib().npm_fill_f32(a.ptr, 0.5, a.size)); _C.check(_C.lib().npm_fill_f32(b.ptr, 0.25, b.size))
grid = (M // 128) * (N // 128)
buf = D._Buffer(grid * 64)
fn = lambda: D.gemm(M, N, K, D.Mat(a, K), D.Mat(b, N), D.Mat(c, N))
for _ in range(3): fn()
D.synchronize()
_C.check(_C.lib().npm_debug_gemm_trace(buf.ptr)); fn(); D.synchronize(); _C.check(_C.lib().npm_debug_gemm_trace(None))
host = np.empty(grid * 8, dtype=np.int64)
_C.check(_C.lib().npm_d2h(host.ctypes.data, buf.ptr, host.nbytes))
t = host.reshape(grid, 8)
xcc = (t[:, 1] & 0xF).astype(int)
bid = np.arange(grid)
print(f'{grid} blocks; XCC ids seen: {sorted(set(xcc))}; blocks per XCC: {[int((xcc == x).sum()) for x in sorted(set(xcc))]}')
print(f'xcc == block mod 8 for {100.0 * np.mean(xcc == bid % 8):.2f} % of the blocks')
for lo in (0, 1024, 4096, grid - 1024):
    sl = slice(lo, lo + 1024)
    print(f'  blocks {lo:5d}..{lo + 1023:5d}: {100.0 * np.mean(xcc[sl] == bid[sl] % 8):6.2f} %   first 16 xcc ids: {xcc[lo:lo + 16].tolist()}')
conf = collections.Counter(zip((bid % 8).tolist(), xcc.tolist()))
print('most common (block mod 8 -> xcc):', sorted(conf.items(), key=lambda kv: -kv[1])[:12])
