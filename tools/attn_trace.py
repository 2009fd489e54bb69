#!/usr/bin/env python3
"""Where one query tile of the attention backward spends its cycles: in-kernel s_memtime stamps at the phase
boundaries (npm_debug_attn_trace), median over blocks, at the C4 / C5 attention shape."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import np_modeling_amd as npm  # noqa: E402
from np_modeling_amd import _C, device as D  # noqa: E402
from np_modeling_amd.device import Mat  # noqa: E402

b, h, s, d = 256, 8, 512, 128
rng = np.random.default_rng(0)
qkv = D.from_host(rng.standard_normal([b, s, 3, h, d], dtype=np.float32))
f = h * d
q, k, v = qkv, qkv.flat_view(f, [qkv.size - f]), qkv.flat_view(2 * f, [qkv.size - 2 * f])
dctx = D.from_host(rng.standard_normal([b, s, h, d], dtype=np.float32))
dqkv = D.empty([b, s, 3, h, d])
dq, dk, dv = dqkv, dqkv.flat_view(f, [dqkv.size - f]), dqkv.flat_view(2 * f, [dqkv.size - 2 * f])
dims = (b, h, s, s, d)
scale = 1.0 / np.sqrt(d)
lib = _C.lib()
save = '--save-scores' in sys.argv
ftrace = D._Buffer(8 * 16 * b * h * 4)
for rep in range(3):
    _C.check(lib.npm_debug_attn_trace(ftrace.ptr))
    ctx, lse, scores = D.mha_core_fwd(Mat(q, 3 * f), Mat(k, 3 * f), Mat(v, 3 * f), dims, scale, save_scores=save)
    D.synchronize()
    _C.check(lib.npm_debug_attn_trace(None))
fh = np.zeros([b * h * 4, 16], dtype=np.int64)
_C.check(lib.npm_d2h(fh.ctypes.data, ftrace.ptr, fh.nbytes))
print('forward, one 32-key tile of one wave (two waves share a SIMD: 128 MFMAs of this wave = 8192 cycles alone, 16384 at half the pipe)')
for i, n in enumerate(['S^T = K Q^T (+next DMA)', 'online softmax', 'O^T += V^T P^T', 'to next tile start']):
    col = fh[:, i + 1] - fh[:, i]
    print(f'{n:26s} {np.median(col):7.0f} {np.percentile(col, 10):6.0f} {np.percentile(col, 90):6.0f}')
print(f'tile total                 {np.median(fh[:, 4] - fh[:, 0]):7.0f}')
print(f'block: prologue {np.median(fh[:, 10] - fh[:, 9]):.0f}, tile loop {np.median(fh[:, 12] - fh[:, 10]):.0f} '
      f'({s // 32} tiles), epilogue {np.median(fh[:, 13] - fh[:, 12]):.0f}, whole {np.median(fh[:, 13] - fh[:, 9]):.0f} cycles')
nblocks = b * h
if '--bwd16' in sys.argv:                      # the shipped 8-wave backward (head size 128, saved scores): NPM_TUNE 14=1 selects it under the trace
    _C.check(lib.npm_set_tuning(14, 1), 'npm_set_tuning')
    ctx, lse, scores = D.mha_core_fwd(Mat(q, 3 * f), Mat(k, 3 * f), Mat(v, 3 * f), dims, scale, save_scores=True)
    trace = D._Buffer(8 * 16 * nblocks)
    for rep in range(3):
        _C.check(lib.npm_debug_attn_trace(trace.ptr))
        D.mha_core_bwd(Mat(q, 3 * f), Mat(k, 3 * f), Mat(v, 3 * f), ctx, lse, dctx, Mat(dq, 3 * f), Mat(dk, 3 * f), Mat(dv, 3 * f), dims, scale, scores=scores)
        D.synchronize()
        _C.check(lib.npm_debug_attn_trace(None))
    assert _C.last_attn_kernel().startswith('mha_bwd16_kernel'), _C.last_attn_kernel()
    host = np.zeros([nblocks, 16], dtype=np.int64)
    _C.check(lib.npm_d2h(host.ctypes.data, trace.ptr, host.nbytes))
    names = ['wait + barrier (tile landed)', 'loads issued + dP = dO V^T', 'exp, dS -> LDS', 'dV', 'barrier (dS)', 'dQ (+next DMA)', 'dK (+dQ stores)']
    dt = np.diff(host[:, :8], axis=1)
    print('mha_bwd16_kernel, thread 0, key block 1 / query tile 5: phase   median   p10    p90   '
          '(cycles of s_memtime; 64 MFMAs of this wave = 2048 cycles alone, 4096 with the SIMD\'s other wave at the pipe rate)')
    for i, n in enumerate(names):
        col = dt[:, i]
        print(f'{n:30s} {np.median(col):7.0f} {np.percentile(col, 10):6.0f} {np.percentile(col, 90):6.0f}')
    tile = host[:, 7] - host[:, 0]
    print(f'tile total                     {np.median(tile):7.0f}   (4 x 4096 = 16384 at the pipe rate)')
    nt = (s // 32) * (s // 128)
    print(f'block: prologue {np.median(host[:, 10] - host[:, 9]):.0f}, all key blocks {np.median(host[:, 12] - host[:, 10]):.0f} '
          f'({nt} tiles = {np.median(host[:, 12] - host[:, 10]) / nt:.0f} per tile), drain {np.median(host[:, 13] - host[:, 12]):.0f}, '
          f'whole {np.median(host[:, 13] - host[:, 9]):.0f} cycles')
    sys.exit(0)
trace = D._Buffer(8 * 16 * nblocks)
for rep in range(3):
    _C.check(lib.npm_debug_attn_trace(trace.ptr))
    D.mha_core_bwd(Mat(q, 3 * f), Mat(k, 3 * f), Mat(v, 3 * f), ctx, lse, dctx, Mat(dq, 3 * f), Mat(dk, 3 * f), Mat(dv, 3 * f), dims, scale, scores=scores)
    D.synchronize()
    _C.check(lib.npm_debug_attn_trace(None))
host = np.zeros([nblocks, 16], dtype=np.int64)
_C.check(lib.npm_d2h(host.ctypes.data, trace.ptr, host.nbytes))
names = ['S = Q K^T (+next DMA)', 'dP = dO V^T (+exp)', 'dV (+dS to LDS)', 'barrier (dS)', 'dQ chain', 'dK (+dQ stores)', '-', 'to next tile start']
order = [0, 1, 2, 3, 4, 5, 6, 7, 8]
dt = np.diff(host[:, order], axis=1)
print('phase                      median   p10    p90   (cycles; 64 MFMAs = 4096 cycles at the pipe rate)')
for i, n in enumerate(names):
    col = dt[:, i]
    print(f'{n:26s} {np.median(col):7.0f} {np.percentile(col, 10):6.0f} {np.percentile(col, 90):6.0f}')
print(f'tile total                 {np.median(host[:, 8] - host[:, 0]):7.0f}   (5 x 4096 = 20480 at the pipe rate; 4 x 4096 with saved scores)')
nt = (s // 32) * (s // 128)
print(f'block: prologue {np.median(host[:, 10] - host[:, 9]):.0f}, first key block {np.median(host[:, 11] - host[:, 10]):.0f} ({s // 32} tiles), '
      f'all key blocks {np.median(host[:, 12] - host[:, 10]):.0f} ({nt} tiles = {np.median(host[:, 12] - host[:, 10]) / nt:.0f} per tile), '
      f'drain {np.median(host[:, 13] - host[:, 12]):.0f}, whole {np.median(host[:, 13] - host[:, 9]):.0f} cycles')
