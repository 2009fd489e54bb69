#!/usr/bin/env python3
"""Per-block timeline of one LDS-DMA GEMM launch (npm_debug_gemm_trace): where blocks run (XCC / CU / SIMD),
when they start, how long prologue / main loop / epilogue take, and which blocks share a CU."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from np_modeling_amd import device as D, _C

# usage: gemm_trace.py [K]            -> 131072 x 1024 x K, NN
#        gemm_trace.py qk             -> the attention score GEMM: 2048 x (512 x 512 x 128), NT
#        gemm_trace.py c2             -> 4096^3, NN
if len(sys.argv) > 1 and sys.argv[1] == 'qk':
    M, N, K, NB, NT = 512, 512, 128, 2048, True
elif len(sys.argv) > 1 and sys.argv[1] == 'c2':          # C2's forward: 1024 tiles = one generation of 4 blocks per CU
    M, N, K, NB, NT = 4096, 4096, 4096, 1, False
elif len(sys.argv) > 3:                                  # gemm_trace.py M N K
    M, N, K, NB, NT = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), 1, False
else:
    M, N, K, NB, NT = 131072, 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 1, False
rng = np.random.default_rng(0)
def filled(n):
    out, chunk = D.empty([n]), rng.standard_normal(min(n, 1 << 22), dtype=np.float32)
    for o in range(0, n, chunk.size):
        m = min(chunk.size, n - o)
        out.flat_view(o, [m]).set(chunk[:m])
    return out
a, b = filled(NB * M * K), filled(NB * K * N)
c = D.empty([NB * M * N])
grid = NB * (M // 128) * (N // 128)
buf = D._Buffer(grid * 64)
if NT:
    fn = lambda: D.gemm(M, N, K, D.Mat(a, K, M * K), D.Mat(b, K, N * K), D.Mat(c, N, M * N), trans_b=True, batch=(NB, 1))
else:
    fn = lambda: D.gemm(M, N, K, D.Mat(a, K), D.Mat(b, N), D.Mat(c, N))
for _ in range(12): fn()
D.synchronize()
_C.check(_C.lib().npm_debug_gemm_trace(buf.ptr))
ev0 = D.Event().record(); fn(); ev1 = D.Event().record(); D.synchronize()
kernel_ms = ev0.elapsed_ms(ev1)
_C.check(_C.lib().npm_debug_gemm_trace(None))
host = np.empty(grid * 8, dtype=np.int64)
_C.check(_C.lib().npm_d2h(host.ctypes.data, buf.ptr, host.nbytes))
t = host.reshape(grid, 8)
hw, xcc = t[:, 0], t[:, 1] & 0xF
cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7; simd = (hw >> 4) & 3; wave_slot = hw & 0xF
t0 = t[:, 2].min()
start, first, loop, end, issued = (t[:, i] - t0 for i in (2, 3, 4, 5, 6))
real = t[:, 7]
us = float(np.median(real / np.maximum(end - start, 1))) * 0.01       # microseconds per s_memtime tick
print(f'K={K} blocks={grid}; s_memtime ticks, 1 tick = {us * 1000:.3f} ns')
print('in us: prologue %.2f | main loop %.2f (per k-tile %.3f) | epilogue issue %.2f, drained %.2f | lifetime %.1f | kernel span %.1f' % (
    np.median(first - start) * us, np.median(loop - first) * us, np.median(loop - first) / (K // 16) * us,
    np.median(issued - loop) * us, np.median(end - loop) * us, np.median(end - start) * us, float(end.max()) * us))
print('prologue (start->first tile landed): median %d  p90 %d' % (np.median(first - start), np.percentile(first - start, 90)))
print('main loop: median %d  (per k-tile %d)' % (np.median(loop - first), np.median(loop - first) / (K // 16)))
print('epilogue (loop end->stores drained): median %d p90 %d' % (np.median(end - loop), np.percentile(end - loop, 90)))
print('epilogue issue only (loop end->last store issued): median %d p90 %d' % (np.median(issued - loop), np.percentile(issued - loop, 90)))
real = t[:, 7]
print('shader clock over block lifetimes: median %.2f GHz (p10 %.2f, p90 %.2f); lifetime median %.1f us' % (np.median((end - start) / real) * 0.1, np.percentile((end - start) / real, 10) * 0.1, np.percentile((end - start) / real, 90) * 0.1, np.median(real) / 100))
print('block lifetime median %d ; kernel span %d' % (np.median(end - start), end.max()))
key = list(zip(xcc, se, sh, cu))
groups = collections.defaultdict(list)
for i, k in enumerate(key):
    groups[k].append(i)
print('distinct (xcc,se,sh,cu):', len(groups), ' blocks per CU over the launch: min %d max %d' % (min(map(len, groups.values())), max(map(len, groups.values()))))
# first-generation co-residents: for one CU list blocks with their start times
for k in list(groups)[:3]:
    ids = sorted(groups[k], key=lambda i: start[i])
    print('CU', k, [(int(i), int(start[i]), int(end[i])) for i in ids[:8]])
# are starts of co-resident blocks clustered?  for each CU sort by start; gaps between consecutive starts
gaps = []
for k, ids in groups.items():
    s = np.sort(start[ids])
    gaps.extend(np.diff(s))
gaps = np.array(gaps)
print('gap between consecutive block starts on a CU: p10 %d p50 %d p90 %d' % tuple(np.percentile(gaps, [10, 50, 90])))

# one-generation launches: who finishes when (a static partition lasts as long as its slowest CU)
per_xcc = collections.defaultdict(list)
for i in range(grid):
    per_xcc[int(xcc[i])].append(i)
print('per XCC: blocks, median main-loop ticks, median end, max end (us)')
for x in sorted(per_xcc):
    ids = per_xcc[x]
    print(f'  xcc {x}: {len(ids):5d}  loop {np.median((loop - first)[ids]) * us:8.1f}  end median {np.median(end[ids]) * us:8.1f}  max {end[ids].max() * us:8.1f}  start max {start[ids].max() * us:6.1f}')
e = np.sort(end) * us
print('block end times (us): p1 %.1f p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f' % tuple(np.percentile(e, [1, 10, 50, 90, 99, 100])))
print('main loop per k-tile (ns): p10 %.1f p50 %.1f p90 %.1f' % tuple(np.percentile((loop - first) / (K // 16) * us * 1000, [10, 50, 90])))

busy_us = float(real.sum()) / 100.0            # block lifetimes on the 100 MHz wall clock
clock_ghz = 1e-3 / us                       # s_memtime ticks are shader cycles
mfma_cycles = grid * (K // 16) * 32 * 64 * 4 / (len(groups) * 4)          # per SIMD: 32 f32 MFMAs of 64 cycles per wave and k-tile
print(f'matrix pipe: {mfma_cycles / (kernel_ms * 1e3 * clock_ghz * 1e3) * 100:.1f} % of the cycles of the launch at {clock_ghz:.2f} GHz '
      f'(nominal 2.4): cycles x clock = {mfma_cycles / (kernel_ms * 1e3 * clock_ghz * 1e3) * clock_ghz / 2.4 * 100:.1f} % of 157.3 TF')
print(f'traced launch {kernel_ms * 1e3:.1f} us = {2.0 * NB * M * N * K / kernel_ms * 1e-9:.1f} TF; sum of block lifetimes {busy_us:.0f} us '
      f'= {busy_us / (kernel_ms * 1e3) / len(groups):.2f} resident blocks per CU on average (4 fit)')
