#!/usr/bin/env python3
"""Times the fused attention core (npm_mha_core_fwd / npm_mha_core_bwd) and the GEMM composition it replaces at
the C4 / C5 attention shape (B 256, H 8, S 512, D 128 by default), fp32-MFMA peak 157.3 TF as the denominator.
FLOPs counted: forward 2 products, backward 4 (the recomputed q.k is not algorithmic work)."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import np_modeling_amd as npm  # noqa: E402
from np_modeling_amd import device as D  # noqa: E402
from np_modeling_amd.device import Mat  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--b', type=int, default=256)
ap.add_argument('--h', type=int, default=8)
ap.add_argument('--s', type=int, default=512)
ap.add_argument('--d', type=int, default=128)
ap.add_argument('--reps', type=int, default=5)
ap.add_argument('--save-scores', action='store_true')
ap.add_argument('--warm', type=int, default=0, help='untimed launches before the timed ones (reports the median)')
ap.add_argument('--tune', default='')
ap.add_argument('--mask', default='', choices=['', 'causal', 'random'], help='attention mask: causal [1,1,S,S] (broadcast over batch and head) or random per (b, h)')
a = ap.parse_args()
for item in filter(None, a.tune.split(',')):
    knob, value = item.split('=')
    npm._C.check(npm._C.lib().npm_set_tuning(int(knob), int(value)))
b, h, s, d = a.b, a.h, a.s, a.d
rng = np.random.default_rng(0)
qkv = D.from_host(rng.standard_normal([b, s, 3, h, d], dtype=np.float32))
f = h * d
q, k, v = qkv, qkv.flat_view(f, [qkv.size - f]), qkv.flat_view(2 * f, [qkv.size - 2 * f])
dctx = D.from_host(rng.standard_normal([b, s, h, d], dtype=np.float32))
dqkv = D.empty([b, s, 3, h, d])
dq, dk, dv = dqkv, dqkv.flat_view(f, [dqkv.size - f]), dqkv.flat_view(2 * f, [dqkv.size - 2 * f])
dims = (b, h, s, s, d)
scale = 1.0 / np.sqrt(d)
mask = None
if a.mask == 'causal':
    mask = D.AttnMask(np.tril(np.ones([s, s], dtype=bool))[None, None], b, h, s, s)
elif a.mask == 'random':
    m = rng.random([b, h, s, s]) < 0.7
    m[..., 0] = True
    mask = D.AttnMask(m, b, h, s, s)
    del m
print(f'B {b} H {h} S {s} D {d}, mask {a.mask or "none"}, scores {"saved" if a.save_scores else "recomputed"}')
def launch(name):
    global ctx, lse, scores
    if name == 'fwd':
        ctx, lse, scores = D.mha_core_fwd(Mat(q, 3 * f), Mat(k, 3 * f), Mat(v, 3 * f), dims, scale, mask=mask, save_scores=a.save_scores)
    else:
        D.mha_core_bwd(Mat(q, 3 * f), Mat(k, 3 * f), Mat(v, 3 * f), ctx, lse, dctx, Mat(dq, 3 * f), Mat(dk, 3 * f),
                       Mat(dv, 3 * f), dims, scale, mask=mask, scores=scores)


for name in ('fwd', 'bwd'):
    if a.warm:                                                   # back-to-back launches: the clock a loaded card holds
        with D.KernelTimer() as t:
            for rep in range(a.warm + a.reps):
                launch(name)
        D.synchronize()
        per_call = len(t.records) // (a.warm + a.reps)           # the backward books its row-term kernel too
        calls = [t.records[i * per_call:(i + 1) * per_call] for i in range(a.warm, a.warm + a.reps)]
        times = [sum(start.elapsed_ms(stop) for _, _, _, start, stop in c) for c in calls]
        flops = sum(fl for _, fl, _, _, _ in calls[0])
        ms = float(np.median(times))
        print(f'mha_core_{name} warm: median {ms:.3f} ms (min {min(times):.3f}) of {a.reps} after {a.warm} untimed  '
              f'{flops / ms / 1e9:.1f} TF  ({flops / ms / 1e9 / 157.3:.1%} of the fp32-MFMA peak)', flush=True)
        continue
    for rep in range(a.reps + 1):
        with D.KernelTimer() as t:
            if name == 'fwd':
                ctx, lse, scores = D.mha_core_fwd(Mat(q, 3 * f), Mat(k, 3 * f), Mat(v, 3 * f), dims, scale, mask=mask, save_scores=a.save_scores)
            else:
                D.mha_core_bwd(Mat(q, 3 * f), Mat(k, 3 * f), Mat(v, 3 * f), ctx, lse, dctx, Mat(dq, 3 * f), Mat(dk, 3 * f),
                               Mat(dv, 3 * f), dims, scale, mask=mask, scores=scores)
        rec = list(t.summary().values())[0]
        if rep:
            print(f'mha_core_{name}: {rec["ms"]:.3f} ms  {rec["flops"] / rec["ms"] / 1e9:.1f} TF  '
                  f'({rec["flops"] / rec["ms"] / 1e9 / 157.3:.1%} of the fp32-MFMA peak)', flush=True)
