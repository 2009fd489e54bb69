#!/usr/bin/env python3
"""Every GEMM launch of the encoder step, timed INSIDE the step (HIP events per launch, medians over repetitions): which launch
of a layout family is the slow one.  (tools/gemm_bench.py times the shapes standalone, operands cold and epilogues plain.)
    python tools/gemm_instep.py"""
import os
import sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import np_modeling_amd as npm
from np_modeling_amd import device as D
import bench

f, h, u = 1024, 8, 4096
enc = npm.layers.TransformerEncoder(num_heads=h, hidden_units=u, norm_first=True)
rng = np.random.default_rng(0)
x = D.from_host(rng.standard_normal([256, 512, f], dtype=np.float32))
dy = D.from_host(rng.standard_normal([256, 512, f], dtype=np.float32) * np.float32(0.01))
enc(D.from_host(np.zeros([1, 8, f], dtype=np.float32)))
bench.bind(enc, bench.make_params(np.random.default_rng(0), f, h, u))
sgd = npm.optimizer.SGDOptimizer(1e-6)
for _ in range(4):
    enc(x); enc(dy, backprop=True, optimizer_=sgd)
res = {}
for rep in range(8):
    with D.KernelTimer() as t:
        enc(x); enc(dy, backprop=True, optimizer_=sgd)
    D.synchronize()
    for i, (n, fl, _, s, e) in enumerate(t.records):
        if n.startswith('sgemm') or n.startswith('mha') or n.startswith('layernorm'):
            res.setdefault((i, n, fl), []).append(s.elapsed_ms(e))
names = ['qkv fwd (bias)', 'out fwd (bias + residual)', 'dense1 fwd (bias, ReLU, saved pre)', 'dense2 fwd (bias + residual)',
         'dense2 dw (+ db)', 'dense2 dx (ReLU mask)', 'dense1 dw (+ db)', 'dense1 dx', 'out dw (+ db)', 'dctx = dy wo (row dots)',
         'qkv dw (+ db)', 'qkv dx (sum of three)']
gi = 0
for (i, n, fl), ms in sorted(res.items()):
    m = float(np.median(ms))
    label = ''
    if n.startswith('sgemm'):
        label = names[gi] if gi < len(names) else ''
        gi += 1
    rate = f'{fl / m / 1e9:7.1f} TF = {fl / m / 1e9 / 157.3:.3f}' if fl else ''
    print(f'#{i:2d} {n:16s} {m:7.3f} ms  {rate}  {label}')
