import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
import np_modeling_amd as npm
from np_modeling_amd import device as D
np.random.seed(0)
enc = npm.layers.TransformerEncoder(num_heads=8, hidden_units=4096, norm_first=True)
x = D.from_host(np.random.normal(size=[256, 512, 1024]).astype(np.float32))
dy = D.from_host(np.random.normal(size=[256, 512, 1024]).astype(np.float32))
for _ in range(3):
    enc(x); enc(dy, backprop=True, learning_rate=1e-3)
res = {}
for rep in range(6):
    with D.KernelTimer() as t:
        enc(x); enc(dy, backprop=True, learning_rate=1e-3)
    D.synchronize()
    tn = [(fl, s.elapsed_ms(e)) for n, fl, _, s, e in t.records if n == 'sgemm_TN']
    for i, (fl, ms) in enumerate(tn):
        res.setdefault(i, []).append((fl, ms))
for i, v in res.items():
    ms = np.median([m for _, m in v]); fl = v[0][0]
    print(f'TN #{i}: {fl/1e12:.2f} TFLOP  median {ms:.3f} ms  {fl/ms/1e9:.1f} TF')
