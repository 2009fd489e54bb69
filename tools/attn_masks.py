#!/usr/bin/env python3
"""The fused attention core under masks of different shapes at the C4 / C5 shape (B 256, H 8, S 512, D 128, saved scores):
what the tile summary (npm_mha_mask_summary) buys.  'first320' visits the same number of key tiles as 'causal' (10 of 16 per
query block) but evenly; 'all' is a mask that excludes nothing (every tile takes the unmasked path)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import np_modeling_amd as npm
from np_modeling_amd import device as D
from np_modeling_amd.device import Mat
b, h, s, d = 256, 8, 512, 128
rng = np.random.default_rng(0)
qkv = D.from_host(rng.standard_normal([b, s, 3, h, d], dtype=np.float32))
f = h * d
q, k, v = qkv, qkv.flat_view(f, [qkv.size - f]), qkv.flat_view(2 * f, [qkv.size - 2 * f])
dctx = D.from_host(rng.standard_normal([b, s, h, d], dtype=np.float32))
dqkv = D.empty([b, s, 3, h, d])
dq, dk, dv = dqkv, dqkv.flat_view(f, [dqkv.size - f]), dqkv.flat_view(2 * f, [dqkv.size - 2 * f])
dims = (b, h, s, s, d)
i, j = np.arange(s)[:, None], np.arange(s)[None, :]
masks = {'none': None, 'causal': j <= i, 'first320': np.broadcast_to(j < 320, (s, s)), 'first256': np.broadcast_to(j < 256, (s, s)), 'blockcausal128': (j // 128) <= (i // 128), 'all': np.ones((s, s), bool)}
for name, m in masks.items():
    mask = None if m is None else D.AttnMask(m[None, None], b, h, s, s)
    for save in (True,):
        for rep in range(4):
            with D.KernelTimer() as t:
                ctx, lse, scores = D.mha_core_fwd(Mat(q, 3 * f), Mat(k, 3 * f), Mat(v, 3 * f), dims, 0.088, mask=mask, save_scores=save)
            fw = list(t.summary().values())[0]['ms']
            with D.KernelTimer() as t:
                D.mha_core_bwd(Mat(q, 3 * f), Mat(k, 3 * f), Mat(v, 3 * f), ctx, lse, dctx, Mat(dq, 3 * f), Mat(dk, 3 * f), Mat(dv, 3 * f), dims, 0.088, mask=mask, scores=scores)
            bw = list(t.summary().values())[0]['ms']
        print(f'{name:14s} fwd {fw:.3f} ms  bwd {bw:.3f} ms', npm._C.last_attn_kernel(), flush=True)
