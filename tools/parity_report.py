#!/usr/bin/env python3
"""Worst ELEMENTWISE-RELATIVE error per BASELINE.json config, beside the tensor-scaled metric the tests use.

BASELINE.json's bound is "1e-4 rel".  tests/conftest.py::assert_close holds |got - ref| <= tol (|ref| + max|ref|):
fp32 contractions reorder sums, so an element's error scales with the tensor's magnitude, and a purely relative
test is meaningless on elements that are tiny because large terms cancelled.  This report states both, once per
config and math mode: `rel` = max |got - ref| / |ref| over the NON-TINY elements, at two thresholds (|ref| >= 0.1 max|ref|
and >= 1e-3 max|ref|), and `scaled` = max |got - ref| / max|ref| over all elements.  The absolute error is uniform over a
tensor (a few 1e-6 of its maximum), so an element a thousand times smaller than the maximum shows a thousand times
the relative error: that is what the second threshold displays, not a loss of accuracy.  Reference = the NumPy oracle in fp64 on the same inputs (batch slices
of the full-shape problem where the oracle could not finish the whole batch in seconds).
    python tools/parity_report.py > profiles/r02_parity_relative_error.log"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import np_modeling_amd as npm  # noqa: E402
from np_modeling_amd import device as D  # noqa: E402
from oracle import np_oracle as O  # noqa: E402


def errs(got, ref):
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    top = np.abs(ref).max()
    rel = []
    for frac in (1e-1, 1e-3):
        big = np.abs(ref) >= frac * top
        rel.append(float((np.abs(got - ref)[big] / np.abs(ref)[big]).max()))
    return rel[0], rel[1], float(np.abs(got - ref).max() / top)


# tests/test_gpu_parity.py asserts these bounds per math mode: the reading of north_star's "1e-4 rel" this build commits to,
# {mode: (REL, SCALED)} as include/npm_hip.h states them (NPM_PARITY_*):
#   REL     elementwise relative error over the elements >= 0.1 of the tensor's largest magnitude
#   SCALED  |got - ref| / max|ref| over ALL elements
BOUNDS = npm._C.parity_bounds()
REL_BOUND, SCALED_BOUND = BOUNDS['f32']

ROWS = []                 # (config, mode, name, rel at 0.1, rel at 1e-3, scaled) of everything reported so far


def report(config, mode, items):
    for name, got, ref in items:
        rel1, rel3, scaled = errs(got, ref)
        ROWS.append((config, mode, name, rel1, rel3, scaled))
        print(f'{config:34s} {mode:7s} {name:10s} rel(|ref|>=0.1 max) {rel1:9.2e}   rel(|ref|>=1e-3 max) {rel3:9.2e}   scaled {scaled:9.2e}', flush=True)


class Rec:
    def __init__(self):
        self.g = {}

    def update(self, obj, attribute, gradient):
        self.g[attribute + str(len([k for k in self.g if k.startswith(attribute)]))] = np.asarray(gradient)


def init(rng, shape, scale=1.0):
    return (np.clip(rng.standard_normal(shape), -1, 1) * scale).astype(np.float32)


def main(modes=tuple(BOUNDS)):
    for mode in modes:
        npm.set_math(mode)
        rng = np.random.default_rng(0)
        # C1 / C2: Dense + ReLU
        for config, n, b in (('C1 Dense 512->512 b64', 512, 64), ('C2 Dense 4096->4096 b4096', 4096, 4096)):
            layer = npm.layers.Dense(units=n)
            x, dy = rng.standard_normal([b, n], dtype=np.float32), rng.standard_normal([b, n], dtype=np.float32)
            layer(np.zeros([1, n], dtype=np.float32))
            w, bias = init(rng, [n, n], 1 / np.sqrt(n)), init(rng, [n])
            layer.linear._w.set(w)
            layer.linear._b.set(bias)
            y = np.asarray(layer(x))
            rec = Rec()
            dx = np.asarray(layer(dy, backprop=True, optimizer_=rec))
            wy, pre = O.dense_fwd(x.astype(np.float64), w.astype(np.float64), bias.astype(np.float64))
            wdx, wdw, wdb = O.dense_bwd(x.astype(np.float64), w.astype(np.float64), pre, dy.astype(np.float64))
            safe = ~(np.abs(pre) < 1e-5).any(axis=1)      # rows whose ReLU branch agrees with the fp64 oracle's for sure
            report(config, mode, [('y', y, wy), ('dx', dx[safe], wdx[safe])] +
                   ([('dw', rec.g['_w0'], wdw), ('db', rec.g['_b0'], wdb)] if safe.all() else []))
        # C3: Conv2D 64 -> 128, k 3, 224 x 224 (two samples; convolution is per sample)
        layer = npm.layers.Conv2D(channels=128, kernel_size=3)
        x = rng.standard_normal([2, 224, 224, 64], dtype=np.float32)
        dy = rng.standard_normal([2, 224, 224, 128], dtype=np.float32) * np.float32(0.01)
        layer(np.zeros([1, 4, 4, 64], dtype=np.float32))
        w, bias = init(rng, [3, 3, 64, 128], 1 / 24.0), init(rng, [128])
        layer._w.set(w)
        layer._b.set(bias)
        y = np.asarray(layer(x))
        rec = Rec()
        dx = np.asarray(layer(dy, backprop=True, optimizer_=rec))
        wy, pre = O.conv_layer_fwd(x.astype(np.float64), w.astype(np.float64), bias.astype(np.float64))
        wdx, wdw, wdb = O.conv_layer_bwd(x.astype(np.float64), w.astype(np.float64), pre, dy.astype(np.float64))
        near = (np.abs(pre) < 4e-6).any(axis=-1)
        grown = near.copy()
        for di in (-1, 0, 1):
            for dj in (-1, 0, 1):
                grown |= np.roll(np.roll(near, di, axis=1), dj, axis=2)
        report('C3 Conv2D 64->128 k3 224^2 (b2)', mode, [('y', y, wy), ('dx', dx[~grown], wdx[~grown])])
        del x, dy, y, dx, wy, pre, wdx
        # C4: MHA d 1024, 8 heads, seq 512 (two samples)
        s, f, h, u = 512, 1024, 8, 4096
        att = npm.layers.MultiHeadAttention(num_heads=h)
        q = rng.standard_normal([2, s, f], dtype=np.float32)
        dy = rng.standard_normal([2, s, f], dtype=np.float32) * np.float32(0.01)
        att(np.zeros([1, 8, f], dtype=np.float32))
        p = {}
        for n in O.MHA_PARAM_NAMES:
            arr = init(rng, getattr(att, '_' + n).shape, 1 / np.sqrt(f) if n[0] == 'w' else 1.0)
            getattr(att, '_' + n).set(arr)
            p[n] = arr.astype(np.float64)
        out = np.asarray(att(q))
        rec = Rec()
        grads = [np.asarray(g) for g in att(dy, backprop=True, optimizer_=rec)]
        want, cache = O.mha_fwd(p, q.astype(np.float64))
        wg, pg = O.mha_bwd(p, cache, dy.astype(np.float64))
        report('C4 MHA d1024 h8 s512 (b2)', mode, [('out', out, want), ('dq+dk+dv', sum(grads), sum(wg)),
                                                  ('dwq', rec.g['_wq0'], pg['wq']), ('dwo', rec.g['_wo0'], pg['wo'])])
        # C5: TransformerEncoder (two samples of the per-GPU shard)
        enc = npm.layers.TransformerEncoder(num_heads=h, hidden_units=u, norm_first=True)
        enc(np.zeros([1, 8, f], dtype=np.float32))
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        params = bench.make_params(np.random.default_rng(0), f, h, u)
        bench.bind(enc, params)
        out = np.asarray(enc(q))
        dx = np.asarray(enc(dy, backprop=True, optimizer_=Rec()))
        p64 = {k: v.astype(np.float64) for k, v in params.items()}
        want, cache = O.encoder_fwd(p64, q.astype(np.float64), True)
        wdx, _ = O.encoder_bwd(p64, cache, dy.astype(np.float64), True)
        near = (np.abs(cache['d1_pre']) < 5e-6).any(axis=1).reshape(2, s)
        report('C5 TransformerEncoder (b2 of 256)', mode, [('out', out, want), ('dx', dx[~near], wdx[~near])])
    npm.set_math('f32')


if __name__ == '__main__':
    main()
