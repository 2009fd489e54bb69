#!/usr/bin/env python3
"""Throughput of the other BASELINE.json configs on one MI355X (C2 Dense+ReLU, C3 Conv2D, C4 MHA).

    python tools/config_bench.py [--steps 3] [--only C3]

Each line: fwd+bwd(+SGD) samples/s, achieved TFLOP/s from the algorithmic FLOPs of SURVEY.md 8d and
the fraction of the fp32-MFMA peak (157.3 TFLOP/s).  bench.py is the headline (C5)."""

import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

PEAK = 157.3


KERNELS = False


MIN_SECONDS = 0.5


def timed(fn, steps, D):
    """Seconds per step.  The card needs tens of milliseconds of continuous load to settle at its working clock: a
    3-step region of C2 (10 ms) read 125 TF where 200 steps read 143 (profiles/r02_config_bench.log), so warm up for
    MIN_SECONDS / 2 and time at least MIN_SECONDS."""
    fn()
    D.synchronize()
    t0 = time.perf_counter()
    fn()
    D.synchronize()
    one = max(time.perf_counter() - t0, 1e-4)
    for _ in range(int(MIN_SECONDS / 2 / one) + 1):
        fn()
    D.synchronize()
    steps = max(steps, int(MIN_SECONDS / one) + 1)
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    D.synchronize()
    sec = (time.perf_counter() - t0) / steps
    if KERNELS:                                   # one more step under the event timer: per-kernel table
        with D.KernelTimer() as timer:
            fn()
        rows = sorted(timer.summary().items(), key=lambda kv: -kv[1]['ms'])
        for name, r in rows:
            rate = (f"{r['flops'] / r['ms'] / 1e9:7.1f} TFLOP/s" if r['flops'] else
                    f"{r['bytes'] / r['ms'] / 1e6:7.0f} GB/s   ")
            print(f"    {name:<22s} x{r['launches']:<3d} {r['ms']:8.3f} ms  {rate}", flush=True)
    return sec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=3, help='at least this many timed steps (and at least --min-seconds of them)')
    ap.add_argument('--min-seconds', type=float, default=0.5)
    ap.add_argument('--only', default='')
    ap.add_argument('--conv-batch', type=int, default=256)
    ap.add_argument('--kernels', action='store_true', help='per-kernel HIP-event table after each line')
    args = ap.parse_args()
    global KERNELS, MIN_SECONDS
    KERNELS = args.kernels
    MIN_SECONDS = args.min_seconds
    import np_modeling_amd as npm
    from np_modeling_amd import device as D
    rng = np.random.default_rng(0)
    sgd = npm.optimizer.SGDOptimizer(1e-6)

    def report(name, batch, flops, sec):
        print(f'{name}: {batch / sec:10.1f} samples/s  {sec * 1e3:8.2f} ms/step  {flops / sec / 1e12:6.1f} TFLOP/s '
              f'({100 * flops / sec / 1e12 / PEAK:4.1f} % of fp32-MFMA peak)', flush=True)

    if not args.only or args.only == 'C2':
        b = 4096
        layer = npm.layers.Dense(units=4096)
        x = D.from_host(rng.standard_normal([b, 4096], dtype=np.float32))
        dy = D.from_host(rng.standard_normal([b, 4096], dtype=np.float32))
        layer(x)
        layer.linear._w = (np.asarray(layer.linear.w) / 64).astype(np.float32)

        def step():
            layer(x)
            layer(dy, backprop=True, optimizer_=sgd)
        report('C2 Dense(4096->4096)+ReLU b=4096', b, 3 * 2.0 * b * 4096 * 4096, timed(step, args.steps, D))

    if not args.only or args.only == 'C3':
        b = args.conv_batch
        layer = npm.layers.Conv2D(channels=128, kernel_size=3)
        x = D.empty([b, 224, 224, 64])
        chunk = rng.standard_normal([8, 224, 224, 64], dtype=np.float32)
        for i in range(0, b, 8):                   # fill the 3.3 GB input without a 3.3 GB host array
            x.flat_view(i * 224 * 224 * 64, [min(8, b - i), 224, 224, 64]).set(chunk[:min(8, b - i)])
        dy = D.empty([b, 224, 224, 128])
        chunk = rng.standard_normal([8, 224, 224, 128], dtype=np.float32) * np.float32(0.01)
        for i in range(0, b, 8):
            dy.flat_view(i * 224 * 224 * 128, [min(8, b - i), 224, 224, 128]).set(chunk[:min(8, b - i)])
        layer(x)
        layer._w = (np.asarray(layer.w) / 24).astype(np.float32)

        def step():
            layer(x)
            layer(dy, backprop=True, optimizer_=sgd)
        flops = 3 * 2.0 * b * 224 * 224 * 128 * 576
        report(f'C3 Conv2D(64->128,k=3) 224x224 b={b}', b, flops, timed(step, args.steps, D))

    if not args.only or args.only == 'C4':
        b, s, f, h = 256, 512, 1024, 8
        layer = npm.layers.MultiHeadAttention(num_heads=h)
        q = D.from_host(rng.standard_normal([b, s, f], dtype=np.float32))
        dy = D.from_host(rng.standard_normal([b, s, f], dtype=np.float32) * np.float32(0.01))
        layer(q)
        for n in ('_wq', '_wk', '_wv', '_wo'):
            setattr(layer, n, (np.asarray(getattr(layer, n)) / 32).astype(np.float32))

        def step():
            layer(q)
            layer(dy, backprop=True, optimizer_=sgd)
        flops = 12 * 2.0 * b * s * f * f + 6 * 2.0 * b * s * s * f
        report('C4 MultiHeadAttention d=1024 h=8 seq=512 b=256', b, flops, timed(step, args.steps, D))


if __name__ == '__main__':
    main()
