#!/usr/bin/env python3
"""Throughput of the other BASELINE.json configs on one MI355X (C2 Dense+ReLU, C3 Conv2D, C4 MHA).

    python tools/config_bench.py [--steps 3] [--only C3] [--kernels] [--cpu]

Each line: fwd+bwd(+SGD) samples/s, achieved TFLOP/s from the algorithmic FLOPs of SURVEY.md 8d and
the fraction of the fp32-MFMA peak (157.3 TFLOP/s).  bench.py is the headline (C5); it imports
:func:`run_config` / :func:`cpu_baseline` from here for the ``configs`` object of its JSON line, so the numbers
of this tool and the driver-timed ones come from the same code."""

import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

PEAK = 157.3
MIN_SECONDS = 0.5
NAMES = ('C2', 'C3', 'C4')
EXTRA = ('DEC', 'C4M', 'C4D64', 'C5D', 'C5')   # not BASELINE.json configs (`--only NAME`): TransformerDecoder at size; C4 under a
#                                 causal mask (the tile summary at work); C4's dimensions with 16 heads of 64; the headline encoder shard with
#                                 drop_rate 0.1 (device-drawn masks, applied inside the LayerNorm kernels) and, for an in-session twin, without


def timed(fn, steps, D, min_seconds=None, kernels=False):
    """(seconds per step, per-kernel summary or None).  The card needs tens of milliseconds of continuous load to
    settle at its working clock: a 3-step region of C2 (10 ms) read 125 TF where 200 steps read 143
    (profiles/r02_config_bench.log), so warm up for min_seconds / 2 and time at least min_seconds."""
    min_seconds = MIN_SECONDS if min_seconds is None else min_seconds
    fn()
    D.synchronize()
    t0 = time.perf_counter()
    fn()
    D.synchronize()
    one = max(time.perf_counter() - t0, 1e-4)
    for _ in range(int(min_seconds / 2 / one) + 1):
        fn()
    D.synchronize()
    steps = max(steps, int(min_seconds / one) + 1)
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    D.synchronize()
    sec = (time.perf_counter() - t0) / steps
    summary = None
    if kernels:                                   # one more step under the event timer: per-kernel table
        with D.KernelTimer() as timer:
            fn()
        summary = timer.summary()
    return sec, steps, summary


def _fill(D, array, rng, scale=1.0, chunk_rows=8):
    """Fill a big device tensor from one small host chunk (no multi-GB host array)."""
    rows = array.shape[0]
    per = int(np.prod(array.shape[1:]))
    chunk = rng.standard_normal([chunk_rows] + list(array.shape[1:]), dtype=np.float32) * np.float32(scale)
    for i in range(0, rows, chunk_rows):
        n = min(chunk_rows, rows - i)
        array.flat_view(i * per, [n] + list(array.shape[1:])).set(chunk[:n])


def build_config(name, npm, D, rng, conv_batch=256):
    """(label, batch, algorithmic FLOP per step, step function) of one BASELINE.json config at FULL size."""
    sgd = npm.optimizer.SGDOptimizer(1e-6)
    if name == 'C2':
        b = 4096
        layer = npm.layers.Dense(units=4096)
        x = D.from_host(rng.standard_normal([b, 4096], dtype=np.float32))
        dy = D.from_host(rng.standard_normal([b, 4096], dtype=np.float32))
        layer(x)
        layer.linear._w = (np.asarray(layer.linear.w) / 64).astype(np.float32)
        label, flops = 'Dense(4096->4096)+ReLU fwd+bwd+SGD, batch 4096 (BASELINE.json configs[1])', 3 * 2.0 * b * 4096 * 4096
    elif name == 'C3':
        b = conv_batch
        layer = npm.layers.Conv2D(channels=128, kernel_size=3)
        x = D.empty([b, 224, 224, 64])
        _fill(D, x, rng)
        dy = D.empty([b, 224, 224, 128])
        _fill(D, dy, rng, 0.01)
        layer(x)
        layer._w = (np.asarray(layer.w) / 24).astype(np.float32)
        label = f'Conv2D(64->128, k=3, SAME)+ReLU on 224x224 NHWC fwd+bwd+SGD, batch {b} (BASELINE.json configs[2])'
        flops = 3 * 2.0 * b * 224 * 224 * 128 * 576
    elif name == 'C4':
        b, s, f, h = 256, 512, 1024, 8
        layer = npm.layers.MultiHeadAttention(num_heads=h)
        x = D.from_host(rng.standard_normal([b, s, f], dtype=np.float32))
        dy = D.from_host(rng.standard_normal([b, s, f], dtype=np.float32) * np.float32(0.01))
        layer(x)
        for n in ('_wq', '_wk', '_wv', '_wo'):     # softmaxes that are not one-hot; scaled IN PLACE: rebinding a parameter to a new
            w = getattr(layer, n)                   # array would take wq / wk / wv out of their packed buffer (three projection GEMMs
            w *= np.float32(1.0 / 32)               # instead of one, layers/attentions.py _params_adjacent)
        label = 'MultiHeadAttention d_model=1024 heads=8 seq=512 fwd+bwd+SGD, batch 256 (BASELINE.json configs[3])'
        flops = 12 * 2.0 * b * s * f * f + 6 * 2.0 * b * s * s * f
    elif name in ('C4M', 'C4D64'):
        b, s, f = 256, 512, 1024
        h = 8 if name == 'C4M' else 16
        layer = npm.layers.MultiHeadAttention(num_heads=h)
        x = D.from_host(rng.standard_normal([b, s, f], dtype=np.float32))
        dy = D.from_host(rng.standard_normal([b, s, f], dtype=np.float32) * np.float32(0.01))
        mask = D.AttnMask(np.tril(np.ones([s, s], dtype=bool))[None, None], b, h, s, s) if name == 'C4M' else None   # made once
        layer(x)
        for n in ('_wq', '_wk', '_wv', '_wo'):     # softmaxes that are not one-hot; scaled IN PLACE: rebinding a parameter to a new
            w = getattr(layer, n)                   # array would take wq / wk / wv out of their packed buffer (three projection GEMMs
            w *= np.float32(1.0 / 32)               # instead of one, layers/attentions.py _params_adjacent)
        label = (f'MultiHeadAttention d_model=1024 heads={h} seq=512 fwd+bwd+SGD, batch 256' +
                 (', CAUSAL mask (reference attentions.py:105-107; TFLOP/s count the UNMASKED products: above the peak means '
                  'skipped tiles)' if mask is not None else '') + ' (not a BASELINE.json config)')
        flops = 12 * 2.0 * b * s * f * f + 6 * 2.0 * b * s * s * f

        def step_masked():
            layer(x, mask=mask)
            layer(dy, backprop=True, optimizer_=sgd)
        return label, b, flops, step_masked
    elif name in ('C5D', 'C5'):
        # the headline workload (bench.py: per-GPU shard of BASELINE.json configs[4]) in this harness; C5D with dropout
        # (reference transformer.py:13,22-23: drop_rate is a constructor argument of the encoder)
        import bench
        b, s, f, h, u = 256, 512, 1024, 8, 4096
        rate = 0.1 if name == 'C5D' else 0.0
        layer = npm.layers.TransformerEncoder(num_heads=h, hidden_units=u, norm_first=True, drop_rate=rate)
        x = D.from_host(rng.standard_normal([b, s, f], dtype=np.float32))
        dy = D.from_host(rng.standard_normal([b, s, f], dtype=np.float32) * np.float32(0.01))
        if rate:
            npm.set_dropout_rng('device', seed=2024)        # masks drawn on the GPU (Philox4x32-10); 'host' = the reference's np.random.binomial
        layer(D.from_host(np.zeros([1, 8, f], dtype=np.float32)))
        bench.bind(layer, bench.make_params(np.random.default_rng(0), f, h, u))
        label = (f'TransformerEncoder d_model={f} heads={h} seq={s} hidden={u} pre-norm fwd+bwd+SGD, batch {b}' +
                 (f', drop_rate {rate} (device-drawn masks: two Philox mask launches per step, both DropOuts applied inside the '
                  'LayerNorm kernels behind them)' if rate else ' (bench.py\'s headline workload in this harness)') +
                 ' (not a BASELINE.json config)')
        flops = b * bench.encoder_flops_per_sample(s, f, h, u)

        def step_enc():
            layer(x)
            layer(dy, backprop=True, optimizer_=sgd)
        return label, b, flops, step_enc
    elif name == 'DEC':
        b, sq, skv, f, h, u = 64, 512, 1024, 1024, 8, 4096
        layer = npm.layers.TransformerDecoder(num_heads=h, hidden_units=u, norm_first=True)
        x = D.from_host(rng.standard_normal([b, sq, f], dtype=np.float32))
        kv = D.from_host(rng.standard_normal([b, skv, f], dtype=np.float32))
        dy = D.from_host(rng.standard_normal([b, sq, f], dtype=np.float32) * np.float32(0.01))
        layer(x, kv)
        for att in (layer._self_attention, layer._cross_attention):
            for n in ('_wq', '_wk', '_wv', '_wo'):
                setattr(att, n, (np.asarray(getattr(att, n)) / 32).astype(np.float32))
        layer._dense1._linear._w = (np.asarray(layer._dense1._linear._w) / 32).astype(np.float32)
        layer._dense2._w = (np.asarray(layer._dense2._w) / 64).astype(np.float32)
        label = (f'TransformerDecoder d_model={f} heads={h} Sq={sq} Skv={skv} hidden={u} pre-norm fwd+bwd+SGD, batch {b} '
                 '(reference transformer.py:95-203; not a BASELINE.json config)')
        # self-attention: 12 projections over Sq rows + 6 products S^2; cross-attention: q/out projections over Sq rows (6),
        # k/v projections over Skv rows (6), 6 products Sq x Skv; feed-forward: 6 GEMMs
        flops = (12 * 2.0 * b * sq * f * f + 6 * 2.0 * b * sq * sq * f
                 + 6 * 2.0 * b * sq * f * f + 6 * 2.0 * b * skv * f * f + 6 * 2.0 * b * sq * skv * f
                 + 6 * 2.0 * b * sq * f * u)

        def step_dec():
            layer(x, kv)
            layer(dy, backprop=True, optimizer_=sgd)
        return label, b, flops, step_dec
    else:
        raise ValueError(name)

    def step():
        layer(x)
        layer(dy, backprop=True, optimizer_=sgd)
    return label, b, flops, step


def run_config(name, npm, D, steps=3, min_seconds=None, kernels=True, conv_batch=256):
    """One config -> the dict bench.py puts under ``configs[name]``."""
    label, batch, flops, step = build_config(name, npm, D, np.random.default_rng(0), conv_batch)
    try:
        sec, steps, summary = timed(step, steps, D, min_seconds, kernels)
    finally:
        npm.set_dropout_rng('host')
    out = {'workload': label, 'value': batch / sec, 'unit': 'samples/s', 'ms_per_step': 1e3 * sec, 'steps': steps,
           'tflops': flops / sec / 1e12, 'frac_of_fp32_mfma_peak': flops / sec / 1e12 / PEAK, 'math': npm._C.current_math()}
    if summary:
        rows = sorted(summary.items(), key=lambda kv: -kv[1]['ms'])
        out['kernels'] = {k: ({'launches': r['launches'], 'ms': r['ms'], 'tflops': r['flops'] / r['ms'] / 1e9} if r['flops'] else
                              {'launches': r['launches'], 'ms': r['ms'], 'GBps': r['bytes'] / r['ms'] / 1e6}) for k, r in rows}
        top, r = rows[0]
        out['dominant_kernel'] = {'name': top, 'share_of_step_time': r['ms'] / (1e3 * sec),
                                  'tflops': r['flops'] / r['ms'] / 1e9 if r['flops'] else None,
                                  'frac': r['flops'] / r['ms'] / 1e9 / PEAK if r['flops'] else None}
    return out


def cpu_baseline(name, budget_s=10.0):
    """The NumPy oracle (checker; closed-form flavour, BLAS GEMMs) timed on the host cores on a bounded sample of the
    config, per BASELINE.md section 3: C2 at the full shape, C3 at batch <= 8 (the reference's fp64 arithmetic, as
    conv.py computes), C4 at batch 2."""
    from oracle import np_oracle as O
    rng = np.random.default_rng(1)
    if name == 'C2':
        b = 4096
        x = rng.standard_normal([b, 4096], dtype=np.float32)
        dy = rng.standard_normal([b, 4096], dtype=np.float32)
        w = rng.standard_normal([4096, 4096], dtype=np.float32) / np.float32(64)
        bias = np.zeros([4096], dtype=np.float32)

        def step():
            y, pre = O.dense_fwd(x, w, bias)
            dx, dw, db = O.dense_bwd(x, w, pre, dy)
            O.sgd_step(w, dw, 1e-6)
            O.sgd_step(bias, db, 1e-6)
        sample = f'Dense(4096->4096)+ReLU fwd+bwd+SGD at the full shape, batch {b}, fp32 sgemm (mlp.py:21-40,70-77)'
    elif name == 'C3':
        b = 4
        x = rng.standard_normal([b, 224, 224, 64], dtype=np.float32)
        dy = rng.standard_normal([b, 224, 224, 128], dtype=np.float32)
        w = rng.standard_normal([3, 3, 64, 128], dtype=np.float32) / np.float32(24)
        bias = np.zeros([128], dtype=np.float32)

        def step():
            y, pre = O.conv_layer_fwd(x, w, bias)
            dx, dw, db = O.conv_layer_bwd(x, w, pre, dy)
            O.sgd_step(w, dw, 1e-6)
            O.sgd_step(bias, db, 1e-6)
        sample = (f'Conv2D(64->128,k=3) 224x224 fwd+bwd+SGD, batch {b} of 256 (the reference pads into fp64, conv.py:97: k*k '
                  'shifted dgemms; its temporaries are 40 GB at the full batch)')
    elif name == 'C4':
        b, s, f, h = 2, 512, 1024, 8
        p = {k: (rng.standard_normal(shape, dtype=np.float32) / np.float32(32 if k.startswith('w') else 1))
             for k, shape in dict(wq=[h, f // h, f], wk=[h, f // h, f], wv=[h, f // h, f], wo=[f, h, f // h],
                                  bq=[h, f // h], bk=[h, f // h], bv=[h, f // h], bo=[f]).items()}
        q = rng.standard_normal([b, s, f], dtype=np.float32)
        dy = rng.standard_normal([b, s, f], dtype=np.float32) * np.float32(0.01)

        def step():
            _, cache = O.mha_fwd(p, q)
            _, grads = O.mha_bwd(p, cache, dy)
            for k in p:
                O.sgd_step(p[k], grads[k], 1e-6)
        sample = (f'MultiHeadAttention d=1024 h=8 seq=512 fwd+bwd+SGD, batch {b} of 256, closed-form softmax gradient and BLAS '
                  'contractions (the reference\'s own einsum + Jacobian form takes 174 s at this batch, BASELINE.md section 2)')
    else:
        raise ValueError(name)
    times = []
    t_all = time.perf_counter()
    while len(times) < 3 and (not times or time.perf_counter() - t_all + times[-1] < budget_s):
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
    best = min(times)
    out = {'value': b / best, 'unit': 'samples/s', 'cores': os.cpu_count(), 'kind': 'port',
           'sample': f'NumPy oracle: {sample}; best of {len(times)} steps ({best:.2f} s/step)'}
    if name == 'C4':
        # Line (1) of BASELINE.md section 3 for this config: the reference's OWN formulation (np.einsum contractions -- single-
        # threaded C loops, no BLAS -- and the explicit [rows, n, n] fp64 softmax Jacobian, attentions.py:67-199 with
        # activations.py:32-45), restated verbatim by the oracle.  Its temporaries grow with seq^3 (2.2 TB at the full shape), so
        # the sample is ONE sequence of 128 tokens at the full width; quoted as tokens/s and, divided by 512, as the samples/s a
        # full-length sample could at best reach.
        sv = 128
        qv = rng.standard_normal([1, sv, f], dtype=np.float32)
        dv = rng.standard_normal([1, sv, f], dtype=np.float32) * np.float32(0.01)
        t0 = time.perf_counter()
        _, cache = O.mha_fwd(p, qv, verbatim=True)
        _, grads = O.mha_bwd(p, cache, dv, verbatim=True)
        for k in p:
            O.sgd_step(p[k], grads[k], 1e-6)
        tv = time.perf_counter() - t0
        out['reference_verbatim'] = {
            'value': sv / tv / s, 'unit': 'samples/s', 'tokens_per_s': sv / tv, 'cores': 1, 'kind': 'port',
            'sample': f'reference-verbatim formulation (einsum contractions, fp64 softmax Jacobian) MultiHeadAttention fwd+bwd+SGD, '
                      f'1 x seq {sv} x d {f}, one step ({tv:.1f} s); samples/s = tokens/s / {s} (an upper bound: the Jacobian grows '
                      'with seq^3)'}
    elif name == 'C3':
        out['reference_verbatim'] = {'same_as': 'the line above: the oracle\'s Conv2D IS the reference\'s formulation (fp64 padded copy, '
                                                'k*k shifted matmuls, conv.py:97-105,185-194), only run on a batch that fits'}
    elif name == 'C2':
        out['reference_verbatim'] = {'same_as': 'the line above: Dense is np.matmul in the reference too (mlp.py:23,35,36)'}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=3, help='at least this many timed steps (and at least --min-seconds of them)')
    ap.add_argument('--min-seconds', type=float, default=0.5)
    ap.add_argument('--only', default='')
    ap.add_argument('--conv-batch', type=int, default=256)
    ap.add_argument('--kernels', action='store_true', help='per-kernel HIP-event table after each line')
    ap.add_argument('--cpu', action='store_true', help='also time the NumPy oracle on the host cores (bounded samples)')
    args = ap.parse_args()
    import np_modeling_amd as npm
    from np_modeling_amd import device as D

    for name in NAMES + EXTRA:
        if (args.only and args.only != name) or (not args.only and name in EXTRA):
            continue
        r = run_config(name, npm, D, args.steps, args.min_seconds, args.kernels, args.conv_batch)
        for k, row in (r.get('kernels') or {}).items():
            rate = f"{row['tflops']:7.1f} TFLOP/s" if 'tflops' in row else f"{row['GBps']:7.0f} GB/s   "
            print(f"    {k:<22s} x{row['launches']:<3d} {row['ms']:8.3f} ms  {rate}", flush=True)
        print(f"{name} {r['workload']}: {r['value']:10.1f} samples/s  {r['ms_per_step']:8.2f} ms/step  {r['tflops']:6.1f} TFLOP/s "
              f"({100 * r['frac_of_fp32_mfma_peak']:4.1f} % of fp32-MFMA peak)", flush=True)
        D.trim_pool()
        if args.cpu:
            c = cpu_baseline(name)
            print(f"    cpu_baseline: {c['value']:.3f} samples/s on {c['cores']} cores -- {c['sample']}", flush=True)
            rv = c.get('reference_verbatim', {})
            if 'value' in rv:
                print(f"    reference-verbatim: {rv['value']:.4f} samples/s on {rv['cores']} core -- {rv['sample']}", flush=True)


if __name__ == '__main__':
    main()
