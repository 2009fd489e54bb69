#!/usr/bin/env python3
"""Micro-benchmark of npm_sgemm at the GEMM shapes of the C5 encoder step (per-GPU batch 256).

    python tools/gemm_bench.py [--reps 5] [--only NAME]

Prints TFLOP/s (HIP events around `reps` back-to-back launches) per shape.  Used to A/B kernel
variants in one process and as the target of rocprofv3 --pmc runs."""

import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--only', default='')
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--tune', default='', help='comma list knob=value, e.g. 0=1,1=256')
    args = ap.parse_args()
    from np_modeling_amd import device as D, _C
    for kv in filter(None, args.tune.split(',')):
        knob, val = kv.split('=')
        _C.check(_C.lib().npm_set_tuning(int(knob), int(val)))
    print('tuning:', args.tune or 'default')
    B, S, F, H, U = args.batch, 512, 1024, 8, 4096
    M, Dk = B * S, F // H
    rng = np.random.default_rng(0)

    def buf(n):
        return D.from_host(rng.standard_normal(n, dtype=np.float32))

    x = buf(M * F)          # [M, F]
    hbuf = buf(M * U)       # [M, U]
    w_ff = buf(F * U)
    w_sq = buf(F * F)
    out_f = D.empty([M * F])
    out_u = D.empty([M * U])
    scores = D.empty([B * H * S * S])
    dw = D.empty([F * U])
    Mat = D.Mat
    shapes = {
        # name: (flops, callable)
        'proj_NT   M=131072 N=1024 K=1024': (2.0 * M * F * F, lambda: D.gemm(M, F, F, Mat(x, F), Mat(w_sq, F), Mat(out_f, F), trans_b=True)),
        'dx_NN     M=131072 N=1024 K=1024': (2.0 * M * F * F, lambda: D.gemm(M, F, F, Mat(x, F), Mat(w_sq, F), Mat(out_f, F))),
        'dw_TN     M=1024 N=1024 K=131072': (2.0 * M * F * F, lambda: D.gemm(F, F, M, Mat(x, F), Mat(out_f, F), Mat(dw, F), trans_a=True)),
        'ffn1_NN   M=131072 N=4096 K=1024': (2.0 * M * F * U, lambda: D.gemm(M, U, F, Mat(x, F), Mat(w_ff, U), Mat(out_u, U))),
        'ffn2_NN   M=131072 N=1024 K=4096': (2.0 * M * F * U, lambda: D.gemm(M, F, U, Mat(hbuf, U), Mat(w_ff, F), Mat(out_f, F))),
        'ffn_dx_NT M=131072 N=4096 K=1024': (2.0 * M * F * U, lambda: D.gemm(M, U, F, Mat(x, F), Mat(w_ff, F), Mat(out_u, U), trans_b=True)),
        'ffn_dx_NT M=131072 N=1024 K=4096': (2.0 * M * F * U, lambda: D.gemm(M, F, U, Mat(hbuf, U), Mat(w_ff, U), Mat(out_f, F), trans_b=True)),
        'ffn_dw_TN M=1024 N=4096 K=131072': (2.0 * M * F * U, lambda: D.gemm(F, U, M, Mat(x, F), Mat(hbuf, U), Mat(dw, U), trans_a=True)),
        'ffn_dw_TN M=4096 N=1024 K=131072': (2.0 * M * F * U, lambda: D.gemm(U, F, M, Mat(hbuf, U), Mat(x, F), Mat(dw, F), trans_a=True)),
        'qk_NT  2048x(512x512x128)': (2.0 * B * H * S * S * Dk, lambda: D.gemm(S, S, Dk, Mat(x, F, S * F, Dk), Mat(out_f, F, S * F, Dk), Mat(scores, S, H * S * S, S * S), trans_b=True, batch=(B, H))),
        'pv_NN  2048x(512x128x512)': (2.0 * B * H * S * S * Dk, lambda: D.gemm(S, Dk, S, Mat(scores, S, H * S * S, S * S), Mat(x, F, S * F, Dk), Mat(out_f, F, S * F, Dk), batch=(B, H))),
        'dv_TN  2048x(512x128x512)': (2.0 * B * H * S * S * Dk, lambda: D.gemm(S, Dk, S, Mat(scores, S, H * S * S, S * S), Mat(x, F, S * F, Dk), Mat(out_f, F, S * F, Dk), trans_a=True, batch=(B, H))),
    }
    # the packed q/k/v weight gradient of self-attention (layers/attentions.py: dqkv^T x with the bias gradients as column sums of
    # dqkv), as ONE product over the packed 3F axis and as three products over its thirds (row pitch 3F)
    dqkv = hbuf                                   # [M, 3F] lives in the first 3F/U of the [M, U] buffer's rows: view with pitch 3 F
    db3 = D.empty([3 * F])

    def qkv_three():
        for i in range(3):
            D.gemm(F, F, M, Mat(dqkv.flat_view(i * F, [dqkv.size - i * F]), 3 * F), Mat(x, F), Mat(dw.flat_view(i * F * F, [F * F]), F), trans_a=True,
                   asum_out=db3.flat_view(i * F, [F]))
    shapes['qkv_dw_TN M=3072 N=1024 K=131072 (one product)'] = (2.0 * M * 3 * F * F, lambda: D.gemm(3 * F, F, M, Mat(dqkv, 3 * F), Mat(x, F), Mat(dw, F), trans_a=True, asum_out=db3))
    shapes['qkv_dw_TN 3 x (M=1024 N=1024 K=131072, lda 3072)'] = (2.0 * M * 3 * F * F, qkv_three)
    shapes['out_dw_TN M=1024 N=1024 K=131072 + column sums'] = (2.0 * M * F * F, lambda: D.gemm(F, F, M, Mat(out_f, F), Mat(x, F), Mat(dw, F), trans_a=True, asum_out=db3.flat_view(0, [F])))
    n2 = 4096            # BASELINE configs[1]: Dense 4096 -> 4096, batch 4096 (one generation of 1024 tiles)
    a2, b2, c2 = buf(n2 * n2), buf(n2 * n2), D.empty([n2 * n2])
    shapes['c2_fwd_NN M=4096 N=4096 K=4096'] = (2.0 * n2 ** 3, lambda: D.gemm(n2, n2, n2, Mat(a2, n2), Mat(b2, n2), Mat(c2, n2)))
    shapes['c2_dx_NT  M=4096 N=4096 K=4096'] = (2.0 * n2 ** 3, lambda: D.gemm(n2, n2, n2, Mat(a2, n2), Mat(b2, n2), Mat(c2, n2), trans_b=True))
    shapes['c2_dw_TN  M=4096 N=4096 K=4096'] = (2.0 * n2 ** 3, lambda: D.gemm(n2, n2, n2, Mat(a2, n2), Mat(b2, n2), Mat(c2, n2), trans_a=True))
    total_ms, total_flops = 0.0, 0.0
    for _ in range(3):           # clocks up before the first measured shape
        list(shapes.values())[4][1]()
    for name, (flops, fn) in shapes.items():
        if args.only and args.only not in name:
            continue
        fn()
        D.synchronize()
        e0 = D.Event().record()
        for _ in range(args.reps):
            fn()
        e1 = D.Event().record()
        D.synchronize()
        ms = e0.elapsed_ms(e1) / args.reps
        total_ms += ms
        total_flops += flops
        print(f'{name:38s} {ms:8.3f} ms  {flops / ms / 1e9:7.1f} TFLOP/s  ({100 * flops / ms / 1e9 / 157.3:5.1f} % of 157.3)')
    print(f'{"sum":38s} {total_ms:8.3f} ms  {total_flops / total_ms / 1e9:7.1f} TFLOP/s')


if __name__ == '__main__':
    main()
