#!/usr/bin/env python3
"""Bounds the 8-GPU step on ONE GPU: the bench step (TransformerEncoder d 1024, seq 512, U 4096, batch 256, fwd + bwd + SGD) with a
stand-in kernel on a second stream doing, at the backward's four flush points, what one rank's share of an 8-rank ring all-reduce
does to the GPU (tools/exchange_shadow/shadow.hip: `channels` resident workgroups, 2 (R - 1) steps of count / R floats through the
local HBM, paced to a bus bandwidth) -- against the same step with no exchange at all.

    python tools/exchange_shadow.py [--steps 12] [--rounds 2] [--channels 8,16,32,64] [--busbw 0,150,300] [--lds 0,65536]

Prints one line per configuration: ms per step (median over rounds), slowdown against "none", the time the stand-in collectives held
their stream and the time the compute stream stood waiting for them per step.  What this does NOT contain: traffic on xGMI, RCCL's
own protocol overheads and launch latencies of a real peer, and skew between ranks.  profiles/r05_exchange_shadow.log is this
tool's output; DESIGN.md section 4.4 reads it."""
import argparse
import ctypes as C
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HERE = os.path.join(ROOT, 'tools', 'exchange_shadow')


def shadow_lib():
    so, src = os.path.join(HERE, 'libshadow.so'), os.path.join(HERE, 'shadow.hip')
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '-shared', '--offload-arch=gfx950', '-o', so, src], check=True)
    lib = C.CDLL(so)
    lib.shadow_last_error.restype = C.c_char_p
    lib.shadow_init.argtypes = [C.c_void_p, C.c_size_t]
    lib.shadow_configure.argtypes = [C.c_int, C.c_double, C.c_int, C.c_int]
    lib.shadow_allreduce.argtypes = [C.c_void_p, C.c_size_t]
    lib.shadow_stats.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=12)
    ap.add_argument('--warmup', type=int, default=4)
    ap.add_argument('--rounds', type=int, default=2)
    ap.add_argument('--channels', default='8,16,32,64')
    ap.add_argument('--busbw', default='0,150,300', help='GB/s the collectives are paced to (0 = unpaced)')
    ap.add_argument('--lds', default='0,65536', help='bytes of LDS the stand-in workgroups allocate')
    ap.add_argument('--ranks', type=int, default=8)
    ap.add_argument('--batch', type=int, default=256)
    a = ap.parse_args()

    import bench
    import np_modeling_amd as npm
    from np_modeling_amd import _C, device as D, parallel

    lib = shadow_lib()

    def chk(rc, what):
        if rc != 0:
            raise RuntimeError(f'{what}: {lib.shadow_last_error().decode()}')

    class Shadow(parallel.Communicator):
        rank, world_size = 0, a.ranks
        calls = 0

        def allreduce_async(self, flat, op):
            Shadow.calls += 1
            chk(lib.shadow_allreduce(flat.ptr, flat.size), 'shadow_allreduce')

        def wait(self):
            chk(lib.shadow_wait(), 'shadow_wait')

    seq, feat, heads, hidden = 512, 1024, 8, 4096
    rng = np.random.default_rng(0)
    params = bench.make_params(rng, feat, heads, hidden)
    qkv = D.from_host(rng.standard_normal([a.batch, seq, feat], dtype=np.float32))
    dy = D.from_host(rng.standard_normal([a.batch, seq, feat], dtype=np.float32) * np.float32(0.01))
    enc = npm.layers.TransformerEncoder(num_heads=heads, hidden_units=hidden, norm_first=True)
    enc(qkv)
    bench.bind(enc, params)
    sgd = npm.optimizer.SGDOptimizer(1e-4)
    chk(lib.shadow_init(_C.lib().npm_stream(), enc._numel() + 4096), 'shadow_init')

    def run(comm, steps):
        parallel.set_communicator(comm)
        for _ in range(a.warmup):
            enc(qkv)
            enc(dy, backprop=True, optimizer_=sgd)
        D.synchronize()
        lib.shadow_stats(None, None, None)
        Shadow.calls = 0
        marks = [D.Event() for _ in range(steps + 1)]
        marks[0].record()
        for i in range(steps):
            enc(qkv)
            enc(dy, backprop=True, optimizer_=sgd)
            marks[i + 1].record()
        D.synchronize()
        ms = sorted(marks[i].elapsed_ms(marks[i + 1]) for i in range(steps))
        coll, exposed, n = C.c_double(0), C.c_double(0), C.c_int(0)
        chk(lib.shadow_stats(C.byref(coll), C.byref(exposed), C.byref(n)), 'shadow_stats')
        parallel.set_communicator(None)
        return ms[len(ms) // 2], coll.value / steps, exposed.value / steps, Shadow.calls / steps

    configs = [('none', None)]
    for lds in (int(x) for x in a.lds.split(',')):
        for bw in (float(x) for x in a.busbw.split(',')):
            for ch in (int(x) for x in a.channels.split(',')):
                configs.append((f'channels {ch:3d}  busbw {bw:5.0f} GB/s  lds {lds // 1024:3d} KB', (ch, bw, lds)))
    results = {name: [] for name, _ in configs}
    bucket_mb = 4.0 * enc._arena.size / 1e6 if enc._arena is not None else float('nan')
    print(f'encoder step, batch {a.batch}: gradient bucket {bucket_mb:.1f} MB, {a.ranks}-rank ring stand-in; {a.steps} steps per entry, '
          f'{a.rounds} rounds, medians', flush=True)
    order = []
    for i, item in enumerate(configs[1:]):          # "none" again every sixth entry and at the end: the card's clock drifts within a round
        if i % 6 == 0:
            order.append(configs[0])
        order.append(item)
    order.append(configs[0])
    for r in range(a.rounds):
        for name, cfg in order:
            if cfg is None:
                results[name].append(run(None, a.steps))
            else:
                lib.shadow_configure(cfg[0], cfg[1], cfg[2], a.ranks)
                results[name].append(run(Shadow(), a.steps))
            print(f'  round {r + 1}: {name:52s} {results[name][-1][0]:8.3f} ms', flush=True)
    base = float(np.median([x[0] for x in results['none']]))
    print(f'\n{"configuration":52s} {"ms/step":>9s} {"vs none":>8s} {"collectives":>12s} {"stream busy":>12s} {"exposed":>9s}')
    worst = {}
    for name, cfg in configs:
        ms = float(np.median([x[0] for x in results[name]]))
        coll = float(np.median([x[1] for x in results[name]]))
        exposed = float(np.median([x[2] for x in results[name]]))
        calls = results[name][0][3]
        print(f'{name:52s} {ms:9.3f} {ms / base:8.4f} {calls:12.1f} {coll:9.3f} ms {exposed:6.3f} ms')
        if cfg is not None:
            worst[cfg[1]] = max(worst.get(cfg[1], 0.0), ms)
    print()
    for bw, ms in sorted(worst.items()):
        label = 'unpaced' if bw == 0 else f'paced to {bw:.0f} GB/s'
        print(f'weak-scaling efficiency bound from this experiment, worst configuration {label}: {base / ms:.4f}  ({base:.3f} / {ms:.3f} ms)')


if __name__ == '__main__':
    main()
