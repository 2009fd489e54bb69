#!/usr/bin/env python3
"""Headline benchmark: fwd+bwd samples/s of TransformerEncoder d_model=1024, 8 heads, seq=512,
batch 256 per GPU (BASELINE.json configs[4]; U = 4096 hidden units, pre-norm).

    python bench.py --gpus N --steps K --warmup W

N = 1 runs in this process.  For N > 1 every rank is its own process on its own GPU: either an
external launcher starts them (the driver: ``python -m torch.distributed.run --nproc-per-node N
... bench.py --gpus N ...``; RANK / LOCAL_RANK / WORLD_SIZE in the environment), or plain
``python bench.py --gpus N`` starts N fresh child processes itself (np_modeling_amd/launch.py)
before anything touches a GPU.  The ranks join an RCCL group (batch sharded: every rank runs the
per-GPU batch, weak scaling; the parameter gradients are all-reduced over xGMI inside
``backward``); the 128-byte RCCL id travels through a file -- the product imports no torch.

A step = forward + backward + the SGD update of every parameter, on synthetic fp32 tensors
already resident in HBM.  Rank 0 prints ONE JSON line (contract in the task description) with
two extra objects: ``roofline`` (the fp32-MFMA GEMM family, algorithmic FLOPs / HIP-event time
measured over the timed region) and ``cpu_baseline`` (the NumPy oracle on a bounded sample).
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2516.6     # same guide: v_mfma_f32_32x32x16_bf16, dense (16x the f32 MFMA rate)
HBM_PEAK_GBS = 8000.0
MATH_NOTE = {     # what a non-default --math does to the matrix products (include/npm_hip.h NPM_MATH_*)
    'bf16x3': 'products as six v_mfma_f32_32x32x16_bf16 of three-way bf16-split operands',
    'bf16x3_fast': 'products as six v_mfma_f32_32x32x16_bf16 of three-way bf16-split operands, one accumulator',
    'f16x2': 'products as three v_mfma_f32_32x32x16_f16 of row-scaled two-way fp16-split operands (the bf16 split for batched products)',
}
MFMA_BOUND = ('sgemm_', 'mha_core_')    # kernel-timer names of the MFMA-bound family: GEMMs and the fused attention core


def encoder_flops_per_sample(seq, feat, heads, hidden):
    """2*M*N*K per GEMM, forward + both gradients: 12 projection GEMMs,
    6 attention GEMMs, 6 feed-forward GEMMs per step."""
    proj = 12 * 2 * seq * feat * feat
    attn = 6 * 2 * seq * seq * feat            # heads * (S*S*Dk) = S*S*F
    ffn = 6 * 2 * seq * feat * hidden
    return float(proj + attn + ffn)


def make_params(rng, feat, heads, hidden):
    dk = feat // heads
    s_f, s_h = 1.0 / np.sqrt(feat), 1.0 / np.sqrt(hidden)

    def init(shape, scale=1.0):      # the reference's initializer (N(0,1) clipped), scaled by 1/sqrt(fan_in)
        return (np.clip(rng.standard_normal(shape, dtype=np.float32), -1, 1) * scale).astype(np.float32)

    return dict(
        att_wq=init([heads, dk, feat], s_f), att_wk=init([heads, dk, feat], s_f),
        att_wv=init([heads, dk, feat], s_f), att_wo=init([feat, heads, dk], s_f),
        att_bq=init([heads, dk]), att_bk=init([heads, dk]), att_bv=init([heads, dk]), att_bo=init([feat]),
        n1_gamma=init([feat]), n1_beta=init([feat]), n2_gamma=init([feat]), n2_beta=init([feat]),
        d1_w=init([feat, hidden], s_f), d1_b=init([hidden]), d2_w=init([hidden, feat], s_h), d2_b=init([feat]))


def bind(enc, p):
    """Overwrite the lazily initialised parameters IN PLACE (same device storage, same layout)."""
    att = enc._self_attention
    for n in ('wq', 'wk', 'wv', 'wo', 'bq', 'bk', 'bv', 'bo'):
        getattr(att, '_' + n).set(p['att_' + n])
    enc._norm1._gamma.set(p['n1_gamma'])
    enc._norm1._beta.set(p['n1_beta'])
    enc._norm2._gamma.set(p['n2_gamma'])
    enc._norm2._beta.set(p['n2_beta'])
    enc._dense1._linear._w.set(p['d1_w'])
    enc._dense1._linear._b.set(p['d1_b'])
    enc._dense2._w.set(p['d2_w'])
    enc._dense2._b.set(p['d2_b'])


def cpu_baseline(args, params):
    """NumPy oracle (closed-form flavour, BLAS GEMMs) on a bounded sample of the same workload."""
    from oracle import np_oracle as O
    b = args.cpu_batch
    rng = np.random.default_rng(1)
    qkv = rng.standard_normal([b, args.seq, args.features], dtype=np.float32)
    dy = rng.standard_normal([b, args.seq, args.features], dtype=np.float32) * np.float32(0.01)
    p = {k: v.copy() for k, v in params.items()}
    times = []
    for _ in range(args.cpu_steps):
        t0 = time.perf_counter()
        _, cache = O.encoder_fwd(p, qkv, True)
        _, grads = O.encoder_bwd(p, cache, dy, True)
        for k in p:
            p[k] = O.sgd_step(p[k], grads[k], 1e-4)
        times.append(time.perf_counter() - t0)
    best = min(times)
    out = dict(value=b / best, unit='samples/s', cores=os.cpu_count(), kind='port',
               sample=f'NumPy oracle (closed-form restatement, OpenBLAS) encoder fwd+bwd+SGD, batch {b} x seq {args.seq} '
                      f'x d {args.features}, U={args.hidden}, best of {args.cpu_steps} steps ({best:.2f} s/step)')
    if args.cpu_verbatim_seq > 0:
        # Line (1) of BASELINE.md section 3: the reference's OWN formulation (explicit [rows, d, d] LayerNorm and
        # [rows, n, n] softmax Jacobians in fp64, np.einsum contractions: single-threaded C loops, no BLAS), restated
        # verbatim by the oracle.  Its temporaries grow with seq^2 and d^2 (TBs at the full shape), so the sample is
        # ONE sequence of `cpu_verbatim_seq` tokens at the full width; the rate is quoted in tokens/s and, divided by
        # seq, as the samples/s a full-length sample could at best reach.
        sv = args.cpu_verbatim_seq
        qv = rng.standard_normal([1, sv, args.features], dtype=np.float32)
        dv = rng.standard_normal([1, sv, args.features], dtype=np.float32) * np.float32(0.01)
        t0 = time.perf_counter()
        _, cache = O.encoder_fwd(params, qv, True, verbatim=True)
        _, grads = O.encoder_bwd(params, cache, dv, True, verbatim=True)
        for k in params:
            O.sgd_step(params[k], grads[k], 1e-4)
        tv = time.perf_counter() - t0
        out['reference_verbatim'] = dict(
            value=sv / tv / args.seq, unit='samples/s', tokens_per_s=sv / tv, cores=1, kind='port',
            sample=f'reference-verbatim formulation (Jacobian einsums, fp64 temporaries) encoder fwd+bwd+SGD, 1 x seq {sv} '
                   f'x d {args.features}, U={args.hidden}, one step ({tv:.1f} s); samples/s = tokens/s / {args.seq} (an upper '
                   'bound: attention and the softmax Jacobian grow faster than linearly in seq)')
    return out


ALT_MATH_LINES = (('alt_math', 'bf16x3'), ('alt_math_f16x2', 'f16x2'))


def reported_math_modes():
    """The arithmetics this file reports a throughput line for: the headline's and one per ALT_MATH_LINES entry -- only modes
    with a parity bound in include/npm_hip.h (NPM_PARITY_*), which tests/test_gpu_parity.py asserts on every BASELINE config."""
    from np_modeling_amd import _C
    bounds = _C.parity_bounds()
    return [m for m in ['f32'] + [mode for _, mode in ALT_MATH_LINES] if m in bounds]


def parity_note(mode):
    from np_modeling_amd import _C
    rel, scaled = _C.parity_bounds()[mode]
    return {'rel_bound': rel, 'scaled_bound': scaled, 'source': f'include/npm_hip.h NPM_PARITY_*_{mode.upper()}',
            'asserted_by': f'tests/test_gpu_parity.py::test_baseline_configs_meet_the_stated_bound[{mode}] (C1-C5 at full width vs the fp64 oracle)'}


def alt_roofline(timer, per_product=6, kernel=None):
    """Kernel-level roofline of a split-precision GEMM family against the 16-bit MFMA peak: every fp32 product is
    executed as `per_product` v_mfma_f32_32x32x16 (bf16 split: 6, scaled fp16 split: 3), so the pipe executes that
    many times the algorithmic fp32 FLOPs."""
    if timer is None:
        return None
    gemm = {k: v for k, v in timer.summary().items() if k.startswith(MFMA_BOUND)}
    ms, flops = sum(v['ms'] for v in gemm.values()), sum(v['flops'] for v in gemm.values())
    launches = sum(v['launches'] for v in gemm.values())
    if ms <= 0:
        return None
    executed = per_product * flops / (ms * 1e-3) / 1e12
    return {'kernel': kernel or 'sgemm_glds_kernel<..., MATH = 2> (six v_mfma_f32_32x32x16_bf16 per fp32 product; attention composed from these GEMMs)',
            'bound': 'mfma', 'achieved': executed, 'peak': BF16_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s (16-bit MFMA work executed)',
            'frac': executed / BF16_MFMA_PEAK_TFLOPS, 'fp32_equivalent_tflops': flops / (ms * 1e-3) / 1e12,
            'launches': launches, 'avg_launch_ms': ms / max(launches, 1), 'traffic': None}


def load_pmc_traffic(args):
    """Bytes per GEMM launch that leave the L2s, from the rocprofv3 PMC passes of THIS build (profiles/summarize_pmc.py): PMC
    counters cannot be read from inside the process, so the figure is a property of a profiling session -- the JSON names it
    (``session``, ``collected``) and the sources its library was built from (``source_id``).  A file collected for other
    sources, or a library older than its sources, gives null: numbers of another build are not this run's."""
    from np_modeling_amd import _C
    path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    default = (args.batch, args.seq, args.features, args.heads, args.hidden) == (256, 512, 1024, 8, 4096)
    if not default:
        return None, 'not the default workload: no PMC session exists for this shape'
    if not os.path.exists(path):
        return None, 'profiles/pmc_traffic.json is missing (tools/refresh_profiles.sh collects it)'
    with open(path) as f:
        data = json.load(f)
    now = _C.source_id()
    if data.get('source_id') != now:
        return None, (f'profiles/pmc_traffic.json was collected for sources {data.get("source_id", "(unrecorded: before round 5)")} in session '
                      f'{data.get("session", "?")}; this build is {now}: stale, rerun tools/refresh_profiles.sh')
    if not _C.library_is_current():
        return None, 'libnpm_hip.so is older than its sources: rebuild before quoting a profile for them'
    return data['gemm_family_bytes_per_launch'], (
        f'profiles/pmc_traffic.json, session {data.get("session")} ({data.get("collected")}), sources {now} = this build: '
        '(2*FETCH_SIZE + WRITE_SIZE)*1024, separate rocprofv3 --pmc passes: requests that leave the L2s (fabric side), Infinity-Cache '
        'hits included.  An HBM-side figure cannot be taken on this pool: rocprofv3 exposes no memory-controller / Infinity-Cache '
        'counters on gfx950 and the SMU mem_busy figure reads 0 (profiles/r03_hbm_side.log, profiles/r03_pmc_tcc_ffn_dw.log)')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=256, help='per-GPU batch')
    ap.add_argument('--seq', type=int, default=512)
    ap.add_argument('--features', type=int, default=1024)
    ap.add_argument('--heads', type=int, default=8)
    ap.add_argument('--hidden', type=int, default=4096)
    ap.add_argument('--cpu-batch', type=int, default=8)
    ap.add_argument('--cpu-steps', type=int, default=3)
    ap.add_argument('--cpu-verbatim-seq', type=int, default=256, help='tokens of the reference-verbatim CPU sample (0 = skip)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-configs', action='store_true',
                    help="skip the 'configs' object (BASELINE.json configs[1..3] = C2 Dense, C3 Conv2D, C4 MHA at full size, N = 1 only)")
    ap.add_argument('--configs-min-seconds', type=float, default=0.5, help='timed seconds per config (tools/config_bench.py rule)')
    ap.add_argument('--no-kernel-timer', action='store_true')
    ap.add_argument('--math', default='f32', choices=['f32', 'bf16x3', 'bf16x3_fast', 'f16x2'],
                    help="arithmetic of the matrix products for the headline measurement (include/npm_hip.h npm_set_math)")
    ap.add_argument('--no-alt-math', action='store_true',
                    help="skip the second timed region that repeats the K steps with --math bf16x3 ('alt_math' in the JSON)")
    args = ap.parse_args()

    if args.gpus > 1 and 'RANK' not in os.environ:
        # Plain `python bench.py --gpus N`: this process becomes the launcher.  It has not touched the GPU (nothing
        # above imports the device library) and never will: the N ranks are FRESH child processes.
        from np_modeling_amd import launch
        sys.exit(launch.spawn_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    args.gpus = world                                    # under a launcher its environment is authoritative

    import np_modeling_amd as npm
    from np_modeling_amd import device as D, parallel

    comm = parallel.init('avg')
    npm.set_math(args.math)
    rng = np.random.default_rng(0)                       # identical parameters on every rank
    params = make_params(rng, args.features, args.heads, args.hidden)
    data_rng = np.random.default_rng(1000 + rank)        # per-rank shard of the global batch
    shape = [args.batch, args.seq, args.features]
    qkv = D.from_host(data_rng.standard_normal(shape, dtype=np.float32))
    dy = D.from_host(data_rng.standard_normal(shape, dtype=np.float32) * np.float32(0.01))

    enc = npm.layers.TransformerEncoder(num_heads=args.heads, hidden_units=args.hidden, norm_first=True)
    # Lazy initialisation happens at the first forward (reference layer.py:33-35).  With warm-up
    # steps the first of them does it at full size, so that EVERY launch a profiler sees belongs to
    # an identical step; without warm-up a tiny probe input does it.
    if args.warmup > 0:
        enc(qkv)
    else:
        enc(D.from_host(np.zeros([1, 8, args.features], dtype=np.float32)))
    bind(enc, params)
    sgd = npm.optimizer.SGDOptimizer(1e-4)

    def step():
        enc(qkv)
        enc(dy, backprop=True, optimizer_=sgd)

    for i in range(args.warmup):
        if i > 0:
            enc(qkv)                                     # (step 0's forward ran above, before bind)
        enc(dy, backprop=True, optimizer_=sgd)
    D.synchronize()
    if comm.active:
        comm.barrier()

    rccl = isinstance(comm, parallel.RcclCommunicator)
    if rccl:
        comm.stats_enable(True)
        comm.stats()                                     # drop what the warm-up steps recorded
    timer = None if args.no_kernel_timer else D.KernelTimer()
    if timer is not None:
        timer.__enter__()
    marks = [D.Event() for _ in range(args.steps + 1)]   # one HIP event per step boundary, on the compute stream
    marks[0].record()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step()
        marks[i + 1].record()
    D.synchronize()
    local_elapsed = time.perf_counter() - t0             # this rank alone, before it waits for the others
    if comm.active:
        comm.barrier()
    elapsed = time.perf_counter() - t0
    if timer is not None:
        timer.__exit__(None, None, None)
    step_order = [marks[i].elapsed_ms(marks[i + 1]) for i in range(args.steps)]
    step_ms = sorted(step_order)
    exchange_stats = comm.stats() if rccl else None
    if rccl:
        comm.stats_enable(False)
    rank_ms = None
    if comm.active:
        elapsed = comm.allreduce_scalar(elapsed, parallel.MAX)
        rank_ms = {'max': 1e3 * comm.allreduce_scalar(local_elapsed, parallel.MAX) / max(args.steps, 1),
                   'min': -1e3 * comm.allreduce_scalar(-local_elapsed, parallel.MAX) / max(args.steps, 1)}

    total_samples = args.batch * world * args.steps
    value = total_samples / elapsed
    fps = encoder_flops_per_sample(args.seq, args.features, args.heads, args.hidden)

    result = {
        'metric': 'fwd+bwd samples/sec, TransformerEncoder d=1024 seq=512',
        'value': value, 'unit': 'samples/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None,
        # `value` / `ms_per_step` are the contract's wall-clock figures (barrier + synchronize on both sides, max over
        # ranks, mean over the K steps).  Beside them, rank 0's per-step DEVICE times (HIP events on the compute stream
        # at every step boundary): the median is what a step costs once the clocks have settled.
        'step_ms_device': ({'median': step_ms[len(step_ms) // 2] if len(step_ms) % 2 else 0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2]),
                            'min': step_ms[0], 'max': step_ms[-1], 'mean': sum(step_ms) / len(step_ms),
                           'first': step_order[0], 'slowest_step': step_order.index(step_ms[-1]),
                            'note': 'HIP events at step boundaries, rank 0; the kernel timer\'s own events are inside these intervals'}
                           if step_ms else None),
        'dtype': 'f32' if args.math == 'f32' else f'f32 ({args.math}: {MATH_NOTE[args.math]}, fp32 accumulate)',
        'math': args.math, 'data': 'synthetic N(0,1) fp32 inputs resident in HBM; random-init '
        'weights (reference initializer scaled by 1/sqrt(fan_in))',
        'config': {'workload': f'TransformerEncoder fwd+bwd+SGD step, d_model={args.features}, heads={args.heads}, '
                               f'seq={args.seq}, hidden_units={args.hidden}, pre-norm, batch {args.batch}/GPU '
                               f'(BASELINE.json configs[4] per-GPU shard; global batch {args.batch * world})',
                   'global_batch': args.batch * world, 'seq_len': args.seq,
                   'parity': 'north_star "within 1e-4 rel" as asserted by tests/test_gpu_parity.py against the fp64 NumPy oracle: scaled error '
                             'max|got-ref|/max|ref| <= 1e-5 over all elements and elementwise relative error <= 1e-4 on elements >= 0.1 max|ref|',
                   'parallelism': f'dp{world} (batch-sharded, RCCL grad all-reduce)' if world > 1 else 'single GPU'},
        'step_tflops_per_gpu': value / world * fps / 1e12,
        'step_frac_of_fp32_mfma_peak': value / world * fps / 1e12 / FP32_MFMA_PEAK_TFLOPS,
    }
    if comm.active:
        result['exchange'] = {'library': parallel.RcclCommunicator.library_path(), 'reduce': 'avg',
                              'launcher': 'external (RANK/WORLD_SIZE in the environment)' if 'NPM_RENDEZVOUS_FILE' not in os.environ
                              else 'np_modeling_amd.launch (self-launched child ranks)', 'torch_imported': 'torch' in sys.modules,
                              'device': npm._C.device_index(), 'visible_devices': npm._C.visible_devices()}
        result['exchange']['cpu_placement'] = parallel.PLACEMENT
        if rank_ms is not None:
            # every rank's own wall clock over the K steps (before the closing barrier): the spread is what the slowest
            # rank costs the others in a weak-scaling run
            result['exchange']['ms_per_step_over_ranks'] = rank_ms
        if exchange_stats is not None:
            # rank 0's view, HIP events (include/npm_comm.h npm_comm_stats): allreduce_ms = time the collectives held the
            # communication stream (overlaps backward); exposed_ms = time the compute stream stood still waiting for them
            k = max(args.steps, 1)
            result['exchange'].update({
                'bytes_per_step': exchange_stats['bytes'] / k, 'flushes_per_step': exchange_stats['allreduce_calls'] / k,
                'allreduce_ms': exchange_stats['allreduce_ms'] / k, 'exposed_ms': exchange_stats['exposed_ms'] / k,
                'exposed_frac_of_step': exchange_stats['exposed_ms'] / (1e3 * elapsed) if elapsed > 0 else None,
                # the flush issued after the LAST gradient of a backward: nothing is left to overlap it
                'last_flush_ms': exchange_stats['last_allreduce_ms'] / k,
                'busbw_GBps': (2 * (world - 1) / world * exchange_stats['bytes'] / (exchange_stats['allreduce_ms'] * 1e-3) / 1e9
                               if world > 1 and exchange_stats['allreduce_ms'] > 0 else None),
                'note': 'per step, rank 0; allreduce_ms overlaps backward, exposed_ms is what the compute stream waited, '
                        'last_flush_ms is the duration of the final collective of each backward (included in allreduce_ms)'})

    if timer is not None:
        summary = timer.summary()
        gemm = {k: v for k, v in summary.items() if k.startswith('sgemm_')}             # the dominant kernel
        attn = {k: v for k, v in summary.items() if k.startswith('mha_core_')}
        g_ms = sum(v['ms'] for v in gemm.values())
        g_flops = sum(v['flops'] for v in gemm.values())
        g_launches = sum(v['launches'] for v in gemm.values())
        achieved = g_flops / (g_ms * 1e-3) / 1e12 if g_ms > 0 else 0.0
        traffic, traffic_src = load_pmc_traffic(args)
        result['roofline'] = {
            'kernel': ('sgemm_glds_kernel (fp32 v_mfma_f32_32x32x2_f32 GEMM family, LDS-DMA pipeline: NN/NT/TN)' if args.math == 'f32' else
                       f'{"sgemm_f16x2_kernel" if args.math == "f16x2" else "sgemm_glds_kernel"} ({args.math}: {MATH_NOTE[args.math]}; achieved '
                       'counts fp32-equivalent FLOPs against the f32 MFMA peak, the 16-bit pipe executes 3-6x that)'),
            'bound': 'mfma', 'achieved': achieved, 'peak': FP32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
            'frac': achieved / FP32_MFMA_PEAK_TFLOPS, 'traffic': traffic, 'traffic_unit': 'bytes per launch',
            'traffic_source': traffic_src,
            'algorithmic_bytes_per_launch': sum(v['bytes'] for v in gemm.values()) / max(g_launches, 1),
            'launches': g_launches, 'avg_launch_ms': g_ms / max(g_launches, 1),
            'share_of_step_time': g_ms / (1e3 * elapsed) if elapsed > 0 else None,
            'by_layout': {k: {'launches': v['launches'], 'avg_ms': v['ms'] / v['launches'],
                              'tflops': v['flops'] / (v['ms'] * 1e-3) / 1e12} for k, v in sorted(gemm.items())},
            'other_mfma_kernels': {k: {'launches': v['launches'], 'avg_ms': v['ms'] / v['launches'],
                                       'tflops': v['flops'] / (v['ms'] * 1e-3) / 1e12,
                                       'frac': v['flops'] / (v['ms'] * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                                       'share_of_step_time': v['ms'] / (1e3 * elapsed)}
                                   for k, v in sorted(attn.items())},       # fused attention core: 2 + 4 algorithmic products
            'hbm_kernels': {k: {'launches': v['launches'], 'avg_ms': v['ms'] / v['launches'],
                                'GBps': v['bytes'] / (v['ms'] * 1e-3) / 1e9, 'frac_of_8TBps': v['bytes'] / (v['ms'] * 1e-3) / 1e9 / HBM_PEAK_GBS}
                            for k, v in sorted(summary.items()) if not k.startswith(MFMA_BOUND) and v['ms'] > 0},
        }
        if 'mha_core_bwd' in result['roofline']['other_mfma_kernels']:
            rowdot = D.ATTN_ROWDOT and args.math == 'f32' and args.features // args.heads == 128
            result['roofline']['other_mfma_kernels']['mha_core_bwd']['row_terms'] = (
                'delta = dctx . ctx per query and head comes out of the epilogue of the dctx = dy wo GEMM (NPM_EPI_ROWDOT: one of the four '
                'sgemm_NN launches of a step carries it, its cost is inside by_layout.sgemm_NN); this entry is the attention backward '
                'kernel plus a 4 MB log2(e)*LSE pass.  Rounds 3-4 ran a pass over dctx and ctx in front of the kernel instead '
                '(0.21 of their 4.45 ms; NPM_ATTN_ROWDOT=0 restores it)' if rowdot else
                'delta = dctx . ctx by a pass over dctx and ctx in front of the kernel (mha_rowterms_kernel), included in this entry')
    def alt_region(mode):
        """A further timed region, same K steps, same barriers, under another arithmetic of the matrix products."""
        npm.set_math(mode)
        step()
        D.synchronize()
        if comm.active:
            comm.barrier()
        alt_timer = None if args.no_kernel_timer else D.KernelTimer()
        if alt_timer is not None:
            alt_timer.__enter__()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        D.synchronize()
        if comm.active:
            comm.barrier()
        alt = time.perf_counter() - t0
        if alt_timer is not None:
            alt_timer.__exit__(None, None, None)
        if comm.active:
            alt = comm.allreduce_scalar(alt, parallel.MAX)
        npm.set_math('f32')
        alt_value = total_samples / alt
        return alt_timer, {'math': mode, 'value': alt_value, 'unit': 'samples/s', 'ms_per_step': 1e3 * alt / args.steps,
                           'steps': args.steps, 'step_tflops_per_gpu': alt_value / world * fps / 1e12,
                           'step_frac_of_fp32_mfma_peak': alt_value / world * fps / 1e12 / FP32_MFMA_PEAK_TFLOPS}

    result['parity'] = parity_note(args.math) if args.math in reported_math_modes() else None
    if args.math == 'f32' and not args.no_alt_math and 'bf16x3' in reported_math_modes():
        # the matrix products on the bf16 pipe (three-way operand split, fp32-class error -- DESIGN.md 4.1,
        # tests/test_gpu_gemm.py::test_split_math_error_statistics)
        alt_timer, obj = alt_region('bf16x3')
        obj.update({
            'executed_bf16_mfma_frac_of_bf16_peak': 6 * obj['value'] / world * fps / 1e12 / BF16_MFMA_PEAK_TFLOPS,
            'roofline': alt_roofline(alt_timer),
            'note': 'same workload and timing protocol with npm_set_math(NPM_MATH_BF16X3): fp32 inputs/outputs/accumulators, '
                    'each product formed from three-way bf16 splits of both operands (six v_mfma_f32_32x32x16_bf16); '
                    'rms error vs fp64 at or below the exact-f32 MFMA path (profiles/r02_math_error.log). '
                    'Not the headline: value above is the exact-f32 MFMA path.'})
        obj['parity'] = parity_note('bf16x3')
        result['alt_math'] = obj
    if args.math == 'f32' and not args.no_alt_math and 'f16x2' in reported_math_modes():
        # ... and on the f16 pipe: two-way fp16 split with row scaling, three MFMAs per product (csrc/npm_gemm_f16x2.hip)
        alt_timer, obj = alt_region('f16x2')
        obj.update({
            'roofline': alt_roofline(alt_timer, per_product=3, kernel='sgemm_f16x2_kernel (three v_mfma_f32_32x32x16_f16 per fp32 '
                                     'product, operand maxima passes included in the launch time; attention composed from the '
                                     'batched bf16-split GEMMs, six MFMAs per product there)'),
            'note': 'same workload and timing protocol with npm_set_math(NPM_MATH_F16X2): every row of op(A) / column of op(B) '
                    'scaled by a power of two, split into two fp16 parts, (hi hi + hi lo + lo hi) / (s_a s_b); error against '
                    'fp64 below a k-ordered fp32 fma chain ROW-NORMWISE (profiles/r02_f16x2_gemm.log), not elementwise. '
                    'Not the headline: value above is the exact-f32 MFMA path.'})
        obj['parity'] = parity_note('f16x2')
        result['alt_math_f16x2'] = obj
    if world == 1 and not args.no_configs and args.math == 'f32':
        # The other single-GPU configurations of BASELINE.json at full size, same process, after the headline regions
        # (tools/config_bench.py: >= configs_min_seconds timed after half of that as warm-up, fwd+bwd+SGD, exact-f32).
        sys.path.insert(0, os.path.join(ROOT, 'tools'))
        import config_bench
        del enc, qkv, dy
        D.trim_pool()
        result['configs'] = {}
        for name in config_bench.NAMES:
            entry = config_bench.run_config(name, npm, D, min_seconds=args.configs_min_seconds)
            D.trim_pool()
            if not args.no_cpu_baseline:
                entry['cpu_baseline'] = config_bench.cpu_baseline(name)
            result['configs'][name] = entry
        # two more lines beside the BASELINE configs (no CPU baseline; not BASELINE.json's): C4 under a causal mask -- the mask's tile
        # summary lets the fused kernels skip empty tiles -- and C4's dimensions with 16 heads of 64 (the 8-wave backward at head size 64)
        result['configs_extra'] = {}
        for name in ('C4M', 'C4D64', 'C5D'):     # C5D: the headline workload with drop_rate 0.1 (dropout inside the LayerNorm kernels)
            result['configs_extra'][name] = config_bench.run_config(name, npm, D, min_seconds=args.configs_min_seconds)
            D.trim_pool()
    if comm.active:
        comm.barrier()                                   # every timed region of every rank is over
    D.synchronize()
    parallel.shutdown()                                  # all ranks leave the group HERE: no collective (and no busy-waiting
    #                                                      barrier kernel) is outstanding while rank 0 samples the CPU below
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result['cpu_baseline'] = cpu_baseline(args, params)      # rank 0's host cores, after every timed region; N = 1 only
    elif rank == 0:
        result['cpu_baseline'] = None                            # (N > 1 runs stay short: the baseline is on the N = 1 line)
    if rank == 0:
        print(json.dumps(result))


if __name__ == '__main__':
    main()
