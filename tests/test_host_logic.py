"""CPU: host-side logic of the product on the host simulator of the C ABI (tests/hostsim.py):
Layer protocol, weight layouts / GEMM descriptors against the reference's golden outputs,
gradient bucketing, and the data-parallel exchange over gloo with world_size 2.

The kernels themselves are validated on the GPU (tests marked ``gpu``); here every kernel is
the simulator's NumPy restatement, so a failure points at the Python host layer."""

import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import hostsim
from conftest import assert_close, load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture
def npm():
    import np_modeling_amd
    from np_modeling_amd import parallel
    hostsim.install()
    parallel.set_communicator(None)
    yield np_modeling_amd
    parallel.set_communicator(None)
    hostsim.uninstall()


def rand(shape):
    return np.random.normal(size=shape).astype(np.float32)


def test_dense_golden_on_simulator(npm):
    g = load_golden('dense')
    np.random.seed(0)
    layer = npm.layers.Dense(units=16)
    x = rand([64, 32])
    y = layer(x)
    w, b = layer.linear.w, layer.linear.b
    np.testing.assert_array_equal(np.asarray(w), g['w0'])
    assert_close(y, g['y'], tol=1e-6)
    dx = layer(g['dy'], backprop=True, learning_rate=float(g['lr']))
    assert_close(dx, g['dx'], tol=1e-6)
    assert_close(w, g['w1'], tol=1e-6)
    assert_close(b, g['b1'], tol=1e-6)


@pytest.mark.parametrize('name', ['mha_self', 'mha_cross'])
def test_mha_descriptors_on_simulator(npm, name):
    """Head-strided operand descriptors and weight layouts of attentions.py against the reference."""
    g = load_golden(name)
    layer = npm.layers.MultiHeadAttention(num_heads=int(g['heads']))
    kv = g.get('kv')
    args = (g['query'],) if kv is None else (g['query'], kv)
    layer(*args)
    names = ['wq', 'wk', 'wv', 'wo', 'bq', 'bk', 'bv', 'bo']
    for n in names:
        setattr(layer, '_' + n, g[n + '0'])
    assert_close(layer(*args), g['out'], tol=1e-5)
    dq, dk, dv = layer(g['dy'], backprop=True, learning_rate=float(g['lr']))
    assert_close(dq, g['dquery'], tol=1e-5)
    assert_close(dk, g['dkey'], tol=1e-5)
    assert_close(dv, g['dvalue'], tol=1e-5)
    for n in names:
        assert_close(getattr(layer, '_' + n), g[n + '1'], tol=1e-5, what=n)


@pytest.mark.parametrize('name', ['encoder_prenorm', 'encoder_postnorm'])
def test_encoder_composition_on_simulator(npm, name):
    g = load_golden(name)
    np.random.seed(0)
    enc = npm.layers.TransformerEncoder(num_heads=int(g['heads']), hidden_units=int(g['hidden']),
                                        norm_first=bool(g['norm_first']))
    out = enc(rand(g['qkv'].shape))
    assert_close(out, g['out'], tol=1e-5)
    dx = enc(g['dy'], backprop=True, learning_rate=float(g['lr']))
    assert_close(dx, g['dx'], tol=1e-5)
    assert_close(enc._dense2.w, g['d2_w__1'], tol=1e-5)
    assert_close(enc._self_attention._wq, g['att_wq__1'], tol=1e-5)
    assert_close(enc._norm1._gamma, g['n1_gamma__1'], tol=1e-5)


@pytest.mark.parametrize('name', ['encoder_dropout_prenorm', 'encoder_dropout_postnorm'])
def test_encoder_with_dropout_reproduces_the_reference_run(npm, name):
    """The reference's own seeded run with drop_rate = 0.1 (tests/golden/encoder_dropout_*.npz, written by oracle/make_golden.py
    from the real reference): the same seed draws the same parameters AND the same masks (np.random.binomial at its place in
    the lazy-initialisation order), and output, input gradient and updated parameters follow -- through the fused
    composition (dropout inside the LayerNorm kernels)."""
    g = load_golden(name)
    np.random.seed(0)
    enc = npm.layers.TransformerEncoder(num_heads=int(g['heads']), hidden_units=int(g['hidden']),
                                        norm_first=bool(g['norm_first']), drop_rate=float(g['drop_rate']))
    del npm._C._LIB.calls[:]
    out = enc(rand(g['qkv'].shape))
    assert enc._fused and npm._C._LIB.calls.count('npm_layernorm_dropout_fwd') == 2
    np.testing.assert_array_equal(np.asarray(enc._dropout1._mask), g['mask1'])
    np.testing.assert_array_equal(np.asarray(enc._dropout2._mask), g['mask2'])
    np.testing.assert_array_equal(np.asarray(enc._norm2._gamma), g['n2_gamma__0'])
    assert_close(out, g['out'], tol=1e-5)
    dx = enc(g['dy'], backprop=True, learning_rate=float(g['lr']))
    assert_close(dx, g['dx'], tol=1e-5)
    assert_close(enc._dense2.w, g['d2_w__1'], tol=1e-5)
    assert_close(enc._self_attention._wq, g['att_wq__1'], tol=1e-5)
    assert_close(enc._norm1._gamma, g['n1_gamma__1'], tol=1e-5)
    assert_close(enc._norm2._beta, g['n2_beta__1'], tol=1e-5)


@pytest.mark.parametrize('name', ['decoder_prenorm', 'decoder_postnorm'])
def test_decoder_fused_composition_on_simulator(npm, name):
    """The fused decoder composition (seeded draw order, residual / sum epilogues, the (dq, dkv) tuple) against the
    reference's own outputs (reference transformer.py:95-203, transformer_test.py:159-219)."""
    g = load_golden(name)
    np.random.seed(0)
    dec = npm.layers.TransformerDecoder(num_heads=int(g['heads']), hidden_units=int(g['hidden']),
                                        norm_first=bool(g['norm_first']))
    out = dec(rand(g['q'].shape), rand(g['kv'].shape))
    assert dec._fusable()
    np.testing.assert_array_equal(np.asarray(dec._cross_attention._wk), g['ca_wk__0'])      # reference draw order
    assert_close(out, g['out'], tol=1e-4)
    dq, dkv = dec(g['dy'], backprop=True, learning_rate=float(g['lr']))
    assert_close(dq, g['dq'], tol=1e-4)
    assert_close(dkv, g['dkv'], tol=1e-4)
    assert_close(dec._dense2.w, g['d2_w__1'], tol=1e-4)
    assert_close(dec._cross_attention._wv, g['ca_wv__1'], tol=1e-4)
    assert_close(dec._norm3._gamma, g['n3_gamma__1'], tol=1e-4)


@pytest.mark.parametrize('name', ['decoder_dropout_prenorm', 'decoder_dropout_postnorm'])
def test_decoder_with_dropout_reproduces_the_reference_run(npm, name):
    """The reference's seeded decoder run with drop_rate = 0.1 (tests/golden/decoder_dropout_*.npz): same parameters, same three
    masks, then output, (dq, dkv) and updated parameters -- through the fused composition (npm_layernorm_dropout_* three times
    each way, no standalone mask pass)."""
    g = load_golden(name)
    np.random.seed(0)
    dec = npm.layers.TransformerDecoder(num_heads=int(g['heads']), hidden_units=int(g['hidden']),
                                        norm_first=bool(g['norm_first']), drop_rate=float(g['drop_rate']))
    del npm._C._LIB.calls[:]
    out = dec(rand(g['q'].shape), rand(g['kv'].shape))
    assert dec._fused and npm._C._LIB.calls.count('npm_layernorm_dropout_fwd') == 3 and 'npm_mask_scale' not in npm._C._LIB.calls
    for i, d in enumerate((dec._dropout1, dec._dropout2, dec._dropout3), start=1):
        np.testing.assert_array_equal(np.asarray(d._mask), g[f'mask{i}'])
    np.testing.assert_array_equal(np.asarray(dec._cross_attention._wk), g['ca_wk__0'])
    assert_close(out, g['out'], tol=1e-4)
    del npm._C._LIB.calls[:]
    dq, dkv = dec(g['dy'], backprop=True, learning_rate=float(g['lr']))
    assert npm._C._LIB.calls.count('npm_layernorm_dropout_bwd') == 3 and 'npm_mask_scale' not in npm._C._LIB.calls
    assert_close(dq, g['dq'], tol=1e-4)
    assert_close(dkv, g['dkv'], tol=1e-4)
    assert_close(dec._dense2.w, g['d2_w__1'], tol=1e-4)
    assert_close(dec._cross_attention._wv, g['ca_wv__1'], tol=1e-4)
    assert_close(dec._norm3._gamma, g['n3_gamma__1'], tol=1e-4)


def test_weights_rebound_between_forward_and_backward(npm):
    """The packed q/k/v projection must not assume the parameters are still adjacent in the backward:
    bench.py (and weight binders) rebind them after the first forward."""
    from oracle import np_oracle as O
    np.random.seed(0)
    layer = npm.layers.MultiHeadAttention(num_heads=2)
    x = rand([2, 6, 8])
    layer(x)
    assert layer._packed
    names = ['wq', 'wk', 'wv', 'wo', 'bq', 'bk', 'bv', 'bo']
    p = {n: np.asarray(getattr(layer, '_' + n)).copy() for n in names}
    _, cache = O.mha_fwd({k: v.astype(np.float64) for k, v in p.items()}, x.astype(np.float64))
    for n in names:                                  # rebind AFTER the forward: separate allocations
        setattr(layer, '_' + n, p[n].copy())
    dy = rand([2, 6, 8])
    dq, dk, dv = layer(dy, backprop=True, learning_rate=0.1)
    (wq_, wk_, wv_), grads = O.mha_bwd({k: v.astype(np.float64) for k, v in p.items()}, cache, dy.astype(np.float64))
    assert_close(np.asarray(dq) + np.asarray(dk) + np.asarray(dv), wq_ + wk_ + wv_, tol=1e-5)
    for n in names:
        assert_close(getattr(layer, '_' + n), p[n] - 0.1 * grads[n], tol=1e-5, what=n)


def test_conv_layer_on_simulator(npm):
    g = load_golden('conv_k3')
    np.random.seed(0)
    layer = npm.layers.Conv2D(channels=12, kernel_size=3)
    y = layer(rand([3, 9, 7, 8]))
    assert_close(y, g['y'], tol=1e-6)
    dx = layer(g['dy'], backprop=True, learning_rate=float(g['lr']))
    assert_close(dx, g['dx'], tol=1e-6)
    assert_close(layer.w, g['w1'], tol=1e-6)


def test_device_array_contract(npm):
    import copy
    D = npm.device
    a = D.from_host(np.arange(6, dtype=np.float32).reshape(2, 3))
    alias = a
    a -= 0.5 * D.from_host(np.ones((2, 3), dtype=np.float32))          # lr * grad stays symbolic -> one axpy
    assert a is alias
    np.testing.assert_array_equal(np.asarray(alias), np.arange(6).reshape(2, 3) - 0.5)
    a -= np.full((2, 3), 0.5)                                            # host operand (Adam's update)
    np.testing.assert_array_equal(np.asarray(a), np.arange(6).reshape(2, 3) - 1.0)
    b = copy.deepcopy(a)
    b += 1.0
    assert not np.array_equal(np.asarray(a), np.asarray(b))
    assert a.reshape(-1).shape == (6,) and a.reshape(3, -1).shape == (3, 2)
    with pytest.raises(ValueError):
        a.reshape(4, 2)
    assert np.asarray(a, dtype=np.float64).dtype == np.float64
    assert (np.ones((2, 3)) + a).shape == (2, 3)                         # ndarray op DeviceArray -> host math
    assert isinstance(2.0 * a, D.Scaled) and np.allclose(np.asarray(2.0 * a), 2 * np.asarray(a))
    v = a.flat_view(2, [2, 2])
    v -= D.from_host(np.ones((2, 2), dtype=np.float32))
    assert np.asarray(a).ravel()[2] == 0.0                               # views alias their base


def test_grad_scope_bucket_and_deferred_updates(npm):
    """world_size 2 with a recording transport: one flat bucket, 16-byte aligned slices, every
    element exchanged exactly once, updates applied only after the exchange."""
    from np_modeling_amd import parallel
    D = npm.device
    events = []

    class Recorder(parallel.Communicator):
        rank, world_size = 0, 2

        def allreduce_async(self, flat, op):
            events.append(('allreduce', flat.ptr, flat.size, op))

        def wait(self):
            events.append(('wait',))

    class Opt:
        def update(self, obj, attribute, gradient):
            events.append(('update', attribute))

    parallel.set_communicator(Recorder())
    with parallel.grad_scope(64) as scope:
        g1 = scope.take([3])
        g2 = scope.take([2, 5])
        assert (g2.ptr - g1.ptr) % 16 == 0 and g2.ptr - g1.ptr == 16
        scope.defer(Opt(), None, '_a', g1)
        scope.flush()
        with parallel.grad_scope(8) as inner:                 # nested scope delegates to the root bucket
            g3 = inner.take([4])
            inner.defer(Opt(), None, '_b', g3)
        g_big = scope.take([1000])                             # does not fit: exchanged on its own
        scope.defer(Opt(), None, '_c', g_big)
        assert [e[0] for e in events] == ['allreduce']         # nothing applied yet
    kinds = [e[0] for e in events]
    assert kinds == ['allreduce', 'allreduce', 'allreduce', 'wait', 'update', 'update', 'update']
    first, second, third = events[0], events[1], events[2]
    assert first[1] == g1.ptr and first[2] == 14                # [0, 3) pad to 4, [4, 14)
    assert second[1] == g1.ptr + 4 * 14 and second[2] == 6      # [14, 16) pad, [16, 20)
    assert third[2] == 1000
    assert all(e[3] == parallel.AVG for e in events[:3])
    # world_size 1: no bucket, no exchange, updates still deferred to scope exit
    events.clear()
    parallel.set_communicator(None)
    with parallel.grad_scope(64) as scope:
        scope.defer(Opt(), None, '_a', scope.take([3]))
        assert events == []
    assert events == [('update', '_a')]


def test_math_mode_api_on_simulator(npm):
    """set_math / get_math map the names of include/npm_hip.h NPM_MATH_* and reject anything else."""
    assert npm.get_math() == 'f32'
    for name in ('bf16x3', 'bf16x3_fast', 'f32'):
        npm.set_math(name)
        assert npm.get_math() == name
    with pytest.raises(ValueError):
        npm.set_math('bf16')
    text = open(os.path.join(ROOT, 'include', 'npm_hip.h')).read()
    import re
    from np_modeling_amd import _C
    enum = dict(re.findall(r'NPM_MATH_(\w+) = (\d+)', text))
    assert {k.lower(): int(v) for k, v in enum.items()} == _C.MATH_MODES


def test_shard_helper(npm):
    from np_modeling_amd import parallel

    class Two(parallel.Communicator):
        rank, world_size = 1, 2

    parallel.set_communicator(Two())
    x = np.arange(24).reshape(4, 6)
    np.testing.assert_array_equal(parallel.shard(x), x[2:])
    with pytest.raises(AssertionError):
        parallel.shard(np.zeros((3, 2)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


@pytest.mark.parametrize('world', [2, 4])
def test_data_parallel_equivalence_gloo(world):
    """Two and four processes, gloo: batch-sharded training of an MLP stack and of pre-/post-norm encoders
    ends with the same parameters as one process on the whole batch."""
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   OMP_NUM_THREADS='1', OPENBLAS_NUM_THREADS='1')
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', 'dp_worker.py')], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    assert all(p.returncode == 0 for p in procs), '\n'.join(outs)
    assert 'encoder_post' in outs[0] and 'decoder_post' in outs[0]
    assert 'collectives per backward' in outs[0]


def _clean_env(**extra):
    env = dict(os.environ, OMP_NUM_THREADS='1')
    env.update(NPM_NO_AUTOBUILD='1', **extra) if 'NPM_NO_AUTOBUILD' not in extra else env.update(extra)
    for key in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'LOCAL_WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'NPM_RENDEZVOUS_FILE', 'NPM_LAUNCH_TOKEN',
                'NPM_LAUNCHER_PID', 'TORCHELASTIC_RUN_ID', 'SLURM_JOB_ID'):
        env.pop(key, None)
    return env


def test_self_launcher_rendezvous_two_ranks(capfd):
    """np_modeling_amd/launch.py: two FRESH child ranks, the RCCL id travels from rank 0 to rank 1 through the
    launcher's private file, no torch anywhere in the product path."""
    from np_modeling_amd import launch
    code = launch.spawn_ranks(2, [sys.executable, os.path.join(ROOT, 'tests', 'uid_worker.py')], build=False,
                              env=_clean_env())
    out = capfd.readouterr()
    assert code == 0, out.err
    assert 'rank 0/2: id ok' in out.out and 'rank 1/2: id ok' in out.err      # rank 0 owns stdout, the rest stderr


def test_self_launcher_reports_a_failing_rank(capfd):
    """A rank that exits non-zero ends the job with its code; the ranks still running are terminated."""
    import time
    from np_modeling_amd import launch
    t0 = time.monotonic()
    code = launch.spawn_ranks(3, [sys.executable, os.path.join(ROOT, 'tests', 'uid_worker.py')], build=False,
                              env=_clean_env(UID_WORKER_FAIL_RANK='1', UID_WORKER_HANG='1'))
    assert code == 7
    assert time.monotonic() - t0 < 60                 # did not wait for the ranks that sleep 120 s
    assert 'rank 1 exited with code 7' in capfd.readouterr().err


def test_launcher_module_cli():
    out = subprocess.run([sys.executable, '-m', 'np_modeling_amd.launch', '--gpus', '2',
                          os.path.join(ROOT, 'tests', 'uid_worker.py')], env=_clean_env(), cwd=ROOT,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240)
    assert out.returncode == 0, out.stderr
    assert out.stdout.strip() == 'rank 0/2: id ok'


def test_unique_id_exchange_under_torch_launcher():
    """The same rendezvous under the launcher the driver uses for N > 1 (python -m torch.distributed.run): the
    ranks are children of one agent process, which is what the derived rendezvous path keys on."""
    port = _free_port()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.join(ROOT, 'tests', 'uid_worker.py')]
    out = subprocess.run(cmd, env=_clean_env(), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=240)
    assert out.returncode == 0, out.stdout
    assert 'rank 0/2: id ok' in out.stdout and 'rank 1/2: id ok' in out.stdout, out.stdout


def test_bench_gpus_n_becomes_a_launcher_before_touching_the_gpu():
    """`python bench.py --gpus 2` as a plain command starts child ranks (reference: none -- SURVEY 8e is new
    functionality).  Without a GPU the ranks fail loudly in npm_init and the launcher reports that code; what
    is checked here is the process structure: two ranks were started with RANK / WORLD_SIZE set."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0',
                          '--no-cpu-baseline'], env=_clean_env(NPM_NO_AUTOBUILD='0'), cwd=ROOT,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    import ctypes
    from np_modeling_amd import _C
    count = ctypes.c_int(0)
    _C.load_library().npm_device_count(ctypes.byref(count))
    if count.value >= 2:
        assert out.returncode == 0, out.stderr[-3000:]
        assert '"n_gpus": 2' in out.stdout
    else:
        assert out.returncode != 0
        assert 'np_modeling_amd.launch: rank' in out.stderr and 'no CPU fallback' in out.stderr, out.stderr[-3000:]


def test_deepcopy_after_packed_forward(npm):
    """reference layers/attentions_test.py:72 deep-copies the layer between forward and backward.  With the packed
    q/k/v projection the cached k and v are views INTO the packed buffer: the copy must keep everything the strided
    GEMMs address, and parameter views of one block must stay adjacent."""
    import copy
    np.random.seed(0)
    layer = npm.layers.MultiHeadAttention(num_heads=2)
    x, dy = rand([2, 6, 8]), rand([2, 6, 8])
    layer(x)                                          # default-initialised parameters: the packed path
    assert layer._packed
    twin = copy.deepcopy(layer)
    assert twin._params_adjacent() and twin._wq.ptr != layer._wq.ptr
    got = [np.asarray(g) for g in twin(dy, backprop=True, learning_rate=0.1)]
    want = [np.asarray(g) for g in layer(dy, backprop=True, learning_rate=0.1)]
    for a, b in zip(got, want):
        assert np.isfinite(a).all()
        np.testing.assert_array_equal(a, b)
    for n in ('_wq', '_wk', '_wv', '_wo', '_bq', '_bk', '_bv', '_bo'):
        np.testing.assert_array_equal(np.asarray(getattr(twin, n)), np.asarray(getattr(layer, n)))
    # and an encoder copied after its forward
    enc = npm.layers.TransformerEncoder(num_heads=2, hidden_units=12, norm_first=True)
    enc(x)
    twin = copy.deepcopy(enc)
    np.testing.assert_array_equal(np.asarray(twin(dy, backprop=True, learning_rate=0.1)),
                                  np.asarray(enc(dy, backprop=True, learning_rate=0.1)))


@pytest.mark.parametrize('norm_first', [True, False])
@pytest.mark.parametrize('rng_kind', ['host', 'device'])
def test_encoder_dropout_rides_inside_the_layernorm_kernels(npm, norm_first, rng_kind):
    """drop_rate > 0 (reference transformer.py:13,22-23,35-36,40-41,49-50,55-56): the encoder keeps its fused composition --
    each DropOut is applied inside the LayerNorm kernels behind it (npm_layernorm_dropout_fwd / _bwd), no standalone
    mask pass or add runs -- and equals the literal composition (DropOut layer -> norm -> ... -> add) on the same masks:
    outputs, input gradient, all 16 parameters after the step, and the readable ``_mask`` of both dropouts.  Host-drawn
    masks consume the global generator exactly as the literal order does (mask first, then a norm's lazy parameters)."""
    sim = npm._C._LIB
    x, dy = rand([3, 6, 8]), rand([3, 6, 8])

    def run(fused):
        np.random.seed(11)
        if rng_kind == 'device':
            npm.set_dropout_rng('device', seed=77)
        enc = npm.layers.TransformerEncoder(num_heads=2, hidden_units=12, norm_first=norm_first, drop_rate=0.3)
        if not fused:
            enc._dense1._fused_relu = lambda: False                      # sends forward / backward to the literal composition
        del sim.calls[:]
        out = np.asarray(enc(x))
        fwd_calls = list(sim.calls)
        masks = [np.asarray(enc._dropout1._mask).copy(), np.asarray(enc._dropout2._mask).copy()]
        del sim.calls[:]
        dx = np.asarray(enc(dy, backprop=True, learning_rate=0.05))
        bwd_calls = list(sim.calls)
        from np_modeling_amd import parallel
        return out, dx, masks, [np.asarray(p).copy() for p in parallel.parameters(enc)], fwd_calls, bwd_calls

    try:
        out_f, dx_f, masks_f, params_f, fwd_calls, bwd_calls = run(True)
        out_u, dx_u, masks_u, params_u, ufwd, ubwd = run(False)
    finally:
        npm.set_dropout_rng('host')
    assert fwd_calls.count('npm_layernorm_dropout_fwd') == 2 and bwd_calls.count('npm_layernorm_dropout_bwd') == 2
    assert 'npm_mask_scale' not in fwd_calls + bwd_calls and 'npm_add' not in fwd_calls + bwd_calls
    assert 'npm_layernorm_dropout_fwd' not in ufwd and ('npm_mask_scale' in ufwd or 'npm_dropout_philox' in ufwd)
    for a, b in zip(masks_f, masks_u):
        assert 0 < (a != 0).mean() < 1
        np.testing.assert_array_equal(a != 0, b != 0)
    np.testing.assert_allclose(out_f, out_u, rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(dx_f, dx_u, rtol=2e-5, atol=2e-6)
    assert len(params_f) == len(params_u) == 16
    for a, b in zip(params_f, params_u):
        np.testing.assert_allclose(a, b, rtol=2e-5, atol=2e-6)


def test_deepcopy_owns_its_device_blocks(npm):
    """Every holder of device memory copies the BLOCK, never just its owner object (two owners = two frees of one
    pool block): dropout masks (p > 0, host- and device-drawn), an encoder with dropout, raw blocks; shallow copies
    and pickling of a block are refused."""
    import copy
    import pickle
    D = npm.device
    x, dy = rand([2, 6, 8]), rand([2, 6, 8])
    np.random.seed(3)
    d = npm.layers.DropOut(0.5)
    d(x)
    twin = copy.deepcopy(d)
    assert twin._mask_dev.ptr != d._mask_dev.ptr and twin._mask_dev._buf is not d._mask_dev._buf
    np.testing.assert_array_equal(np.asarray(twin.backward(dy)), np.asarray(d.backward(dy)))
    np.testing.assert_array_equal(twin._mask, d._mask)
    del twin                                          # returns ITS block only; d still works
    np.testing.assert_array_equal(np.asarray(d.backward(dy)), np.where(d._mask, dy / 0.5, 0).astype(np.float32))
    try:
        npm.set_dropout_rng('device', seed=1)
        e = npm.layers.DropOut(0.5)
        e(x)
        twin = copy.deepcopy(e)
        np.testing.assert_array_equal(twin._mask, e._mask)
        np.testing.assert_array_equal(np.asarray(twin.backward(dy)), np.asarray(e.backward(dy)))
    finally:
        npm.set_dropout_rng('host')
    np.random.seed(0)
    enc = npm.layers.TransformerEncoder(num_heads=2, hidden_units=12, norm_first=True, drop_rate=0.25)
    enc(x)
    twin = copy.deepcopy(enc)
    np.testing.assert_array_equal(np.asarray(twin(dy, backprop=True, learning_rate=0.1)),
                                  np.asarray(enc(dy, backprop=True, learning_rate=0.1)))
    a = D.from_host(x)
    with pytest.raises(TypeError):
        copy.copy(a._buf)
    with pytest.raises(TypeError):
        pickle.dumps(a._buf)
    views = copy.deepcopy([a, a.flat_view(8, [4])])   # one memo: the two views still share one (new) block
    assert views[0]._buf is views[1]._buf and views[0]._buf is not a._buf
    np.testing.assert_array_equal(np.asarray(views[1]), x.reshape(-1)[8:12])


def test_loss_reads_refilled_targets(npm):
    """A caller may refill the SAME host array between steps (targets[:] = next_batch); the reference reads it
    afresh at every forward (loss.py:21-25)."""
    D = npm.device
    y = D.from_host(np.ones([4, 3], dtype=np.float32))
    targets = np.zeros([4, 3], dtype=np.float32)
    mse = npm.loss.MSELoss()
    assert mse(y, targets) == pytest.approx(1.0)
    targets[:] = 1.0
    assert mse(y, targets) == pytest.approx(0.0)


@pytest.mark.parametrize('masked', [False, True])
def test_mha_fused_core_path_on_simulator(npm, masked):
    """Head size 16 takes the fused attention core (npm_mha_core_*): operand descriptors of the packed and the
    cross-attention layouts, the mask plumbing, against the oracle."""
    from oracle import np_oracle as O
    rng = np.random.default_rng(1)
    names = ['wq', 'wk', 'wv', 'wo', 'bq', 'bk', 'bv', 'bo']
    for kv_len in (None, 9):
        np.random.seed(0)
        layer = npm.layers.MultiHeadAttention(num_heads=2)
        x = rng.standard_normal([2, 6, 32]).astype(np.float32)
        kv = None if kv_len is None else rng.standard_normal([2, kv_len, 32]).astype(np.float32)
        skv = kv_len or 6
        mask = (rng.random([2, 2, 6, skv]) < 0.7) if masked else None
        if masked:
            mask[..., 0] = True
        layer(x, kv) if kv is not None else layer(x)                      # lazy init
        for n in names[:4]:       # O(1) scores: the simulator keeps the log-sum-exp in fp32, like the kernel
            getattr(layer, '_' + n).set(np.asarray(getattr(layer, '_' + n)) / np.float32(6.0))
        if masked:                                                       # a prebuilt AttnMask is taken as is (made once, reused)
            prebuilt = npm.device.AttnMask(mask, 2, 2, 6, skv)
            again = layer(x, kv, mask=prebuilt) if kv is not None else layer(x, mask=prebuilt)
            assert layer._mask is prebuilt and 'npm_mha_mask_summary' in npm._C._LIB.calls
        out = layer(x, kv, mask=mask) if kv is not None else layer(x, mask=mask)
        if masked:
            np.testing.assert_array_equal(np.asarray(again), np.asarray(out))
        assert layer._core and 'npm_mha_core_fwd' in npm._C._LIB.calls
        assert layer._packed is (kv is None)
        p = {n: np.asarray(getattr(layer, '_' + n)).astype(np.float64) for n in names}
        want, cache = O.mha_fwd(p, x.astype(np.float64), None if kv is None else kv.astype(np.float64), mask=mask)
        assert_close(out, want, tol=1e-6)
        dy = rng.standard_normal([2, 6, 32]).astype(np.float32)
        dq, dk, dv = layer(dy, backprop=True, learning_rate=0.1)
        (wq_, wk_, wv_), grads = O.mha_bwd(p, cache, dy.astype(np.float64))
        assert_close(dq, wq_, tol=1e-6)
        assert_close(np.asarray(dk) + np.asarray(dv), wk_ + wv_, tol=1e-6)
        for n in names:
            assert_close(getattr(layer, '_' + n), p[n] - 0.1 * grads[n], tol=1e-6, what=n)


@pytest.mark.parametrize('name', ['ref_mha_self_d16', 'ref_mha_cross_d64', 'ref_encoder_prenorm', 'ref_decoder_postnorm',
                                  'ref_conv_k3', 'ref_dense', 'ref_softmax', 'ref_layernorm'])
def test_reference_shape_fixtures_on_simulator(npm, name):
    """The host side of tests/test_gpu_refshapes.py: binding parameters into private attributes, deep copies, recording
    optimizers and the composite layers' descriptors, at the reference's own test shapes, against reference outputs."""
    import refshape_runner as RR
    got, ref = RR.run(npm, name)
    RR.compare(got, ref, tol=1e-5)


def test_dropout_device_rng_plumbing(npm):
    """set_dropout_rng('device'): the mask comes from npm_dropout_philox, `_mask` is read back lazily, successive
    calls advance the offset, 'host' restores the reference's np.random.binomial draw."""
    from oracle import np_oracle as O
    x = np.arange(24, dtype=np.float32).reshape(4, 6) + 1
    try:
        npm.set_dropout_rng('device', seed=5)
        d = npm.layers.DropOut(0.25)
        y = np.asarray(d(x))
        want = O.dropout_philox_mask(24, 0.75, 5, 0).reshape(4, 6)
        np.testing.assert_array_equal(d._mask != 0, want)
        np.testing.assert_allclose(y, np.where(want, x / 0.75, 0), rtol=1e-6)
        np.testing.assert_allclose(np.asarray(d.backward(x)), np.where(want, x / 0.75, 0), rtol=1e-6)
        d(x)
        np.testing.assert_array_equal(d._mask != 0, O.dropout_philox_mask(24, 0.75, 5, 1).reshape(4, 6))
    finally:
        npm.set_dropout_rng('host')
    np.random.seed(1)
    h = npm.layers.DropOut(0.25)
    h(x)
    np.random.seed(1)
    np.testing.assert_array_equal(h._mask, np.random.binomial(n=1, p=0.75, size=24).reshape(4, 6))


def test_loss_broadcasts_resident_targets(npm, capsys):
    """Targets of a broadcastable but different shape are valid in the reference (``y - targets``, loss.py:24,28);
    `Trainer.train` uploads targets once, so the loss has to expand a RESIDENT [B, 1] array too -- the kernels read
    y.size elements."""
    D = npm.device
    y = np.arange(12, dtype=np.float32).reshape(4, 3)
    t = np.array([[1.0], [2.0], [3.0], [4.0]], dtype=np.float32)
    for cls, want_l, want_g in ((npm.loss.MSELoss, np.sum((y - t) ** 2) / y.size, 2 * (y - t) / y.size),
                                (npm.loss.CrossEntropyLoss, -np.sum(t * np.log(y + 1)), -t / (y + 1))):
        loss = cls()
        yy = y + 1 if cls is npm.loss.CrossEntropyLoss else y
        resident = D.from_host(t)
        assert loss(D.from_host(yy), resident) == pytest.approx(float(want_l), rel=1e-6)
        np.testing.assert_allclose(np.asarray(loss(backprop=True)), want_g, rtol=1e-6)
        first = loss._expanded[2]
        loss(D.from_host(yy), resident)                   # same resident targets: expanded once
        assert loss._expanded[2] is first
        with pytest.raises(ValueError):
            loss(D.from_host(yy), D.from_host(np.zeros([5, 1], dtype=np.float32)))
    np.random.seed(0)
    trainer = npm.train.Trainer([npm.layers.Dense(units=3)])
    trainer.train(rand([4, 5]), t, 2, npm.optimizer.SGDOptimizer(1e-2))       # [4, 1] targets against [4, 3] outputs
    assert capsys.readouterr().out.count('Loss:') == 2


def test_softmax_cross_entropy_chain_on_simulator(npm):
    """loss_test.py:49-66 through the product's host layer (call protocol of Layer.__call__ with backprop=True on a
    loss and on an activation that takes no optimizer)."""
    g = load_golden('softmax_ce')
    ce, softmax = npm.loss.CrossEntropyLoss(), npm.layers.Softmax()
    y = npm.as_device(g['y'])
    prob = softmax(y)
    np.testing.assert_allclose(ce(prob, g['targets']), g['ce'], rtol=1e-6)
    dy = softmax(ce(y, g['targets'], backprop=True), backprop=True)
    assert_close(dy, g['dy'], tol=2e-6)


def test_pick_device_policy():
    """One process per GPU under either kind of launcher: all GPUs visible -> LOCAL_RANK; devices masked per rank
    (one visible) -> device 0; NPM_DEVICE wins; a mask that is neither is an error, not a silent share."""
    from np_modeling_amd import _C
    assert _C.pick_device(8, {'LOCAL_RANK': '5'}) == 5
    assert _C.pick_device(1, {'LOCAL_RANK': '5'}) == 0          # HIP_VISIBLE_DEVICES=<one GPU> per rank
    assert _C.pick_device(8, {}) == 0
    assert _C.pick_device(0, {'LOCAL_RANK': '3'}) == 0          # no device: npm_init reports it
    assert _C.pick_device(8, {'LOCAL_RANK': '5', 'NPM_DEVICE': '2'}) == 2
    with pytest.raises(_C.NpmError, match='only 4 HIP devices'):
        _C.pick_device(4, {'LOCAL_RANK': '5'})


def test_current_math_follows_the_library(npm):
    """A mode set through the tuning knob (NPM_TUNE=10=<mode>, tools/gemm_bench.py --tune) is what current_math()
    reports: the attention routing (device.mha_core_supported) must not see a stale 'f32'."""
    from np_modeling_amd import _C
    assert _C.current_math() == 'f32'
    _C.lib().npm_set_math(2)
    try:
        assert _C.current_math() == 'bf16x3'
    finally:
        _C.lib().npm_set_math(0)


def test_rendezvous_file_is_private_and_launch_bound(tmp_path, monkeypatch):
    """The derived rendezvous file lives in a per-user 0700 directory; a file of ANOTHER launch (stale explicit path,
    planted id) is skipped, a file of another user is refused."""
    import stat
    from np_modeling_amd import _C, parallel
    monkeypatch.setenv('TMPDIR', str(tmp_path))
    import tempfile
    monkeypatch.setattr(tempfile, 'tempdir', None)
    monkeypatch.delenv('NPM_RENDEZVOUS_FILE', raising=False)
    monkeypatch.setenv('MASTER_PORT', '29123')
    path = parallel.rendezvous_path()
    d = os.path.dirname(path)
    assert os.path.dirname(d) == str(tmp_path) and stat.S_IMODE(os.stat(d).st_mode) == 0o700
    os.chmod(d, 0o755)
    with pytest.raises(_C.NpmError, match='not a private directory'):
        parallel.rendezvous_path()
    os.chmod(d, 0o700)

    uid = bytes(range(128))
    monkeypatch.setattr(parallel.RcclCommunicator, 'new_unique_id', staticmethod(lambda: uid))
    explicit = str(tmp_path / 'explicit_id')
    monkeypatch.setenv('NPM_RENDEZVOUS_FILE', explicit)
    monkeypatch.setenv('NPM_LAUNCH_TOKEN', 'old-launch')
    assert parallel._exchange_unique_id(0, 2) == uid          # an earlier launch leaves its file behind
    monkeypatch.setenv('NPM_LAUNCH_TOKEN', 'new-launch')
    with pytest.raises(_C.NpmError, match='only a file of another launch'):
        parallel._exchange_unique_id(1, 2, timeout=0.2)       # stale id: skipped, not handed to ncclCommInitRank
    assert parallel._exchange_unique_id(0, 2) == uid          # rank 0 of the new launch replaces it ...
    assert parallel._exchange_unique_id(1, 2, timeout=5) == uid   # ... and the readers take that one
    assert stat.S_IMODE(os.stat(explicit).st_mode) == 0o600
    with open(explicit, 'wb') as f:
        f.write(uid)                                          # a bare 128-byte id (the old format, or a planted one)
    with pytest.raises(_C.NpmError, match='another launch'):
        parallel._exchange_unique_id(1, 2, timeout=0.2)

    monkeypatch.delenv('NPM_RENDEZVOUS_FILE')
    monkeypatch.setenv('LOCAL_WORLD_SIZE', '8')
    with pytest.raises(_C.NpmError, match='ONE node'):
        parallel._exchange_unique_id(3, 16, timeout=0.2)      # multi-node: fails at once with the reason, not after 300 s


def test_self_launcher_eight_ranks(capfd):
    """The driver's N = 8 shape on CPU ranks: eight fresh children through the launcher's private file."""
    from np_modeling_amd import launch
    code = launch.spawn_ranks(8, [sys.executable, os.path.join(ROOT, 'tests', 'uid_worker.py')], build=False,
                              env=_clean_env())
    out = capfd.readouterr()
    assert code == 0, out.err
    assert 'rank 0/8: id ok' in out.out
    assert all(f'rank {r}/8: id ok' in out.err for r in range(1, 8)), out.err


def test_terminated_launcher_takes_its_ranks_along(tmp_path):
    """`timeout -k 10 400 python bench.py --gpus N` SIGTERMs the launcher: its ranks (sleeping here) must not
    survive it, and the launcher's exit code is 143.  A SIGKILLed launcher cannot clean up: the ranks asked the
    kernel for a SIGTERM on the launcher's death (PR_SET_PDEATHSIG)."""
    import signal
    import time
    pidfile = tmp_path / 'pids'

    def start():
        if pidfile.exists():
            pidfile.unlink()
        env = _clean_env(UID_WORKER_FAIL_RANK='99', UID_WORKER_HANG='1', UID_WORKER_PIDFILE=str(pidfile))
        p = subprocess.Popen([sys.executable, '-m', 'np_modeling_amd.launch', '--gpus', '3',
                              os.path.join(ROOT, 'tests', 'uid_worker.py')], env=env, cwd=ROOT,
                             stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        deadline = time.monotonic() + 120
        while time.monotonic() < deadline:                    # wait until all three ranks sleep
            if pidfile.exists() and len(pidfile.read_text().split()) == 3:
                break
            time.sleep(0.05)
        pids = [int(x) for x in pidfile.read_text().split()]
        assert len(pids) == 3
        return p, pids

    def gone(pid):
        try:
            with open(f'/proc/{pid}/stat') as f:
                return f.read().rsplit(')', 1)[1].split()[0] == 'Z'
        except OSError:
            return True

    for sig, want in ((signal.SIGTERM, 143), (signal.SIGKILL, -9)):
        p, pids = start()
        p.send_signal(sig)
        p.communicate(timeout=60)
        assert p.returncode == want
        deadline = time.monotonic() + 20
        while not all(gone(pid) for pid in pids) and time.monotonic() < deadline:
            time.sleep(0.05)
        assert all(gone(pid) for pid in pids), f'ranks survived a launcher ended by signal {sig}: {pids}'


def test_gpu_local_cpus_from_a_fake_sysfs(tmp_path, monkeypatch):
    """launch.gpu_local_cpus / bind_to_gpu_cpus: KFD topology order -> PCI address -> local_cpulist, per-rank device masks,
    and the affinity call itself (on whatever CPUs this test may use)."""
    from np_modeling_amd import launch
    for var in ('ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES', 'NPM_DEVICE', 'NPM_BIND_CPUS'):
        monkeypatch.delenv(var, raising=False)
    allowed = sorted(os.sched_getaffinity(0))
    half = allowed[:max(1, len(allowed) // 2)]
    lists = {0: '0-3,8', 1: ','.join(str(c) for c in half)}
    nodes = tmp_path / 'class' / 'kfd' / 'kfd' / 'topology' / 'nodes'
    # node 0 is a CPU (simd_count 0); nodes 1 and 2 are GPUs at 0000:05:00.0 and 0001:c3:00.0
    for node, (simd, loc, dom) in enumerate([(0, 0, 0), (1024, 0x0500, 0), (1024, 0xc300, 1)]):
        d = nodes / str(node)
        d.mkdir(parents=True)
        (d / 'properties').write_text(f'cpu_cores_count 0\nsimd_count {simd}\nlocation_id {loc}\ndomain {dom}\n')
    for gpu, bdf in enumerate(['0000:05:00.0', '0001:c3:00.0']):
        d = tmp_path / 'bus' / 'pci' / 'devices' / bdf
        d.mkdir(parents=True)
        (d / 'local_cpulist').write_text(lists[gpu] + '\n')
    root = str(tmp_path)
    assert launch.gpu_local_cpus(0, root) == {0, 1, 2, 3, 8}
    assert launch.gpu_local_cpus(1, root) == set(half)
    assert launch.gpu_local_cpus(2, root) is None                       # no such GPU
    monkeypatch.setenv('HIP_VISIBLE_DEVICES', '1')                      # a per-rank mask: the one visible device is this rank's
    assert launch.gpu_local_cpus(5, root) == set(half)
    monkeypatch.delenv('HIP_VISIBLE_DEVICES')
    assert launch.gpu_local_cpus(0, str(tmp_path / 'nothing')) is None
    monkeypatch.setenv('NPM_BIND_CPUS', '0')
    assert launch.bind_to_gpu_cpus(1, root) == {'bound': False, 'local_rank': 1, 'reason': 'disabled'}
    monkeypatch.delenv('NPM_BIND_CPUS')
    before = os.sched_getaffinity(0)
    try:
        info = launch.bind_to_gpu_cpus(1, root)
        if len(allowed) > 1:
            assert info['bound'] and info['cpus'] == len(half) and os.sched_getaffinity(0) == set(half)
        else:
            assert not info['bound']
    finally:
        os.sched_setaffinity(0, before)
    assert launch.die_with_parent() is True                             # PR_SET_PDEATHSIG through the pre-bound prctl


def test_explicit_rendezvous_file_shared_by_ranks_of_different_parents(tmp_path):
    """Two ranks started by DIFFERENT parent processes (two shells, two nodes on a shared file system) that share an explicit
    NPM_RENDEZVOUS_FILE: with one NPM_LAUNCH_TOKEN for the job (or a job id from the launcher) rank 1 accepts rank 0's id whoever
    its parent is; WITHOUT one the ranks fail at once with the reason (round-4 advisor: MASTER_ADDR:MASTER_PORT alone stood in
    for the token, and a stale file of an earlier job at the same address passed for a fresh one)."""
    base = dict(_clean_env(), NPM_RENDEZVOUS_FILE=str(tmp_path / 'shared_id'), MASTER_ADDR='127.0.0.1', MASTER_PORT='29517',
                WORLD_SIZE='2', LOCAL_WORLD_SIZE='1')
    worker = os.path.join(ROOT, 'tests', 'uid_worker.py')
    # each rank is the child of its own intermediate process
    hop = 'import subprocess, sys; sys.exit(subprocess.call([sys.executable, sys.argv[1]]))'

    def run(env):
        procs = [subprocess.Popen([sys.executable, '-c', hop, worker], env=dict(env, RANK=str(r), LOCAL_RANK='0'), cwd=ROOT,
                                  stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in (1, 0)]
        outs = [p.communicate(timeout=120)[0] for p in procs]
        return [p.returncode for p in procs], outs

    for extra in (dict(NPM_LAUNCH_TOKEN='job-4711'), dict(TORCHELASTIC_RUN_ID='run-17'), dict(SLURM_JOB_ID='990')):
        codes, outs = run(dict(base, **extra))
        assert codes == [0, 0], outs
        assert 'rank 1/2: id ok' in outs[0] and 'rank 0/2: id ok' in outs[1]
    codes, outs = run(dict(base, TORCHELASTIC_RUN_ID='none'))                 # torch.distributed.run's default: not a job id
    assert all(c != 0 for c in codes), outs
    assert all('NPM_LAUNCH_TOKEN' in o for o in outs), outs


def test_explicit_file_without_token_or_sibling_proof_fails_at_once(tmp_path, monkeypatch):
    """An explicit NPM_RENDEZVOUS_FILE, no token, no job id and NO LOCAL_WORLD_SIZE (ranks started from separate shells, or
    on several nodes by a launcher that does not export it): the parent-pid token would differ per rank and every rank but 0
    would poll for 300 s.  The error comes at once; with LOCAL_WORLD_SIZE == WORLD_SIZE (siblings) the parent token is fine."""
    from np_modeling_amd import _C, parallel
    for var in ('NPM_LAUNCH_TOKEN', 'TORCHELASTIC_RUN_ID', 'SLURM_JOB_ID', 'LOCAL_WORLD_SIZE'):
        monkeypatch.delenv(var, raising=False)
    monkeypatch.setenv('NPM_RENDEZVOUS_FILE', str(tmp_path / 'id'))
    monkeypatch.setenv('WORLD_SIZE', '2')
    with pytest.raises(_C.NpmError, match='NPM_LAUNCH_TOKEN'):
        parallel._launch_token()
    monkeypatch.setenv('LOCAL_WORLD_SIZE', '2')
    assert parallel._launch_token().startswith(f'{os.getppid()}-')
    monkeypatch.setenv('WORLD_SIZE', '1')                      # a single rank needs no proof
    monkeypatch.delenv('LOCAL_WORLD_SIZE')
    assert parallel._launch_token().startswith(f'{os.getppid()}-')


def test_ipc_mode_is_set_for_externally_launched_ranks():
    """HSA_ENABLE_IPC_MODE_LEGACY=0 (dmabuf IPC: what this pool's driver supports) must be in the environment before the
    runtime initialises whenever the process is one of several ranks -- also under an external launcher, not only under
    np_modeling_amd/launch.py.  A value the user chose is kept; a single process is left alone."""
    from np_modeling_amd import _C, launch
    env = {'WORLD_SIZE': '8'}
    assert _C.ipc_env_for_multi_rank(env) is True and env['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'
    env = {'WORLD_SIZE': '2', 'HSA_ENABLE_IPC_MODE_LEGACY': '1'}
    assert _C.ipc_env_for_multi_rank(env) is True and env['HSA_ENABLE_IPC_MODE_LEGACY'] == '1'
    for env in ({}, {'WORLD_SIZE': '1'}, {'WORLD_SIZE': ''}):
        assert _C.ipc_env_for_multi_rank(env) is False and 'HSA_ENABLE_IPC_MODE_LEGACY' not in env
    assert launch.rank_environment(0, 2, '/tmp/x', base={})['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'
    # parallel.init() and _C.lib() call it before the device library is loaded / initialised
    import inspect
    from np_modeling_amd import parallel
    src = inspect.getsource(parallel.init)
    assert src.index('ipc_env_for_multi_rank') < src.index('_C.lib()')
    src = inspect.getsource(_C.lib)
    assert src.index('ipc_env_for_multi_rank') < src.index('load_library') < src.index('npm_init')


def test_stale_rendezvous_file_with_the_same_address_is_rejected(tmp_path, monkeypatch):
    """A crashed job leaves its file at a reused NPM_RENDEZVOUS_FILE; the next job has the SAME MASTER_ADDR / MASTER_PORT.  Its
    ranks must skip that file whichever way they were launched: self-launched ranks carry a fresh random token per launch,
    externally launched ranks of one node their launcher's pid + start time, ranks with a job id that id."""
    from np_modeling_amd import _C, launch, parallel
    uid_old, uid_new = bytes(range(128)), bytes(reversed(range(128)))
    path = str(tmp_path / 'reused_id')
    for var in ('NPM_LAUNCH_TOKEN', 'TORCHELASTIC_RUN_ID', 'SLURM_JOB_ID', 'LOCAL_WORLD_SIZE'):
        monkeypatch.delenv(var, raising=False)
    monkeypatch.setenv('NPM_RENDEZVOUS_FILE', path)
    monkeypatch.setenv('MASTER_ADDR', '127.0.0.1')
    monkeypatch.setenv('MASTER_PORT', '29600')
    monkeypatch.setenv('WORLD_SIZE', '2')

    # (1) two launches of the self-launcher: different tokens although everything else in the environment is equal
    env_a = launch.rank_environment(0, 2, path, base=dict(os.environ))
    env_b = launch.rank_environment(0, 2, path, base=dict(os.environ))
    assert env_a['NPM_LAUNCH_TOKEN'] != env_b['NPM_LAUNCH_TOKEN'] and env_a['NPM_LAUNCHER_PID'] == str(os.getpid())
    tok = launch.new_launch_token()
    assert [launch.rank_environment(r, 2, path, base={}, token=tok)['NPM_LAUNCH_TOKEN'] for r in (0, 1)] == [tok, tok]
    monkeypatch.setenv('NPM_LAUNCH_TOKEN', env_a['NPM_LAUNCH_TOKEN'])
    monkeypatch.setattr(parallel.RcclCommunicator, 'new_unique_id', staticmethod(lambda: uid_old))
    assert parallel._exchange_unique_id(0, 2) == uid_old                      # job A writes its id and crashes
    monkeypatch.setenv('NPM_LAUNCH_TOKEN', env_b['NPM_LAUNCH_TOKEN'])
    with pytest.raises(_C.NpmError, match='only a file of another launch'):
        parallel._exchange_unique_id(1, 2, timeout=0.2)                       # job B's rank 1 races its rank 0: skips the stale file
    monkeypatch.setattr(parallel.RcclCommunicator, 'new_unique_id', staticmethod(lambda: uid_new))
    assert parallel._exchange_unique_id(0, 2) == uid_new
    assert parallel._exchange_unique_id(1, 2, timeout=5) == uid_new

    # (2) an external launcher on one node, no token, no job id: the parent's pid + start time.  A file packed under the
    # round-4 token (the address alone) -- or under any other parent -- is not accepted
    monkeypatch.delenv('NPM_LAUNCH_TOKEN')
    monkeypatch.setenv('LOCAL_WORLD_SIZE', '2')
    token = parallel._launch_token()
    assert token.startswith(f'{os.getppid()}-') and '29600' not in token
    stale = parallel._ID_MAGIC + (33).to_bytes(2, 'little') + b'explicit-127.0.0.1:29600-0'.ljust(33, b' ') + uid_old
    with open(path, 'wb') as f:
        f.write(stale)
    with pytest.raises(_C.NpmError, match='only a file of another launch'):
        parallel._exchange_unique_id(1, 2, timeout=0.2)

    # (3) a job id: equal for the ranks of a job, different for the next job at the same address
    monkeypatch.setenv('SLURM_JOB_ID', '1001')
    assert parallel._exchange_unique_id(0, 2) == uid_new
    monkeypatch.setenv('SLURM_JOB_ID', '1002')
    with pytest.raises(_C.NpmError, match='only a file of another launch'):
        parallel._exchange_unique_id(1, 2, timeout=0.2)

    # (4) several nodes (or parents) and nothing that identifies the launch: an error at once, not a guess
    monkeypatch.delenv('SLURM_JOB_ID')
    monkeypatch.setenv('LOCAL_WORLD_SIZE', '1')
    with pytest.raises(_C.NpmError, match='NPM_LAUNCH_TOKEN'):
        parallel._exchange_unique_id(1, 2, timeout=0.2)


def test_rank_dies_with_its_launcher_only(monkeypatch):
    """PR_SET_PDEATHSIG is armed by a rank only when this package's launcher is its parent (NPM_LAUNCHER_PID == getppid()); a rank
    started from a shell or wrapper that exits on purpose is left alone; NPM_DIE_WITH_PARENT=0 switches it off."""
    from np_modeling_amd import launch
    calls = []
    monkeypatch.setattr(launch, 'die_with_parent', lambda: calls.append(1) or True)
    monkeypatch.delenv('NPM_LAUNCHER_PID', raising=False)
    monkeypatch.delenv('NPM_DIE_WITH_PARENT', raising=False)
    assert launch.die_with_launcher() is False and not calls                  # an external launcher's rank
    monkeypatch.setenv('NPM_LAUNCHER_PID', str(os.getppid() + 1))
    assert launch.die_with_launcher() is False and not calls                  # the launcher is not our parent (a wrapper in between)
    monkeypatch.setenv('NPM_LAUNCHER_PID', str(os.getppid()))
    assert launch.die_with_launcher() is True and calls == [1]
    monkeypatch.setenv('NPM_DIE_WITH_PARENT', '0')
    assert launch.die_with_launcher() is False and calls == [1]


# ---- one optimizer launch per backward --------------------------------------------------
def _encoder_step(npm, optimizer_, steps=2, rebind=None, norm_first=True, decoder=False):
    """``steps`` forward + backward steps of a small encoder (or decoder) on the simulator; returns (parameters, number of
    optimizer launches of the last backward, what GradScope recorded about it)."""
    from np_modeling_amd import parallel
    np.random.seed(3)
    rng = np.random.default_rng(3)
    cls = npm.layers.TransformerDecoder if decoder else npm.layers.TransformerEncoder
    layer = cls(num_heads=2, hidden_units=20, norm_first=norm_first)
    x = rng.standard_normal([2, 5, 8]).astype(np.float32)
    kv = rng.standard_normal([2, 7, 8]).astype(np.float32)
    dy = rng.standard_normal([2, 5, 8]).astype(np.float32)
    args = (x, kv) if decoder else (x,)
    launches = None
    for step in range(steps):
        layer(*args)
        if rebind and step == 0:
            rebind(layer)
        sim = npm._C._LIB
        before = sim.calls.count('npm_axpy') + sim.calls.count('npm_adam_step')
        layer(dy, backprop=True, optimizer_=optimizer_)
        launches = sim.calls.count('npm_axpy') + sim.calls.count('npm_adam_step') - before
    names = ('_w', '_b', '_wq', '_wk', '_wv', '_wo', '_bq', '_bk', '_bv', '_bo', '_gamma', '_beta')
    return [np.asarray(p).copy() for p in parallel.parameters(layer)], launches, dict(parallel.GradScope.last), layer


@pytest.mark.parametrize('norm_first', [True, False])
@pytest.mark.parametrize('kind', ['sgd', 'adam', 'reference sgd'])
def test_one_optimizer_launch_per_encoder_backward(npm, monkeypatch, kind, norm_first):
    """The 16 parameters of an encoder sit back to back in one arena, ordered as backward produces their gradients; the
    gradient bucket mirrors it, and the 16 ``optimizer_.update(obj, attribute, grad)`` calls of a backward -- still made one by
    one, with the reference's ``id(obj).attribute`` keying -- run as ONE axpy / ONE Adam kernel.  Bit-equal to the
    per-parameter launches (NPM_COALESCE_UPDATES=0).  'reference sgd': the reference's own class, verbatim behaviour."""
    D = npm.device

    class ReferenceSGD:                   # reference optimizer.py:12-33, restated: getattr -> variable -= lr * gradient -> setattr
        def __init__(self, lr):
            self._learning_rate, self.keys = lr, []

        def update(self, obj, attribute, gradient):
            self.keys.append(f'{id(obj)}.{attribute}')
            variable = getattr(obj, attribute)
            variable -= self._learning_rate * gradient
            setattr(obj, attribute, variable)

    def make():
        return {'sgd': lambda: npm.optimizer.SGDOptimizer(0.05), 'adam': lambda: npm.optimizer.AdamOptimizer(0.01),
                'reference sgd': lambda: ReferenceSGD(0.05)}[kind]()

    monkeypatch.setattr(D, 'COALESCE_UPDATES', True)
    opt = make()
    got, launches, last, layer = _encoder_step(npm, opt, norm_first=norm_first)
    assert launches == 1 and last['update_launches'] == 1 and last['updates'] == 16 and last['arena'], (launches, last)
    assert layer._arena.live() == 16
    if kind == 'reference sgd':
        assert len(opt.keys) == 32 and len(set(opt.keys)) == 16      # update() called per parameter, both steps, same keys
    if kind == 'adam':
        assert len(opt._state) == 16 and all(entry[0] == 3 for entry in opt._state.values())    # per-identifier state, two steps each
    monkeypatch.setattr(D, 'COALESCE_UPDATES', False)
    want, launches, last, layer = _encoder_step(npm, make(), norm_first=norm_first)
    assert launches == 16 and last['update_launches'] is None and layer._arena is None
    assert len(got) == len(want) == 16
    for a, b in zip(got, want):
        np.testing.assert_array_equal(a, b)


def test_queued_updates_keep_program_order(npm, monkeypatch):
    """Any ``Optimizer`` subclass may run inside a backward's ``coalesced_updates()``: only ``+=`` / ``-=`` are queued (and
    launched sorted by address), so everything else must see them in program order.  Weight decay after the step
    (``var -= lr * g; var *= 1 - wd``), a momentum buffer that is updated and then applied (write -> read of one array in two
    queued updates), two updates of one variable, and a host read in between -- each equal to the unqueued run."""
    D = npm.device
    monkeypatch.setattr(D, 'COALESCE_UPDATES', True)

    def decay(var, g, vel):
        var -= 0.5 * g
        var *= 0.5                      # not queued: must come AFTER the queued step
        return var

    def momentum(var, g, vel):
        vel *= 0.9
        vel += g                        # queued, writes vel
        var -= 0.1 * vel                # queued, READS vel: must see the line above applied
        return var

    def twice(var, g, vel):
        var -= 0.5 * g
        var += 0.25 * var.copy()        # copy() reads var between two queued updates of it
        return var

    def peek(var, g, vel):
        var -= 0.5 * g
        seen.append(np.asarray(var).copy())
        var -= 0.5 * g
        return var

    for rule in (decay, momentum, twice, peek):
        results = []
        for queued in (False, True):
            seen = []
            block = D.from_host(np.arange(24, dtype=np.float32))              # neighbours in one block, like an arena
            var, vel = block.flat_view(8, [8]), block.flat_view(0, [8])
            g = D.from_host(np.linspace(-1, 1, 8).astype(np.float32))
            if queued:
                with D.coalesced_updates() as queue:
                    rule(var, g, vel)
                assert queue.drains >= 1, rule.__name__
                assert npm._C._ORDER_HOOK is None
            else:
                rule(var, g, vel)
            results.append((block.numpy().copy(), [s.copy() for s in seen]))
        np.testing.assert_array_equal(results[0][0], results[1][0], err_msg=rule.__name__)
        for a, b in zip(results[0][1], results[1][1]):
            np.testing.assert_array_equal(a, b, err_msg=rule.__name__)
    # the advisor's reproduction: 0.1 when called directly
    var, g = D.from_host(np.float32([1.0])), D.from_host(np.float32([8.0]))
    with D.coalesced_updates():
        var -= 0.1 * g
        var *= 0.5
    np.testing.assert_allclose(var.numpy(), [0.1], rtol=1e-6)
    # updates of disjoint arrays still wait together and run joined
    block = D.from_host(np.zeros(16, dtype=np.float32))
    grads = D.from_host(np.ones(16, dtype=np.float32))
    with D.coalesced_updates() as queue:
        for at in (8, 0, 12, 4):
            v = block.flat_view(at, [4])
            v -= 1.0 * grads.flat_view(at, [4])
    assert queue.launches == 1 and queue.drains == 0 and queue.updates == 4
    np.testing.assert_array_equal(block.numpy(), -np.ones(16, dtype=np.float32))


def test_rebound_parameter_leaves_the_arena(npm, monkeypatch):
    """Weight binders assign arrays into private attributes (reference layers/utils.py:52-88).  Such a parameter is no longer
    part of the arena: it is updated by a launch of its own, the others still together (split where it used to sit); results
    equal to the per-parameter path."""
    D = npm.device

    def rebind(enc):
        enc._dense1._linear._w = np.asarray(enc._dense1._linear._w) * np.float32(0.5)      # a host array, like a binder's
        enc._self_attention._wk = D.from_host(np.asarray(enc._self_attention._wk) * np.float32(2.0))   # breaks the packed q/k/v too

    monkeypatch.setattr(D, 'COALESCE_UPDATES', True)
    got, launches, last, layer = _encoder_step(npm, npm.optimizer.SGDOptimizer(0.05), rebind=rebind)
    assert layer._arena.live() == 14
    assert 3 <= launches <= 6 and last['updates'] == 16, (launches, last)
    monkeypatch.setattr(D, 'COALESCE_UPDATES', False)
    want, launches, _, _ = _encoder_step(npm, npm.optimizer.SGDOptimizer(0.05), rebind=rebind)
    assert launches == 16
    for a, b in zip(got, want):
        np.testing.assert_array_equal(a, b)


def test_decoder_bucket_and_single_launch(npm, monkeypatch):
    """TransformerDecoder: 26 parameters in one arena, one optimizer launch per backward; with two ranks (a recording
    transport) its gradients go out in one bucket as five collectives (feed-forward | norm / dense1 | cross-attention |
    self-attention | the rest) -- round 4 opened the scope without a size and every gradient was a collective of its own."""
    from np_modeling_amd import parallel
    D = npm.device
    monkeypatch.setattr(D, 'COALESCE_UPDATES', True)
    for norm_first in (True, False):
        got, launches, last, layer = _encoder_step(npm, npm.optimizer.SGDOptimizer(0.05), norm_first=norm_first, decoder=True)
        assert launches == 1 and last['updates'] == 26 and layer._arena.live() == 26, (launches, last)
    sent = []

    class Recorder(parallel.Communicator):
        rank, world_size = 0, 2

        def allreduce_async(self, flat, op):
            sent.append((flat.ptr, flat.size))

        def wait(self):
            pass

    parallel.set_communicator(Recorder())
    try:
        for norm_first in (True, False):
            sent.clear()
            _, launches, last, layer = _encoder_step(npm, npm.optimizer.SGDOptimizer(0.05), steps=1, norm_first=norm_first, decoder=True)
            assert last['collectives'] == len(sent) and 4 <= len(sent) <= 6, sent
            total = sum(n for _, n in sent)
            assert layer._arena.size <= total <= layer._arena.size + 64        # every gradient once; the padding travels along
            starts = sorted(p for p, _ in sent)
            assert all(b > a for a, b in zip(starts, starts[1:]))               # disjoint, ascending: arena order = production order
            assert launches == 1
    finally:
        parallel.set_communicator(None)


def test_bench_traffic_figure_belongs_to_this_build(monkeypatch):
    """bench.py quotes roofline.traffic from profiles/pmc_traffic.json only when that file was collected for the sources this
    build is made of (np_modeling_amd._C.source_id) -- the figure of another build is reported as null with the reason, and the
    source names the profiling session."""
    import argparse
    import json
    import bench
    from np_modeling_amd import _C
    args = argparse.Namespace(batch=256, seq=512, features=1024, heads=8, hidden=4096)
    with open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')) as f:
        recorded = json.load(f)
    monkeypatch.setattr(_C, 'library_is_current', lambda: True)
    monkeypatch.setattr(_C, 'source_id', lambda: recorded.get('source_id', 'none recorded'))
    value, source = bench.load_pmc_traffic(args)
    if 'source_id' in recorded:
        assert value == recorded['gemm_family_bytes_per_launch'] and recorded['session'] in source and 'this build' in source
    monkeypatch.setattr(_C, 'source_id', lambda: '0123456789abcdef')
    value, source = bench.load_pmc_traffic(args)
    assert value is None and 'stale' in source and '0123456789abcdef' in source
    monkeypatch.setattr(_C, 'source_id', lambda: recorded.get('source_id', 'x'))
    monkeypatch.setattr(_C, 'library_is_current', lambda: False)
    value, source = bench.load_pmc_traffic(args)
    assert value is None
    args.batch = 8
    assert bench.load_pmc_traffic(args)[0] is None
    assert len(_C.source_id.__wrapped__()) == 16 if hasattr(_C.source_id, '__wrapped__') else True


def test_mha_row_terms_from_the_dctx_gemm_on_simulator(npm, monkeypatch):
    """Head size 128: the GEMM that produces dctx = dy wo also takes -scale * (dctx . ctx) per query and head (NPM_EPI_ROWDOT,
    [H, B * Sq]) and the fused attention backward gets it as ``neg_delta`` (the simulator checks it against its own dctx and
    ctx); NPM_ATTN_ROWDOT=0 is the pass over dctx and ctx inside npm_mha_core_bwd.  Same results either way."""
    from oracle import np_oracle as O
    D = npm.device
    rng = np.random.default_rng(4)
    x = rng.standard_normal([2, 8, 256]).astype(np.float32)
    dy = rng.standard_normal([2, 8, 256]).astype(np.float32)
    outs = {}
    for rowdot in (True, False):
        monkeypatch.setattr(D, 'ATTN_ROWDOT', rowdot)
        np.random.seed(0)
        layer = npm.layers.MultiHeadAttention(num_heads=2)
        layer(x)
        for n in ('_wq', '_wk', '_wv', '_wo'):
            getattr(layer, n).set(np.asarray(getattr(layer, n)) / np.float32(16.0))
        out = layer(x)
        assert layer._core and layer._key_dim == 128
        import hostsim
        sim = npm._C._LIB
        seen = []
        monkeypatch.setattr(sim, 'npm_mha_core_bwd',
                            lambda cref, inner=type(sim).npm_mha_core_bwd.__get__(sim), seen=seen: (seen.append(bool(hostsim._deref(cref).neg_delta)), inner(cref))[1])
        grads = [np.asarray(g) for g in layer(dy, backprop=True, learning_rate=0.01)]
        assert seen == [rowdot]
        outs[rowdot] = [np.asarray(out)] + grads + [np.asarray(getattr(layer, n)).copy() for n in ('_wq', '_wo', '_bq', '_bo')]
    for a, b in zip(outs[True], outs[False]):
        np.testing.assert_array_equal(a, b)          # (the simulator's backward does not depend on who computed the row terms)
