"""GPU: the RCCL shim (libnpm_rccl.so) with a one-rank communicator -- unique id, init, stream
ordering of an in-place all-reduce against the compute stream, broadcast, barrier, host scalar
reduction -- and a full encoder step with the exchange path active (flat bucket + deferred
updates) giving the same parameters as the plain single-GPU path."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def comm():
    import np_modeling_amd  # noqa: F401
    from np_modeling_amd import _C, parallel
    _C.lib()
    c = parallel.RcclCommunicator(0, 1, parallel.RcclCommunicator.new_unique_id())
    yield c
    parallel.set_communicator(None)
    c.close()


def test_rccl_single_rank_collectives(comm):
    from np_modeling_amd import device as D, parallel
    x = np.random.default_rng(0).standard_normal(1 << 20).astype(np.float32)
    d = D.from_host(x)
    d *= 2.0                                        # compute-stream work the collective must wait for
    comm.allreduce_async(d, parallel.AVG)
    comm.wait()
    d += 1.0                                        # compute-stream work that must wait for the collective
    np.testing.assert_array_equal(d.numpy(), x * 2 + 1)
    comm.allreduce_async(d.flat_view(5, [1000]), parallel.SUM)
    comm.broadcast(d)
    comm.barrier()
    assert comm.allreduce_scalar(3.5, parallel.MAX) == 3.5
    np.testing.assert_array_equal(d.numpy(), x * 2 + 1)


def test_encoder_step_with_exchange_path_active(comm):
    import np_modeling_amd as npm
    from np_modeling_amd import parallel

    def run(active):
        parallel.set_communicator(comm if active else None)
        np.random.seed(0)
        enc = npm.layers.TransformerEncoder(num_heads=4, hidden_units=64, norm_first=True)
        x = np.random.normal(size=[4, 16, 32]).astype(np.float32)
        dy = np.random.normal(size=[4, 16, 32]).astype(np.float32)
        enc(x)
        dx = np.asarray(enc(dy, backprop=True, learning_rate=0.01))
        return dx, np.asarray(enc._dense2.w), np.asarray(enc._self_attention._wq), np.asarray(enc._norm1._beta)

    for a, b in zip(run(False), run(True)):
        np.testing.assert_array_equal(a, b)


def test_sync_parameters_walks_composites(comm):
    import np_modeling_amd as npm
    from np_modeling_amd import parallel
    parallel.set_communicator(comm)
    np.random.seed(1)
    enc = npm.layers.TransformerEncoder(num_heads=2, hidden_units=16, norm_first=False)
    enc(np.random.normal(size=[2, 4, 8]).astype(np.float32))
    params = parallel.parameters(enc)
    assert len(params) == 16 and sum(p.size for p in params) == 4 * 64 + 3 * 8 + 8 + 4 * 8 + 8 * 16 + 16 + 16 * 8 + 8
    before = [np.asarray(p).copy() for p in params]
    parallel.sync_parameters([enc])                 # one rank: broadcast is the identity
    for a, p in zip(before, params):
        np.testing.assert_array_equal(a, np.asarray(p))
    parallel.set_communicator(None)


def test_rccl_init_keeps_stdout_clean():
    """RCCL prints a version banner on stdout when it initialises; a program whose stdout is data (bench.py's single
    JSON line) must not see it: the shim points stdout at stderr while RCCL initialises."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from np_modeling_amd import _C, parallel\n"
            "_C.lib()\n"
            "c = parallel.RcclCommunicator(0, 1, parallel.RcclCommunicator.new_unique_id())\n"
            "c.barrier(); c.close()\n"
            "print('only-this-line')\n" % root)
    out = subprocess.run([sys.executable, '-c', code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.strip().splitlines() == ['only-this-line'], out.stdout


def test_self_launched_rank_runs_the_exchange_path():
    """`python -m np_modeling_amd.launch --gpus 1 bench.py ...` with NPM_FORCE_RCCL=1: the spawned-rank path end to
    end on one GPU -- a fresh child process, an RCCL communicator bound to /opt/rocm's librccl (no torch in the
    process), the flat-bucket all-reduce inside backward, and ONE JSON line on stdout."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NPM_FORCE_RCCL='1')
    for key in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'NPM_RENDEZVOUS_FILE'):
        env.pop(key, None)
    cmd = [sys.executable, '-m', 'np_modeling_amd.launch', '--gpus', '1', os.path.join(root, 'bench.py'), '--gpus', '1',
           '--steps', '2', '--warmup', '1', '--batch', '4', '--seq', '64', '--features', '128', '--heads', '4',
           '--hidden', '256', '--no-cpu-baseline', '--no-alt-math', '--no-configs']
    out = subprocess.run(cmd, env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = out.stdout.strip().splitlines()
    assert len(lines) == 1, out.stdout
    line = json.loads(lines[0])
    assert line['n_gpus'] == 1 and line['value'] > 0
    assert line['exchange']['library'].startswith('/opt/rocm'), line['exchange']
    assert line['exchange']['torch_imported'] is False
    assert 'self-launched' in line['exchange']['launcher']


def test_exchange_statistics(comm):
    """npm_comm_stats: bytes / calls / waits counted, the times non-negative and reset by the read."""
    from np_modeling_amd import device as D, parallel
    comm.stats_enable(True)
    comm.stats()
    d = D.from_host(np.ones(1 << 18, dtype=np.float32))
    comm.allreduce_async(d, parallel.AVG)
    comm.allreduce_async(d.flat_view(0, [1024]), parallel.AVG)
    comm.wait()
    got = comm.stats()
    assert got['bytes'] == 4 * ((1 << 18) + 1024) and got['allreduce_calls'] == 2 and got['waits'] == 1
    assert got['allreduce_ms'] > 0 and got['exposed_ms'] >= 0
    again = comm.stats()
    assert again['bytes'] == 0 and again['allreduce_calls'] == 0 and again['allreduce_ms'] == 0
    comm.stats_enable(False)
    comm.allreduce_async(d, parallel.AVG)
    comm.wait()
    assert comm.stats()['allreduce_calls'] == 0


def test_rank_under_a_per_rank_device_mask():
    """A launcher that shows each rank ONE GPU (HIP_VISIBLE_DEVICES=<its gpu>) still sets LOCAL_RANK=3: the rank must
    bind to device 0, the only one it can see -- `bench.py`'s exchange object reports what it bound to."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NPM_FORCE_RCCL='1', HIP_VISIBLE_DEVICES='0', LOCAL_RANK='3', RANK='0', WORLD_SIZE='1')
    env.pop('NPM_DEVICE', None)
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '1', '--steps', '2', '--warmup', '1', '--batch', '4',
           '--seq', '64', '--features', '128', '--heads', '4', '--hidden', '256', '--no-cpu-baseline', '--no-alt-math',
           '--no-configs']
    out = subprocess.run(cmd, env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads(out.stdout.strip())
    assert line['exchange']['device'] == 0 and line['exchange']['visible_devices'] == 1
    assert line['exchange']['flushes_per_step'] >= 3 and line['exchange']['bytes_per_step'] > 0
    assert line['exchange']['allreduce_ms'] > 0 and line['exchange']['exposed_ms'] >= 0
