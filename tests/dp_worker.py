"""Worker of tests/test_host_logic.py::test_data_parallel_equivalence_gloo (CPU, gloo, world_size 2).

Each rank runs the PRODUCT's host path (layers, GradScope bucketing, deferred updates) on the
host simulator of the C ABI, with a gloo transport standing in for RCCL, on its shard of a global
batch; rank 0 also runs the whole batch alone and checks that the parameters after one
SGD step are the same (loss normalised by the LOCAL size + AVG reduction)."""

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

os.environ['NPM_AUTO_PARALLEL'] = '0'          # this worker installs a gloo transport explicitly

import torch                      # noqa: E402
import torch.distributed as dist  # noqa: E402

import hostsim                    # noqa: E402
import np_modeling_amd as npm     # noqa: E402
from np_modeling_amd import parallel  # noqa: E402


class GlooCommunicator(parallel.Communicator):
    def __init__(self):
        self.rank, self.world_size = dist.get_rank(), dist.get_world_size()
        self.calls = []

    def allreduce_async(self, flat, op):
        view = torch.from_numpy(hostsim._vec(flat.ptr, flat.size))      # shares the simulator's memory
        dist.all_reduce(view, op=dist.ReduceOp.MAX if op == parallel.MAX else dist.ReduceOp.SUM)
        if op == parallel.AVG:
            view /= self.world_size
        self.calls.append(flat.size)

    def wait(self):
        pass

    def barrier(self):
        dist.barrier()

    def allreduce_scalar(self, value, op):
        t = torch.tensor([float(value)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX if op == parallel.MAX else dist.ReduceOp.SUM)
        return float(t[0]) / (self.world_size if op == parallel.AVG else 1)


def build(kind, seed=0):
    np.random.seed(seed)
    if kind == 'encoder_pre':
        return [npm.layers.TransformerEncoder(num_heads=2, hidden_units=24, norm_first=True)]
    if kind == 'encoder_post':
        return [npm.layers.TransformerEncoder(num_heads=2, hidden_units=24, norm_first=False)]
    return [npm.layers.Dense(units=12), npm.layers.Dense(units=6)]


def params_of(layers):
    out = []
    for layer in layers:
        stack = [layer]
        while stack:
            obj = stack.pop()
            for key in sorted(vars(obj)):
                val = getattr(obj, key)
                if isinstance(val, npm.DeviceArray) and key in ('_w', '_b', '_wq', '_wk', '_wv', '_wo', '_bq', '_bk',
                                                                '_bv', '_bo', '_gamma', '_beta'):
                    out.append((type(obj).__name__ + key, np.asarray(val).copy()))
                elif isinstance(val, npm.layers.Layer):
                    stack.append(val)
    return out


def run(kind, x, t, comm):
    parallel.set_communicator(comm, 'avg')
    layers = build(kind)
    trainer = npm.train.Trainer(layers)
    sys.stdout = open(os.devnull, 'w')
    try:
        trainer.train(inputs=x, targets=t, steps=2, optimizer_=npm.optimizer.SGDOptimizer(0.05))
    finally:
        sys.stdout = sys.__stdout__
    return params_of(layers)


def run_decoder(norm_first, q, kv, dy, comm, scale):
    """Two SGD steps of a TransformerDecoder on (q, kv) with the upstream gradient ``scale * dy`` (AVG of the ranks' gradients of
    ``world * dy_shard`` = the whole batch's gradient of dy).  Returns the parameters and the collectives of each backward."""
    parallel.set_communicator(comm, 'avg')
    np.random.seed(0)
    dec = npm.layers.TransformerDecoder(num_heads=2, hidden_units=24, norm_first=norm_first)
    opt = npm.optimizer.SGDOptimizer(0.05)
    per_backward = []
    for _ in range(2):
        dec(q, kv)
        before = len(getattr(comm, 'calls', []))
        dec(npm.device.from_host(dy * np.float32(scale)), backprop=True, optimizer_=opt)
        per_backward.append(len(getattr(comm, 'calls', [])) - before)
    return params_of([dec]), per_backward


def main():
    dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{os.environ["MASTER_PORT"]}',
                            rank=int(os.environ['RANK']), world_size=int(os.environ['WORLD_SIZE']))
    hostsim.install()
    rank, world = dist.get_rank(), dist.get_world_size()
    failures = []
    for kind in ('mlp', 'encoder_pre', 'encoder_post'):
        rng = np.random.default_rng(7)
        shape = [4 * world, 10] if kind == 'mlp' else [2 * world, 6, 8]
        tshape = [4 * world, 6] if kind == 'mlp' else shape
        x = rng.standard_normal(shape).astype(np.float32)
        t = rng.standard_normal(tshape).astype(np.float32)
        comm = GlooCommunicator()
        parallel.set_communicator(comm)
        got = run(kind, parallel.shard(x), parallel.shard(t), comm)
        assert len(comm.calls) > 0
        if rank == 0:
            want = run(kind, x, t, parallel.Communicator())          # whole batch, one process
            for (name, a), (name2, b) in zip(got, want):
                assert name == name2
                err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-12)
                if not err < 2e-6:
                    failures.append(f'{kind}:{name} rel err {err:.2e}')
            print(f'{kind}: {len(got)} params compared, all-reduce sizes {comm.calls[:4]}...', flush=True)
        # every rank must end with identical parameters
        for name, a in got:
            buf = torch.from_numpy(a.copy())
            dist.broadcast(buf, 0)
            if not np.array_equal(buf.numpy(), a):
                failures.append(f'{kind}:{name} differs between ranks')
    # TransformerDecoder (two inputs: not a Trainer layer).  Its 26 gradients travel in ONE bucket: a handful of collectives
    # per backward (feed-forward, its norm / dense1, cross-attention, self-attention, the rest), not one per gradient (round 4)
    for norm_first in (True, False):
        kind = 'decoder_pre' if norm_first else 'decoder_post'
        rng = np.random.default_rng(11)
        q = rng.standard_normal([2 * world, 6, 8]).astype(np.float32)
        kv = rng.standard_normal([2 * world, 9, 8]).astype(np.float32)
        dy = (rng.standard_normal([2 * world, 6, 8]) * 1e-3).astype(np.float32)      # an upstream gradient of the size MSE's is
        comm = GlooCommunicator()
        parallel.set_communicator(comm)
        got, per_backward = run_decoder(norm_first, parallel.shard(q), parallel.shard(kv), parallel.shard(dy), comm, world)
        if not all(1 <= n <= 6 for n in per_backward):
            failures.append(f'{kind}: {per_backward} collectives per backward (want at most 6: one bucket, a few flushes)')
        if rank == 0:
            want, _ = run_decoder(norm_first, q, kv, dy, parallel.Communicator(), 1)
            for (name, a), (name2, b) in zip(got, want):
                assert name == name2
                err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-12)
                if not err < 1e-5:            # fp32 sums over another partition of the batch, gradients of O(world) here
                    failures.append(f'{kind}:{name} rel err {err:.2e}')
            print(f'{kind}: {len(got)} params compared, {per_backward} collectives per backward, sizes {comm.calls[:5]}', flush=True)
        for name, a in got:
            buf = torch.from_numpy(a.copy())
            dist.broadcast(buf, 0)
            if not np.array_equal(buf.numpy(), a):
                failures.append(f'{kind}:{name} differs between ranks')
    dist.barrier()
    dist.destroy_process_group()
    if failures:
        print('\n'.join(failures))
        sys.exit(1)


if __name__ == '__main__':
    main()
