"""Data-parallel gradient exchange: batch sharding across the 8 MI355X of a node.

The reference is single-process; this module is the build's one
exchange step (section 8e): between a layer's gradient computation and its
``optimizer_.update`` calls (reference layers/mlp.py:38-39, normalizations.py:73-74,
attentions.py:190-197) the parameter gradients are all-reduced over RCCL/xGMI.

Design
* one process per GPU, described by RANK / LOCAL_RANK / WORLD_SIZE in the environment -- set by
  the self-launcher (np_modeling_amd/launch.py: fresh child processes, no third-party runtime) or
  by any external one-process-per-GPU launcher; the 128-byte RCCL id travels from rank 0 to
  the others through a file of this node (:func:`rendezvous_path`);
* gradients of one backward are carved out of ONE flat device bucket
  (:class:`GradScope`), so the exchange is a few large all-reduces issued on a separate
  communication stream as soon as each sub-layer's gradients exist (overlapping the rest
  of backward) instead of 16 latency-bound calls;
* every parameter update of the scope is applied after the exchange.  That is legal for
  composites because every dx is computed from pre-update weights (reference
  layers/transformer.py:61-92), so deferring the updates does not change results;
* reduction is AVG by default: with the reference's unchanged ``MSELoss`` (divides by the
  LOCAL ``y.size``, loss.py:25,29) the average of per-rank gradients equals the gradient
  of the global-batch loss.
"""

from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence, Tuple

from np_modeling_amd import _C
from np_modeling_amd import device as D

SUM, AVG, MAX = 0, 1, 2


class Communicator:
    """What a GradScope needs from a transport (RCCL here; tests plug gloo)."""

    rank = 0
    world_size = 1

    @property
    def active(self) -> bool:
        """Whether gradients have to be exchanged at all."""
        return self.world_size > 1

    def allreduce_async(self, flat: 'D.DeviceArray', op: int) -> None:
        raise NotImplementedError

    def wait(self) -> None:
        raise NotImplementedError

    def barrier(self) -> None:
        raise NotImplementedError

    def allreduce_scalar(self, value: float, op: int) -> float:
        raise NotImplementedError

    def broadcast(self, flat: 'D.DeviceArray', root: int = 0) -> None:
        raise NotImplementedError

    def close(self) -> None:
        pass


class RcclCommunicator(Communicator):
    """RCCL through libnpm_rccl.so (include/npm_comm.h)."""

    active = True      # also with one rank (NPM_FORCE_RCCL=1): exercises the whole exchange path

    def __init__(self, rank: int, world_size: int, unique_id: bytes):
        self._lib = _C.comm_lib()
        self.rank, self.world_size = rank, world_size
        stream = _C.lib().npm_stream()
        _C.check_comm(self._lib.npm_comm_init(unique_id, rank, world_size, stream), 'npm_comm_init')

    @staticmethod
    def library_path() -> str:
        """The librccl.so this process is bound to (include/npm_comm.h npm_comm_library_path)."""
        buf = C.create_string_buffer(1024)
        _C.check_comm(_C.comm_lib().npm_comm_library_path(buf, 1024), 'npm_comm_library_path')
        return buf.value.decode()

    @staticmethod
    def new_unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        _C.check_comm(_C.comm_lib().npm_comm_unique_id(buf), 'npm_comm_unique_id')
        return buf.raw

    def allreduce_async(self, flat, op):
        _C.check_comm(self._lib.npm_comm_allreduce_f32(flat.ptr, flat.size, op), 'npm_comm_allreduce_f32')

    def wait(self):
        _C.check_comm(self._lib.npm_comm_wait(), 'npm_comm_wait')

    def barrier(self):
        _C.check_comm(self._lib.npm_comm_barrier(), 'npm_comm_barrier')

    def allreduce_scalar(self, value, op):
        v = C.c_double(float(value))
        _C.check_comm(self._lib.npm_comm_allreduce_host_f64(C.byref(v), op), 'npm_comm_allreduce_host_f64')
        return float(v.value)

    def broadcast(self, flat, root=0):
        _C.check_comm(self._lib.npm_comm_broadcast_f32(flat.ptr, flat.size, root), 'npm_comm_broadcast_f32')
        self.wait()

    def stats_enable(self, on: bool = True) -> None:
        _C.check_comm(self._lib.npm_comm_stats_enable(int(on)), 'npm_comm_stats_enable')

    def stats(self) -> dict:
        """Exchange statistics since the previous call (include/npm_comm.h npm_comm_stats; synchronises)."""
        out = _C.npm_comm_exchange_stats()
        _C.check_comm(self._lib.npm_comm_stats(C.byref(out)), 'npm_comm_stats')
        return {'bytes': int(out.bytes), 'allreduce_calls': out.allreduce_calls, 'waits': out.waits,
                'allreduce_ms': out.allreduce_ms, 'exposed_ms': out.exposed_ms,
                'last_allreduce_ms': out.last_allreduce_ms, 'dropped': out.dropped}

    def close(self):
        self._lib.npm_comm_destroy()


_COMM: Optional[Communicator] = None
_REDUCE_OP = AVG
PLACEMENT: dict = {'bound': False, 'reason': 'single process'}      # what init() did about CPU affinity (launch.bind_to_gpu_cpus)


def _job_id() -> str:
    """A per-job identifier an external launcher or scheduler exports (every rank of the job sees the same one, a later
    job another): torch.distributed.run's ``--rdzv-id`` (its default 'none' is not one) or the SLURM job id."""
    for var in ('TORCHELASTIC_RUN_ID', 'SLURM_JOB_ID'):
        value = os.environ.get(var, '')
        if value and value.lower() != 'none':
            return f'{var}={value}'
    return ''


def _launch_token() -> str:
    """Identifies ONE launch of the ranks of a job, so that a rendezvous file left behind by another launch (a crashed job at a
    reused path) is never taken for this launch's.  In order:

    * ``NPM_LAUNCH_TOKEN`` -- the self-launcher (np_modeling_amd/launch.py) exports a fresh random one per launch; any other
      launcher may set its own;
    * an explicitly named ``NPM_RENDEZVOUS_FILE`` with a job id in the environment (torch.distributed.run's run id, the SLURM
      job id): job id + ``MASTER_ADDR:MASTER_PORT`` + the agent's restart count -- the same for every rank of the job whoever
      its parent process is, different for a later job;
    * an explicitly named file WITHOUT either, shared by ranks that are not known to be siblings on this node (``LOCAL_WORLD_SIZE``
      unset or ``!= WORLD_SIZE``): an immediate error -- nothing in the environment tells this launch from an earlier one with the same address (round 4 used the
      address alone: a stale file then passed for a fresh one and ranks other than 0 could hand a dead id to ncclCommInitRank);
    * otherwise the ranks of a node are children of one launcher process (torch.distributed.run's agent, a shell): its pid
      plus its start time (field 22 of /proc/<pid>/stat; pids are recycled, start times are not) plus the restart count."""
    explicit = os.environ.get('NPM_LAUNCH_TOKEN')
    if explicit:
        return explicit
    restart = os.environ.get('TORCHELASTIC_RESTART_COUNT', '0')
    if os.environ.get('NPM_RENDEZVOUS_FILE'):
        job = _job_id()
        if job:
            return f'{job}-{os.environ.get("MASTER_ADDR", "")}:{os.environ.get("MASTER_PORT", "")}-{restart}'
        world, local_world = os.environ.get('WORLD_SIZE', '1'), os.environ.get('LOCAL_WORLD_SIZE')
        if int(world) > 1 and local_world != world:
            # LOCAL_WORLD_SIZE absent: nothing says the ranks are siblings (separate shells, several nodes) -- their parent
            # pids would differ and every rank but 0 would poll until the timeout instead of failing here
            raise _C.NpmError('NPM_RENDEZVOUS_FILE is shared by ranks that are not known to be children of one launcher (WORLD_SIZE='
                              f'{world}, LOCAL_WORLD_SIZE={local_world}) and nothing identifies this launch: set NPM_LAUNCH_TOKEN '
                              'to a value that is new for every job (or run under a launcher that exports TORCHELASTIC_RUN_ID '
                              'or SLURM_JOB_ID); MASTER_ADDR:MASTER_PORT alone cannot tell a stale file from a fresh one')
    ppid = os.getppid()
    try:
        with open(f'/proc/{ppid}/stat') as f:
            started = f.read().rsplit(')', 1)[1].split()[19]
    except (OSError, IndexError):
        started = '0'
    return f'{ppid}-{started}-{restart}'


def _private_dir() -> str:
    """A directory only this user can write: ``$TMPDIR/npm-<uid>`` with mode 0700, owned by us (anything else at that
    name -- another user's directory, a symlink -- is refused: a shared /tmp must not let someone else plant an id)."""
    import stat
    import tempfile
    path = os.path.join(tempfile.gettempdir(), f'npm-{os.getuid()}')
    try:
        os.mkdir(path, 0o700)
    except FileExistsError:
        pass
    st = os.lstat(path)
    if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise _C.NpmError(f'{path} exists and is not a private directory of uid {os.getuid()} (mode 0700): remove it, '
                          'or name the rendezvous file with NPM_RENDEZVOUS_FILE')
    return path


def rendezvous_path() -> str:
    """Where rank 0 leaves the RCCL id for the other ranks of this node.  ``NPM_RENDEZVOUS_FILE`` (set by the
    self-launcher) names it outright; under an external launcher it lives in a per-user 0700 directory and is derived
    from the master port and the launch token, so concurrent or earlier jobs never share a file."""
    explicit = os.environ.get('NPM_RENDEZVOUS_FILE')
    if explicit:
        return explicit
    return os.path.join(_private_dir(), 'rccl_id.' + os.environ.get('MASTER_PORT', '0') + '.' + _launch_token())


_ID_MAGIC = b'NPMRCCL1'


def _pack_id(uid: bytes) -> bytes:
    """magic + 2-byte token length + launch token + the 128-byte id.  The token makes a file left behind by another
    launch (a reused NPM_RENDEZVOUS_FILE, a crashed job) recognisable: readers skip it and keep polling instead of
    handing a dead bootstrap address to ncclCommInitRank, which would hang."""
    token = _launch_token().encode()
    return _ID_MAGIC + len(token).to_bytes(2, 'little') + token + uid


def _unpack_id(blob: bytes) -> Optional[bytes]:
    """The id carried by ``blob`` if it is complete and belongs to THIS launch, else None."""
    head = len(_ID_MAGIC) + 2
    if len(blob) < head or not blob.startswith(_ID_MAGIC):
        return None
    n = int.from_bytes(blob[len(_ID_MAGIC):head], 'little')
    if len(blob) != head + n + 128 or blob[head:head + n] != _launch_token().encode():
        return None
    return blob[head + n:]


def _exchange_unique_id(rank: int, world_size: int, timeout: float = 300.0) -> bytes:
    """Carry rank 0's RCCL id to the other ranks of the node through a file: rank 0 writes it under a temporary
    name and renames it into place (readers see all of it or nothing); the others poll, and accept only a file
    that this user wrote for this launch.  No third-party runtime is involved -- the product path imports neither
    torch nor an MPI."""
    import time
    local_world = os.environ.get('LOCAL_WORLD_SIZE')
    if local_world and int(local_world) != world_size and not os.environ.get('NPM_RENDEZVOUS_FILE'):
        raise _C.NpmError(f'WORLD_SIZE={world_size} but LOCAL_WORLD_SIZE={local_world}: the RCCL id travels through a '
                          'file of ONE node (SURVEY.md 8e: the 8 GPUs of a node); for several nodes put '
                          'NPM_RENDEZVOUS_FILE on a file system all of them share and give every rank the same '
                          'NPM_LAUNCH_TOKEN, new for every job (or a TORCHELASTIC_RUN_ID / SLURM_JOB_ID): the file carries '
                          'that token and ranks accept only a file with theirs')
    path = rendezvous_path()
    if rank == 0:
        uid = RcclCommunicator.new_unique_id()
        assert len(uid) == 128
        tmp = f'{path}.{os.getpid()}.tmp'
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
        with os.fdopen(fd, 'wb') as f:
            f.write(_pack_id(uid))
        os.replace(tmp, path)                      # also replaces whatever an earlier launch left at this name
        return uid
    deadline = time.monotonic() + timeout
    seen_foreign = False
    while True:
        try:
            with open(path, 'rb') as f:
                if os.fstat(f.fileno()).st_uid != os.getuid():
                    raise _C.NpmError(f'rank {rank}: {path} belongs to uid {os.fstat(f.fileno()).st_uid}, not to this '
                                      'user: refusing an RCCL id somebody else wrote')
                uid = _unpack_id(f.read())
            if uid is not None:
                return uid
            seen_foreign = True                    # incomplete, or written by another launch: rank 0 will replace it
        except FileNotFoundError:
            pass
        if time.monotonic() > deadline:
            what = 'only a file of another launch' if seen_foreign else 'no file'
            raise _C.NpmError(f'rank {rank}: no RCCL id from rank 0 at {path} after {timeout:.0f} s ({what}); '
                              f'launch token {_launch_token()!r}. Ranks must be children of one launcher process on one '
                              'node, or share an explicit NPM_RENDEZVOUS_FILE together with the same NPM_LAUNCH_TOKEN (or job id)')
        time.sleep(0.01)


def init(reduce: str = 'avg') -> Communicator:
    """Join the data-parallel group described by RANK / WORLD_SIZE (idempotent)."""
    global _COMM, _REDUCE_OP
    _REDUCE_OP = {'avg': AVG, 'sum': SUM}[reduce]
    if _COMM is not None:
        return _COMM
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world <= 1 and os.environ.get('NPM_FORCE_RCCL', '0') != '1':
        _COMM = Communicator()
        return _COMM
    global PLACEMENT
    _C.ipc_env_for_multi_rank()                          # before the device library initialises the runtime (external launchers)
    if world > 1:
        # one rank of several on this node: die with the launcher, and run on the CPUs of the GPU's own NUMA node --
        # both before anything touches the device (the driver's threads inherit the mask)
        from np_modeling_amd import launch
        launch.die_with_launcher()
        if _C._LIB is None:
            PLACEMENT = launch.bind_to_gpu_cpus(int(os.environ.get('LOCAL_RANK', str(rank))))
        else:
            PLACEMENT = {'bound': False, 'reason': 'the device library was loaded before parallel.init()'}
    _C.lib()                                             # bind this process to cuda:LOCAL_RANK first
    uid = _exchange_unique_id(rank, world) if world > 1 else RcclCommunicator.new_unique_id()
    _COMM = RcclCommunicator(rank, world, uid)
    if world > 1 and rank == 0:
        # ncclCommInitRank is collective: every rank has read the id by now.  Explicit paths are removed too, so a
        # reused NPM_RENDEZVOUS_FILE never hands a later launch this launch's id.
        try:
            os.unlink(rendezvous_path())
        except OSError:
            pass
    return _COMM


def set_communicator(comm: Optional[Communicator], reduce: str = 'avg') -> None:
    """Install a transport explicitly (tests use a gloo transport on host memory)."""
    global _COMM, _REDUCE_OP
    _COMM = comm
    _REDUCE_OP = {'avg': AVG, 'sum': SUM}[reduce]


def communicator() -> Communicator:
    """The active transport.  Under a one-process-per-GPU launcher (WORLD_SIZE > 1 in the environment) the
    RCCL group is joined on first use, so a training script written for one GPU synchronises its gradients
    without changes; NPM_AUTO_PARALLEL=0 disables that."""
    if _COMM is None:
        if int(os.environ.get('WORLD_SIZE', '1')) > 1 and os.environ.get('NPM_AUTO_PARALLEL', '1') != '0':
            return init()
        return Communicator()
    return _COMM


def world_size() -> int:
    return communicator().world_size


def rank() -> int:
    return communicator().rank


def shutdown() -> None:
    global _COMM
    if _COMM is not None:
        _COMM.close()
    _COMM = None


def shard(array, axis: int = 0):
    """This rank's contiguous slice of a host batch along ``axis`` (axis 0 of qkv / dy)."""
    import numpy as np
    w, r = world_size(), rank()
    n = array.shape[axis]
    assert n % w == 0, f'batch {n} is not divisible by world_size {w}'
    per = n // w
    index = [slice(None)] * array.ndim
    index[axis] = slice(r * per, (r + 1) * per)
    return np.ascontiguousarray(array[tuple(index)])


def parameters(layer) -> List['D.DeviceArray']:
    """Every device-resident parameter of a (possibly composite) layer, in a deterministic order."""
    from np_modeling_amd.layers.layer import Layer
    names = ('_w', '_b', '_wq', '_wk', '_wv', '_wo', '_bq', '_bk', '_bv', '_bo', '_gamma', '_beta')
    found, stack, seen = [], [layer], set()
    while stack:
        obj = stack.pop()
        if id(obj) in seen:
            continue
        seen.add(id(obj))
        for key in sorted(vars(obj)):
            value = getattr(obj, key)
            if key in names and value is not None:
                found.append(obj._param(key))
            elif isinstance(value, Layer):
                stack.append(value)
    return found


def sync_parameters(layers: Sequence, root: int = 0) -> None:
    """Broadcast rank ``root``'s parameters to every rank (call once after the first forward, when ranks
    did not seed NumPy identically).  No-op for one process."""
    comm = communicator()
    if not comm.active:
        return
    for layer in layers:
        for param in parameters(layer):
            comm.broadcast(param.reshape(-1), root)


# --------------------------------------------------------------------------------------------
class GradScope:
    """Collects the gradients and deferred parameter updates of one ``backward``.

    The outermost scope of a backward owns the flat bucket; nested scopes (sub-layers of a
    composite) delegate to it.  ``take`` hands out gradient storage, ``defer`` queues an
    ``optimizer_.update``, ``flush`` starts the all-reduce of everything taken so far, and
    leaving the outermost scope waits for the exchange and applies the updates in order.

    With a ``device.ParamArena`` (the layer's parameters back to back, in the order backward produces their
    gradients) the bucket MIRRORS the arena: ``take(shape, owner=(layer, attribute))`` returns the slice at the
    parameter's own offset, so parameter range and gradient range line up and the deferred updates run as one launch
    (``device.coalesced_updates``).  Then there is a bucket with one rank too.
    """

    _active: Optional['GradScope'] = None

    def __init__(self, numel_hint: int = 0, arena: Optional['D.ParamArena'] = None):
        self._hint = int(numel_hint)
        self._arena = arena if D.COALESCE_UPDATES else None
        self._outer: Optional[GradScope] = None
        self._bucket: Optional[D.DeviceArray] = None
        self._offset = 0               # end of the sequentially carved part
        self._spans: List[Tuple[int, int]] = []      # slices handed out and not yet flushed (start, end), any order
        self._flushed = 0
        self._started = False          # an all-reduce of this scope has been issued
        self._loose: List[D.DeviceArray] = []
        self._updates: List[Tuple[object, object, str, object]] = []
        self.collectives = 0           # all-reduce calls this scope issued (tests: one bucket, a handful of flushes)
        self.update_launches = None    # optimizer kernels its deferred updates ran as (None: not coalesced)

    # -- context management ----------------------------------------------------------
    def __enter__(self) -> 'GradScope':
        self._outer = GradScope._active
        if self._outer is None:
            GradScope._active = self
            if self._arena is not None and not self._arena.live():
                self._arena = None           # every parameter has moved on (into a composite's arena, or rebound): nothing to mirror
            if self._arena is not None and self._arena.size > 0:
                self._offset = self._arena.size                      # sequential takes (parameters outside the arena) go behind it
                self._bucket = D.empty([self._arena.size + max(self._hint - self._arena.size, 0) + 64])
            elif communicator().active and self._hint > 0:
                self._bucket = D.empty([self._hint])
        return self

    def __exit__(self, exc_type, exc, tb) -> bool:
        if self._outer is not None:
            return False
        GradScope._active = None
        if exc_type is None:
            self._finish()
        else:
            # all-reduces started by flush() may still be writing the bucket on the communication stream, and the
            # pool is stream-ordered for the COMPUTE stream only: drain both before the storage goes back to it
            comm = communicator()
            if comm.active and self._started:
                try:
                    comm.wait()
                    D.synchronize()
                except Exception:          # the original exception is the one to report
                    pass
            self._updates = []
            self._bucket = None
        return False

    @property
    def root(self) -> 'GradScope':
        return self if self._outer is None else self._outer.root

    # -- gradient storage ------------------------------------------------------------------
    def take(self, shape: Sequence[int], owner: Optional[Tuple[object, str]] = None) -> D.DeviceArray:
        """Storage for a gradient of ``shape``.  ``owner`` = (layer, attribute) of the parameter it is the gradient of
        (the FIRST of several adjacent ones when ``shape`` covers them all, like the packed wq / wk / wv): inside an
        arena-backed scope the slice then sits where the parameter sits in the arena."""
        root = self.root
        n = D._prod(shape)
        if root._bucket is not None:
            if owner is not None and root._arena is not None:
                at = root._arena.offset_of(owner[0], owner[1], n)
                if at is not None:
                    root._spans.append((at, at + n))
                    return root._bucket.flat_view(at, shape)
            start = (root._offset + 3) // 4 * 4                # 16-byte aligned slices
            if start + n <= root._bucket.size:
                root._offset = start + n
                root._spans.append((start, start + n))
                return root._bucket.flat_view(start, shape)
        g = D.empty(shape)
        if communicator().active:
            root._loose.append(g)
        return g

    def defer(self, optimizer_, obj, attribute: str, grad) -> None:
        self.root._updates.append((optimizer_, obj, attribute, grad))

    def _ready_ranges(self) -> List[Tuple[int, int]]:
        """The slices handed out since the last flush, joined where they touch (alignment gaps of up to 3 floats are
        part of the bucket: reducing them along costs nothing).  Every slice a caller has TAKEN has been written by the
        kernels enqueued before this call (stream order), so each range may go out."""
        spans, self._spans = sorted(self._spans), []
        ranges: List[List[int]] = []
        for start, end in spans:
            if ranges and start <= ranges[-1][1] + 3:
                ranges[-1][1] = max(ranges[-1][1], end)
            else:
                if self._flushed <= start <= self._flushed + 3:
                    start = self._flushed        # the padding behind the previous flush travels with this one: one unbroken range
                ranges.append([start, end])
        if ranges:
            self._flushed = max(self._flushed, ranges[-1][1])
        return [(a, b) for a, b in ranges if b > a]

    def flush(self) -> None:
        """Start exchanging every gradient produced so far (asynchronous)."""
        root = self.root
        comm = communicator()
        if not comm.active:
            root._spans = []
            return
        if root._bucket is not None:
            for begin, end in root._ready_ranges():
                root._started = True
                root.collectives += 1
                comm.allreduce_async(root._bucket.flat_view(begin, [end - begin]), _REDUCE_OP)
        for g in root._loose:
            root._started = True
            root.collectives += 1
            comm.allreduce_async(g.reshape(-1), _REDUCE_OP)
        root._loose = []

    def _finish(self) -> None:
        comm = communicator()
        if comm.active:
            self.flush()
            comm.wait()
        updates, self._updates = self._updates, []
        with D.coalesced_updates() as queue:
            for optimizer_, obj, attribute, grad in updates:
                optimizer_.update(obj, attribute, grad)
        if queue is not None:
            self.update_launches = queue.launches
        GradScope.last = {'collectives': self.collectives, 'update_launches': self.update_launches,
                          'updates': len(updates), 'arena': self._arena is not None}
        self._bucket = None
        self._arena = None


GradScope.last = None          # numbers of the most recent outermost scope that finished (diagnostics / tests); no buffers


def grad_scope(numel_hint: int = 0, arena: Optional['D.ParamArena'] = None) -> GradScope:
    return GradScope(numel_hint, arena)
