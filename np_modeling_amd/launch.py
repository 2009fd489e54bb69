"""Self-launcher: one fresh process per GPU of this node, no third-party runtime.

The reference is single-process; the batch-sharded path (section 8e) runs one process per
MI355X.  ``spawn_ranks(n, argv)`` starts ``n`` children ``argv`` with ``RANK`` / ``LOCAL_RANK`` / ``WORLD_SIZE``
set and a private rendezvous file for the RCCL id (``NPM_RENDEZVOUS_FILE``, read by
np_modeling_amd/parallel.py).  Rules it keeps:

* the parent never touches the GPU (no HIP call, no ``npm_*`` call): children are FRESH processes started before
  anything initialises a device -- a process that has initialised the GPU is never re-executed;
* the shared libraries are built (if missing) BEFORE the ranks start, so N ranks never race in ``make``;
* a rank that exits non-zero (or dies on a signal) ends the job: the others are terminated and the launcher
  returns that rank's code;
* rank 0's stdout is the launcher's stdout (bench.py prints its one JSON line there); all ranks share stderr;
* a launcher that is TERMINATED (SIGTERM / SIGHUP / SIGINT -- ``timeout -k 10 400 python bench.py --gpus N`` sends
  exactly that) takes its ranks with it: the signal is turned into an exception that runs the clean-up below, and
  every rank asks the kernel for a SIGTERM of its own should the launcher die without cleaning up (SIGKILL).

    python -m np_modeling_amd.launch --gpus 8 train_script.py --its-own --flags
"""

from __future__ import annotations

import os
import shutil
import signal
import subprocess
import sys
import tempfile
import time
from typing import Dict, List, Optional, Sequence


class _Terminated(Exception):
    def __init__(self, signum: int):
        super().__init__(signum)
        self.signum = signum


_HANDLED = (signal.SIGTERM, signal.SIGHUP)
_PRCTL = None           # libc's prctl, resolved ONCE in the parent: the forked child only calls it


def _resolve_prctl():
    global _PRCTL
    if _PRCTL is None:
        import ctypes
        try:
            _PRCTL = ctypes.CDLL(None, use_errno=True).prctl
        except (OSError, AttributeError):
            _PRCTL = False
    return _PRCTL


def die_with_parent() -> bool:
    """PR_SET_PDEATHSIG(SIGTERM): the kernel signals this process when the thread that started it exits, however it
    exits (the clean-up of a SIGKILLed launcher).  Called between fork and exec by the launcher (with a function that
    was bound before the fork: no import, no dlopen in the child) and once more by every rank at start-up
    (np_modeling_amd/parallel.py init), which also covers a failure of the first call."""
    prctl = _PRCTL if _PRCTL is not None else _resolve_prctl()
    return bool(prctl) and prctl(1, int(signal.SIGTERM), 0, 0, 0) == 0            # PR_SET_PDEATHSIG = 1


def die_with_launcher() -> bool:
    """What a rank does at start-up (np_modeling_amd/parallel.py init): arm PR_SET_PDEATHSIG again -- but only when THIS
    package's launcher started the rank (it exports its pid as NPM_LAUNCHER_PID, and the rank's parent is that process).  A
    rank started from a shell or a wrapper that exits on purpose (nohup, setsid, ``bash -c '... &'``) must not be killed
    when that parent goes away.  The call can also come too late (no signal is delivered for a parent that died before it):
    the parent is checked again afterwards and the rank ends itself.  NPM_DIE_WITH_PARENT=0 switches it off."""
    if os.environ.get('NPM_DIE_WITH_PARENT', '1') == '0':
        return False
    launcher = os.environ.get('NPM_LAUNCHER_PID', '')
    if not launcher.isdigit():
        return False
    if os.getppid() != int(launcher):
        # started through an intermediate process (its lifetime is not ours to tie to), or the launcher is gone already --
        # in which case the request armed between fork and exec (_child_setup) has delivered its signal
        return False
    armed = die_with_parent()
    if os.getppid() != int(launcher):               # it died between the check and the prctl: no signal will come
        os._exit(143)
    return armed


def _child_setup() -> None:
    """Between fork and exec.  The launcher blocks its handled signals around each spawn (a signal must find every
    started rank in its list); the block is inherited across exec, so the child lifts it again."""
    signal.pthread_sigmask(signal.SIG_UNBLOCK, _HANDLED)
    die_with_parent()


# ---- NUMA placement ---------------------------------------------------------------------------------------------
def _parse_cpulist(text: str) -> set:
    cpus = set()
    for part in text.strip().split(','):
        if not part:
            continue
        lo, _, hi = part.partition('-')
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_local_cpus(local_rank: int, sysfs: str = '/sys') -> Optional[set]:
    """The CPUs of the NUMA node the ``local_rank``-th visible GPU hangs off, from sysfs alone (no HIP call):
    the KFD topology lists the GPUs in the order HIP numbers them (nodes with simd_count > 0); ``location_id`` /
    ``domain`` give the PCI address, whose ``local_cpulist`` the kernel provides.  HIP_VISIBLE_DEVICES /
    ROCR_VISIBLE_DEVICES given as a plain list of indices are honoured.  None when anything is missing."""
    try:
        nodes_dir = os.path.join(sysfs, 'class', 'kfd', 'kfd', 'topology', 'nodes')
        gpus = []
        for name in sorted(os.listdir(nodes_dir), key=int):
            props = {}
            with open(os.path.join(nodes_dir, name, 'properties')) as f:
                for line in f:
                    key, _, value = line.strip().partition(' ')
                    props[key] = value
            if int(props.get('simd_count', '0')) > 0:
                gpus.append(props)
        for var in ('ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES'):
            mask = os.environ.get(var)
            if mask:
                gpus = [gpus[int(i)] for i in mask.split(',')]
        # np_modeling_amd/_C.py pick_device: NPM_DEVICE wins; a per-rank mask that leaves one device makes it device 0
        index = int(os.environ['NPM_DEVICE']) if os.environ.get('NPM_DEVICE') else (0 if len(gpus) == 1 else local_rank)
        props = gpus[index]
        loc, domain = int(props['location_id']), int(props.get('domain', '0'))
        bdf = f'{domain:04x}:{(loc >> 8) & 0xff:02x}:{(loc >> 3) & 0x1f:02x}.{loc & 7}'
        with open(os.path.join(sysfs, 'bus', 'pci', 'devices', bdf, 'local_cpulist')) as f:
            return _parse_cpulist(f.read()) or None
    except (OSError, ValueError, IndexError, KeyError):
        return None


def bind_to_gpu_cpus(local_rank: int, sysfs: str = '/sys') -> dict:
    """Restrict this process (before it starts any thread or touches the GPU) to the CPUs local to its GPU, so its host
    thread, the driver's helper threads and its pinned buffers stay on the socket the card hangs off.  Returns what was
    done, for bench.py's ``exchange`` object.  NPM_BIND_CPUS=0 switches it off."""
    info = {'bound': False, 'local_rank': local_rank}
    if os.environ.get('NPM_BIND_CPUS', '1') == '0' or not hasattr(os, 'sched_setaffinity'):
        info['reason'] = 'disabled'
        return info
    allowed = os.sched_getaffinity(0)
    local = gpu_local_cpus(local_rank, sysfs)
    if not local:
        info['reason'] = 'no NUMA information for this GPU in sysfs'
        return info
    cpus = local & allowed
    if not cpus or cpus == allowed:
        info['reason'] = 'the GPU-local CPUs are the whole affinity mask already' if cpus else 'the GPU-local CPUs are outside the affinity mask'
        info['cpus'] = len(allowed)
        return info
    os.sched_setaffinity(0, cpus)
    info.update(bound=True, cpus=len(cpus), first_cpu=min(cpus), last_cpu=max(cpus))
    return info


def new_launch_token() -> str:
    """A value no other launch has: every rank of one spawn_ranks() call gets the same one (NPM_LAUNCH_TOKEN), and the
    rendezvous file carries it (np_modeling_amd/parallel.py _pack_id)."""
    import uuid
    return uuid.uuid4().hex


def rank_environment(rank: int, world: int, rendezvous_file: str, base: Optional[Dict[str, str]] = None,
                     token: Optional[str] = None) -> Dict[str, str]:
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               NPM_RENDEZVOUS_FILE=rendezvous_file, NPM_LAUNCH_TOKEN=token or new_launch_token(),
               NPM_LAUNCHER_PID=str(os.getpid()))
    env.setdefault('MASTER_ADDR', '127.0.0.1')
    # RCCL maps its peers' buffers through HIP IPC handles; this pool's host driver implements the dmabuf IPC mode
    # only, and with the legacy mode (the ROCr default) hipIpcGetMemHandle returns "invalid argument" (documented
    # for this image, which exports the variable itself).  setdefault: an outer setting wins.
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    return env


def spawn_ranks(n: int, argv: Sequence[str], *, build: bool = True, poll: float = 0.05,
                env: Optional[Dict[str, str]] = None) -> int:
    """Run ``argv`` as ``n`` ranks; returns 0 when every rank returned 0, else the first failing rank's code
    (128 + signal for a rank killed by a signal)."""
    if n < 1:
        raise ValueError('spawn_ranks: n must be >= 1')
    if build:
        from np_modeling_amd import _C
        _C.build_if_missing()                     # make only; nothing here touches a device
    workdir = tempfile.mkdtemp(prefix='npm_launch_')
    rendezvous = os.path.join(workdir, 'rccl_id')
    token = new_launch_token()                    # one per launch: a file of any other launch is not this one's
    procs: List[subprocess.Popen] = []
    code = 0

    def on_signal(signum, frame):
        raise _Terminated(signum)

    handled = _HANDLED
    _resolve_prctl()
    previous = {}
    try:
        for sig in handled:
            previous[sig] = signal.signal(sig, on_signal)
    except ValueError:                              # not the main thread: the caller owns signal handling
        previous = {}
    try:
        for rank in range(n):
            # a handled signal that arrives while Popen runs would raise out of it and leave a child that is in no
            # list: hold it until the child is recorded (the child lifts the inherited block, _child_setup)
            blocked = signal.pthread_sigmask(signal.SIG_BLOCK, handled) if previous else None
            try:
                procs.append(subprocess.Popen(list(argv), env=rank_environment(rank, n, rendezvous, env, token),
                                              stdout=None if rank == 0 else sys.stderr, preexec_fn=_child_setup))
            finally:
                if blocked is not None:
                    signal.pthread_sigmask(signal.SIG_SETMASK, blocked)
        pending = set(range(n))
        while pending and code == 0:
            for rank in sorted(pending):
                rc = procs[rank].poll()
                if rc is None:
                    continue
                pending.discard(rank)
                if rc != 0:
                    code = rc if rc > 0 else 128 - rc
                    print(f'np_modeling_amd.launch: rank {rank} exited with code {rc}; stopping the other ranks',
                          file=sys.stderr, flush=True)
                    break
            if pending and code == 0:
                time.sleep(poll)
    except KeyboardInterrupt:
        code = 130
    except _Terminated as stop:
        code = 128 + stop.signum                    # 143 for SIGTERM, like a shell reports it
    finally:
        for sig in previous:                        # a second signal during the clean-up must not abandon it
            signal.signal(sig, signal.SIG_IGN)
        for p in procs:                             # exactly the processes started here, by handle
            if p.poll() is None:
                p.send_signal(signal.SIGTERM)
        deadline = time.monotonic() + 10.0
        for p in procs:
            try:
                p.wait(timeout=max(0.0, deadline - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        shutil.rmtree(workdir, ignore_errors=True)
        for sig, handler in previous.items():
            signal.signal(sig, handler)
    return code


def main(args: Optional[Sequence[str]] = None) -> int:
    import argparse
    ap = argparse.ArgumentParser(prog='python -m np_modeling_amd.launch',
                                 description='run a script as one process per GPU of this node')
    ap.add_argument('--gpus', type=int, required=True)
    ap.add_argument('script')
    ap.add_argument('script_args', nargs=argparse.REMAINDER)
    ns = ap.parse_args(args)
    return spawn_ranks(ns.gpus, [sys.executable, ns.script] + list(ns.script_args))


if __name__ == '__main__':
    sys.exit(main())
