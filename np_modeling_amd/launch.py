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


def _die_with_parent() -> None:
    """Child side, between fork and exec: PR_SET_PDEATHSIG(SIGTERM) -- the kernel signals the rank when the launcher
    thread that started it exits, however it exits (the clean-up of a SIGKILLed launcher)."""
    import ctypes
    try:
        ctypes.CDLL(None, use_errno=True).prctl(1, int(signal.SIGTERM), 0, 0, 0)      # PR_SET_PDEATHSIG = 1
    except Exception:
        pass


def rank_environment(rank: int, world: int, rendezvous_file: str, base: Optional[Dict[str, str]] = None) -> Dict[str, str]:
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               NPM_RENDEZVOUS_FILE=rendezvous_file)
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
    procs: List[subprocess.Popen] = []
    code = 0

    def on_signal(signum, frame):
        raise _Terminated(signum)

    handled = (signal.SIGTERM, signal.SIGHUP)
    previous = {}
    try:
        for sig in handled:
            previous[sig] = signal.signal(sig, on_signal)
    except ValueError:                              # not the main thread: the caller owns signal handling
        previous = {}
    try:
        for rank in range(n):
            procs.append(subprocess.Popen(list(argv), env=rank_environment(rank, n, rendezvous, env),
                                          stdout=None if rank == 0 else sys.stderr, preexec_fn=_die_with_parent))
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
