"""Worker of the rendezvous tests in tests/test_host_logic.py: started as one of several ranks (by the product's
self-launcher np_modeling_amd/launch.py, or by ``python -m torch.distributed.run``) it checks that rank 0's
communicator id reaches every other rank through np_modeling_amd/parallel.py:_exchange_unique_id, and that the
product imported no torch on the way.  No GPU, no RCCL: the id is a fixed byte pattern.

``UID_WORKER_FAIL_RANK=r`` makes rank r exit with code 7 after the exchange (the launcher must stop the others);
``UID_WORKER_HANG=1`` makes the other ranks wait (they must be terminated, not waited for) after appending their
pid to ``UID_WORKER_PIDFILE``."""

import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from np_modeling_amd import parallel  # noqa: E402

PATTERN = bytes((37 * i + 11) % 256 for i in range(128))      # includes NUL bytes, like a real ncclUniqueId


def main():
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    made = []

    def fake_id():
        made.append(rank)
        return PATTERN

    parallel.RcclCommunicator.new_unique_id = staticmethod(fake_id)
    uid = parallel._exchange_unique_id(rank, world, timeout=60.0)
    assert uid == PATTERN, (rank, len(uid))
    assert made == ([0] if rank == 0 else []), made       # only rank 0 creates the id
    assert 'torch' not in sys.modules, 'the product path imported torch'
    print(f'rank {rank}/{world}: id ok', flush=True)
    if not os.environ.get('NPM_RENDEZVOUS_FILE'):          # derived path (external launcher): tidy up after the last reader
        path = parallel.rendezvous_path()
        if rank:
            open(f'{path}.ack{rank}', 'w').close()
        else:
            deadline = time.monotonic() + 60
            while not all(os.path.exists(f'{path}.ack{r}') for r in range(1, world)) and time.monotonic() < deadline:
                time.sleep(0.01)
            for name in [path] + [f'{path}.ack{r}' for r in range(1, world)]:
                if os.path.exists(name):
                    os.unlink(name)
    fail_rank = os.environ.get('UID_WORKER_FAIL_RANK')
    if fail_rank is not None:
        if rank == int(fail_rank):
            sys.exit(7)
        if os.environ.get('UID_WORKER_HANG') == '1':
            if os.environ.get('UID_WORKER_PIDFILE'):        # the terminated-launcher test watches these processes
                with open(os.environ['UID_WORKER_PIDFILE'], 'a') as f:
                    f.write(f'{os.getpid()}\n')
            time.sleep(120)


if __name__ == '__main__':
    main()
