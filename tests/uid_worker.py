"""Worker for test_unique_id_exchange_under_launcher: runs under ``python -m torch.distributed.run`` with
two ranks and checks that rank 0's communicator id reaches the other rank through the launcher's store
(np_modeling_amd/parallel.py:_exchange_unique_id).  No GPU, no RCCL: the id is a fixed byte pattern."""

import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from np_modeling_amd import parallel  # noqa: E402

PATTERN = bytes((37 * i + 11) % 256 for i in range(128))      # includes NUL bytes, like a real ncclUniqueId


def main():
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    assert os.environ.get('TORCHELASTIC_USE_AGENT_STORE', '').lower() == 'true'
    made = []

    def fake_id():
        made.append(rank)
        return PATTERN

    parallel.RcclCommunicator.new_unique_id = staticmethod(fake_id)
    uid = parallel._exchange_unique_id(rank, world)
    assert uid == PATTERN, (rank, len(uid))
    assert made == ([0] if rank == 0 else []), made       # only rank 0 creates the id
    print(f'rank {rank}/{world}: id ok', flush=True)


if __name__ == '__main__':
    main()
