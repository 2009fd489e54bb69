#!/usr/bin/env python3
"""HBM-side traffic of the GEMM family from two rocprofv3 PMC passes of bench.py.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d A -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d B -- python3 bench.py ... (same)
    python profiles/summarize_pmc.py A/*/*_counter_collection.csv B/*/*_counter_collection.csv [--session NAME] > profiles/pmc_traffic.json

The result names the session it comes from (``session``) and the sources the measured library was built from
(``source_id`` = np_modeling_amd._C.source_id()); bench.py quotes the figure only for that build.

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md section HBM: both counters are in KiB; on gfx950
FETCH_SIZE reports exactly half of the bytes of wide (16 B/lane) coalesced streaming reads, so it is doubled;
WRITE_SIZE is exact.  FETCH_SIZE counts the L2's fabric-side requests: Infinity-Cache hits are included, so this is
an upper bound of true HBM reads.  The K-major DMA loads of this kernel are 64-byte segments per row, for which the
doubling may over-count (calibration: softmax/LayerNorm rows in the same run read exactly their algorithmic bytes
after doubling)."""

import collections
import csv
import json
import sys


def per_kernel(path):
    out = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        out.setdefault(r['Kernel_Name'], []).append(float(r['Counter_Value']))
    return out


def main():
    import datetime
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from np_modeling_amd import _C
    session = 'unnamed'
    if '--session' in sys.argv:
        at = sys.argv.index('--session')
        session = sys.argv[at + 1]
        del sys.argv[at:at + 2]
    fetch, write = per_kernel(sys.argv[1]), per_kernel(sys.argv[2])
    kernels = {}
    for name in fetch:
        short = name.replace('(anonymous namespace)::', '').replace('void ', '')
        f, w = fetch[name], write.get(name, [0.0])
        kernels[short] = dict(launches=len(f), fetch_kib_raw_per_launch=sum(f) / len(f), write_kib_per_launch=sum(w) / len(w),
                              bytes_per_launch=(2 * sum(f) / len(f) + sum(w) / len(w)) * 1024)
    gemm = {k: v for k, v in kernels.items() if k.startswith('sgemm_')}      # bench.py's roofline kernel
    n = sum(v['launches'] for v in gemm.values())
    family = sum(v['bytes_per_launch'] * v['launches'] for v in gemm.values()) / max(n, 1)
    print(json.dumps(dict(gemm_family_bytes_per_launch=family, gemm_family_launches=n,
                          session=session, collected=datetime.datetime.now().isoformat(timespec='seconds'),
                          source_id=_C.source_id(), library_current=_C.library_is_current(),
                          command='rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --kernel-trace -- python3 bench.py --steps 2 --warmup 1 '
                                  '--no-cpu-baseline --no-kernel-timer --no-alt-math --no-configs', kernels=kernels,
                          note='FETCH_SIZE doubled (gfx950 correction), WRITE_SIZE exact, KiB units; includes Infinity-Cache hits'),
                     indent=1))


if __name__ == '__main__':
    main()
