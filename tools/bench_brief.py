#!/usr/bin/env python3
"""stdin: bench.py's JSON line -> one short line (ms per step, GEMM / attention / LayerNorm figures)."""
import json
import sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
o, h = r.get('other_mfma_kernels', {}), r.get('hbm_kernels', {})
print(round(d['value'], 1), 'samples/s', round(d['ms_per_step'], 3), 'ms; gemm', round(r['frac'], 4),
      {k: round(v['avg_ms'], 3) for k, v in r['by_layout'].items()},
      {k: (round(v['avg_ms'], 3), round(v['frac'], 3)) for k, v in o.items()},
      {k: (round(v['avg_ms'], 4), round(v.get('frac_of_8TBps', 0), 3)) for k, v in h.items()})
