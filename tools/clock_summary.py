#!/usr/bin/env python3
"""Summarise a tools/clock_sampler.sh log: the card under load (most samples above 800 W), its clock and power while loaded."""
import re
import statistics
import sys

rows = [line.split('|')[1:] for line in open(sys.argv[1]) if '|' in line]
n = max(len(r) for r in rows)
rows = [r for r in rows if len(r) == n]
watts = lambda cell: int(re.findall(r'(\d+) W', cell)[0])
mhz = lambda cell: int(re.findall(r'(\d+) MHz', cell)[0])
# the card under THIS load: loaded for the longest part of the log (another tenant's card of the host may peak higher for a moment)
card = max(range(n), key=lambda i: sum(watts(r[i]) > 800 for r in rows))
pw = [watts(r[card]) for r in rows]
ck = [mhz(r[card]) for r in rows]
hot = [(c, p) for c, p in zip(ck, pw) if p > 0.8 * max(pw)]
print(f'card {card}: {len(hot)} samples under load; clock MHz median {statistics.median(c for c, _ in hot):.0f} '
      f'(min {min(c for c, _ in hot)}, max {max(c for c, _ in hot)}); power W median {statistics.median(p for _, p in hot):.0f} (max {max(pw)})')
