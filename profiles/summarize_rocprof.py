#!/usr/bin/env python3
"""Summarise a ``rocprofv3 --kernel-trace --output-format csv`` trace of bench.py.

    python profiles/summarize_rocprof.py <kernel_trace.csv> --steps K --warmup W > profiles/<name>.md

bench.py runs W warm-up steps and K timed steps that launch the same kernels; per kernel name
the launches are split evenly over the W+K steps, so the timed region is the last K/(W+K) of
each kernel's launches in time order.  Reports both the whole run and the timed region."""

import argparse
import collections
import csv


def short(name):
    return name.replace('(anonymous namespace)::', '').replace('void ', '')


def table(groups, title):
    total = sum(sum(d) for d in groups.values())
    lines = [f'### {title}', '', '| kernel | calls | total ms | avg ms | min ms | max ms | % |', '|---|---|---|---|---|---|---|']
    for name, durs in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
        lines.append(f'| `{short(name)[:110]}` | {len(durs)} | {sum(durs) / 1e6:.3f} | {sum(durs) / len(durs) / 1e6:.4f} | '
                     f'{min(durs) / 1e6:.4f} | {max(durs) / 1e6:.4f} | {100 * sum(durs) / total:.2f} |')
    lines.append(f'| **all kernels** | {sum(len(d) for d in groups.values())} | {total / 1e6:.3f} | | | | 100 |')
    return '\n'.join(lines)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('trace')
    ap.add_argument('--steps', type=int, required=True)
    ap.add_argument('--warmup', type=int, required=True)
    args = ap.parse_args()
    rows = list(csv.DictReader(open(args.trace)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    by_name = collections.OrderedDict()
    for r in rows:
        by_name.setdefault(r['Kernel_Name'], []).append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    timed = {}
    for name, durs in by_name.items():
        per_step, rem = divmod(len(durs), args.steps + args.warmup)
        if per_step == 0 or rem:
            continue                      # not a per-step kernel (copies, one-off launches)
        timed[name] = durs[-per_step * args.steps:]
    print(f'# rocprofv3 kernel trace: {args.trace}\n')
    print(f'bench.py --steps {args.steps} --warmup {args.warmup}; durations are End-Start of each dispatch.\n')
    print(table(timed, f'timed region (last {args.steps} of {args.steps + args.warmup} steps)'))
    print()
    print(table(by_name, 'whole process'))


if __name__ == '__main__':
    main()
