#!/usr/bin/env python3
"""Guards the HAND-COUNTED `s_waitcnt vmcnt(N)` of the LDS-DMA pipelines against the toolchain.

The attention backward kernels, the fused Conv2D filter gradient and the three-stage GEMM issue their LDS-DMA pieces
(`buffer_load_dwordx4 ... lds`) where the compiler's wait-count pass does not see them (inline assembly; a branch between
issue and read) and order them with waits written by hand: "everything but the N youngest vector-memory instructions of
this wave has completed".  N is the number of vector-memory instructions the wave issues AFTER its last piece -- a count
of COMPILER-EMITTED loads and stores.  A toolchain that splits, merges, adds or reorders one of them makes the wait too
lax, and a kernel that reads a tile before it has landed can still pass the parity tests.

This tool compiles the sources to gfx950 assembly (hipcc -S, device only) and reduces every kernel that contains a tagged
wait (`; npm:wait` behind the instruction) to a SIGNATURE: the order, in the text of the kernel, of
    D<n>  n consecutive LDS-DMA pieces          L<n>  n other vector-memory loads      S<n>  n vector-memory stores
    W<N>  a tagged hand-written wait            B     s_barrier
The signatures of the build the counts were validated on are pinned in tools/waitcnt_pins.json together with the compiler's
version; tests/test_abi_exports.py::test_hand_counted_waits_match_the_pinned_disassembly compares.  After a toolchain
change: re-derive every N from the new signature (the comments at the waits say what is counted), re-run the GPU suite,
then `python tools/waitcnt_check.py --pin`.

    python tools/waitcnt_check.py            # compare with the pins, exit 1 on a difference
    python tools/waitcnt_check.py --show     # print the current signatures
"""

import argparse
import functools
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'np_modeling_amd', 'csrc')
PINS = os.path.join(ROOT, 'tools', 'waitcnt_pins.json')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
SOURCES = ('npm_attn.hip', 'npm_conv.hip', 'npm_gemm.hip')
# The three-stage GEMM's counted wait looks at DMA pieces only (`D8 W4 B D4`: the K loop); everything behind the loop is the
# epilogue, whose store sequences change with every epilogue specialisation and are not what the wait counts: its signature
# ends with the loop.  The attention and convolution kernels keep theirs whole (their waits DO count trailing stores).
LOOP_ONLY = ('npm_gemm.hip',)
VMEM = re.compile(r'^\s*(buffer_(load|store)|global_(load|store)|scratch_(load|store)|flat_(load|store))')


def compiler_version() -> str:
    out = subprocess.run([HIPCC, '--version'], capture_output=True, text=True, check=True).stdout
    hip = re.search(r'HIP version: (\S+)', out)
    clang = re.search(r'clang version (\S+)', out)
    return f'hip {hip.group(1) if hip else "?"} clang {clang.group(1) if clang else "?"}'


@functools.lru_cache(maxsize=None)
def assembly(source: str) -> str:
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, 'a.s')
        subprocess.run([HIPCC, '-S', '--offload-arch=gfx950', '--cuda-device-only', '-O3', '-std=c++17',
                        f'-I{os.path.join(ROOT, "include")}', f'-I{CSRC}', os.path.join(CSRC, source), '-o', out],
                       check=True, capture_output=True)
        with open(out) as f:
            return f.read()


def demangle(names):
    import shutil
    tool = shutil.which('c++filt') or '/opt/rocm/lib/llvm/bin/llvm-cxxfilt'
    if not names:
        return []
    try:
        out = subprocess.run([tool] + list(names), capture_output=True, text=True)
    except OSError:
        return list(names)
    return out.stdout.split('\n')[:len(names)] if out.returncode == 0 else list(names)


def signatures(text: str, loop_only: bool = False) -> dict:
    """{demangled kernel name: signature} for the kernels of one assembly file that contain a tagged wait."""
    found = {}
    blocks = re.split(r'\n(?=_Z\w+:\s)', text)
    for block in blocks:
        head = re.match(r'(_Z\w+):', block)
        if not head or 'npm:wait' not in block:
            continue
        body = block.split('.Lfunc_end', 1)[0]
        tokens = []

        def push(kind):
            if tokens and tokens[-1][0] == kind:
                tokens[-1][1] += 1
            else:
                tokens.append([kind, 1])

        for line in body.splitlines():
            code = line.split(';', 1)[0]
            if 'npm:wait' in line:
                n = re.search(r'vmcnt\((\d+)\)', line)
                tokens.append([f'W{n.group(1)}', 0])
            elif 's_barrier' in code:
                tokens.append(['B', 0])
            elif VMEM.match(code):
                if re.search(r'\blds\b', code):
                    push('D')
                elif 'store' in code.split()[0]:
                    push('S')
                else:
                    push('L')
        if loop_only:                    # up to the last tagged wait, then the barriers / DMA groups that follow it directly
            last = max(i for i, (k, _) in enumerate(tokens) if k.startswith('W'))
            end = last + 1
            while end < len(tokens) and tokens[end][0] in ('B', 'D'):
                end += 1
            tokens = tokens[:end]
        found[head.group(1)] = ' '.join(k if n == 0 else f'{k}{n}' for k, n in tokens)
    names = list(found)
    return {d.replace('(anonymous namespace)::', ''): found[n] for n, d in zip(names, demangle(names))}


def scratch_depth(text: str) -> dict:
    """{demangled kernel name: deepest loop nesting at which a scratch (spill) instruction sits} for the kernels of one
    assembly file that touch scratch at all; 0 = straight-line code.  The compiler annotates every block of a loop with
    `in Loop: Header=... Depth=N` (`Inner Loop Header: Depth=N` / `Loop Header: Depth=N` on the header itself)."""
    found = {}
    for block in re.split(r'\n(?=_Z\w+:\s)', text):
        head = re.match(r'(_Z\w+):', block)
        if not head:
            continue
        depth, worst = 0, None
        for line in block.split('.Lfunc_end', 1)[0].splitlines():
            if re.match(r'\.LBB\w+:', line):
                d = re.search(r'Depth=(\d+)', line)
                depth = int(d.group(1)) if d else 0
            elif re.match(r'\s*scratch_(load|store)', line):
                worst = depth if worst is None else max(worst, depth)
        if worst is not None:
            found[head.group(1)] = worst
    names = list(found)
    return {d.replace('(anonymous namespace)::', ''): found[n] for n, d in zip(names, demangle(names))}


def current() -> dict:
    from concurrent.futures import ThreadPoolExecutor
    out = {'compiler': compiler_version(), 'kernels': {}}
    with ThreadPoolExecutor(max_workers=len(SOURCES)) as pool:          # three hipcc processes side by side
        texts = list(pool.map(assembly, SOURCES))
    for src, text in zip(SOURCES, texts):
        for name, sig in signatures(text, src in LOOP_ONLY).items():
            out['kernels'][f'{src}: {name}'] = sig
    return out


def compare(now: dict, pinned: dict) -> list:
    problems = []
    if now['compiler'] != pinned['compiler']:
        problems.append(f'compiler is {now["compiler"]!r}, the counts were validated on {pinned["compiler"]!r}')
    for name in sorted(set(now['kernels']) | set(pinned['kernels'])):
        a, b = now['kernels'].get(name), pinned['kernels'].get(name)
        if a != b:
            problems.append(f'{name}\n    now   : {a}\n    pinned: {b}')
    return problems


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument('--pin', action='store_true', help='write the current signatures as the pins')
    ap.add_argument('--show', action='store_true')
    args = ap.parse_args()
    now = current()
    if args.show or args.pin:
        print(json.dumps(now, indent=1))
    if args.pin:
        with open(PINS, 'w') as f:
            json.dump(now, f, indent=1)
            f.write('\n')
        return 0
    with open(PINS) as f:
        problems = compare(now, json.load(f))
    for line in problems:
        print(line)
    return 1 if problems else 0


if __name__ == '__main__':
    sys.exit(main())
