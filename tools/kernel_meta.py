#!/usr/bin/env python3
"""Register / spill / LDS metadata of every gfx950 kernel in a built object (or in libnpm_hip.so's objects).

    python tools/kernel_meta.py [np_modeling_amd/lib/npm_attn.o ...] [--filter mha_] [--disasm KERNEL_SUBSTRING]

Reads the code object's AMDGPU metadata note (llvm-readelf --notes) after unbundling the device image with
llvm-objdump --offloading.  tests/test_abi_exports.py uses :func:`kernel_metadata` to assert that no attention kernel
spills vector registers.  --disasm prints an opcode histogram of one kernel (instruction diet of the MFMA loops)."""

import argparse
import collections
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = '/opt/rocm/lib/llvm/bin'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIELDS = ('.vgpr_count', '.agpr_count', '.sgpr_count', '.vgpr_spill_count', '.sgpr_spill_count',
          '.private_segment_fixed_size', '.group_segment_fixed_size')


def device_image(obj, workdir):
    """Path of the gfx950 code object embedded in the host object `obj` (extracted into workdir)."""
    local = os.path.join(workdir, os.path.basename(obj))
    shutil.copy(obj, local)
    subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '--offloading', local], check=True, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL, cwd=workdir)
    images = glob.glob(local + '.*gfx950*')
    if not images:
        raise RuntimeError(f'no gfx950 image in {obj}')
    return images[0]


def demangle(names):
    tool = shutil.which('c++filt') or os.path.join(LLVM, 'llvm-cxxfilt')
    try:
        out = subprocess.run([tool] + list(names), capture_output=True, text=True)
    except OSError:
        return list(names)
    return out.stdout.split('\n')[:len(names)] if out.returncode == 0 else list(names)


def kernel_metadata(obj):
    """{demangled kernel name: {field: int}} for every kernel of a built host object."""
    with tempfile.TemporaryDirectory() as tmp:
        try:
            image = device_image(obj, tmp)
        except RuntimeError:
            return {}                     # an object without device code (npm_runtime.o)
        notes = subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '--notes', image], check=True,
                               capture_output=True, text=True).stdout
    if 'amdhsa.kernels' not in notes:
        return {}
    import yaml
    doc = yaml.safe_load(notes[notes.index('---'):notes.rindex('...')])
    kernels = {k['.name']: k for k in doc['amdhsa.kernels']}
    names = list(kernels)
    return {d: {f: int(kernels[n].get(f, 0)) for f in FIELDS} for n, d in zip(names, demangle(names))}


def disassemble(obj, pattern):
    with tempfile.TemporaryDirectory() as tmp:
        text = subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '-d', device_image(obj, tmp)], check=True,
                              capture_output=True, text=True).stdout
    blocks = re.split(r'\n(?=[0-9a-f]{16} <)', text)
    return [b for b in blocks if pattern in b.split('\n', 1)[0]]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('objects', nargs='*')
    ap.add_argument('--filter', default='')
    ap.add_argument('--disasm', default='', help='print the opcode histogram of kernels whose name contains this')
    ap.add_argument('--dump', default='', help='with --disasm: write the disassembly of the matching kernels here')
    args = ap.parse_args()
    objects = args.objects or sorted(glob.glob(os.path.join(ROOT, 'np_modeling_amd', 'lib', '*.o')))
    for obj in objects:
        if args.disasm:
            for block in disassemble(obj, args.disasm):
                head, body = block.split('\n', 1)
                ops = collections.Counter(line.split()[0] for line in body.splitlines() if line.startswith('\t'))
                total = sum(ops.values())
                print(head.strip(), f'-- {total} instructions')
                print('   ', ', '.join(f'{n} {op}' for op, n in ops.most_common(28)))
                if args.dump:
                    with open(args.dump, 'a') as f:
                        f.write(block + '\n')
            continue
        for name, meta in kernel_metadata(obj).items():
            if args.filter in name:
                print(f"{os.path.basename(obj)}: {name[:110]:<110s} vgpr {meta.get('.vgpr_count', 0):3d} agpr {meta.get('.agpr_count', 0):3d} "
                      f"sgpr {meta.get('.sgpr_count', 0):3d} spill v{meta.get('.vgpr_spill_count', 0)} s{meta.get('.sgpr_spill_count', 0)} "
                      f"scratch {meta.get('.private_segment_fixed_size', 0)} lds {meta.get('.group_segment_fixed_size', 0)}")


if __name__ == '__main__':
    sys.exit(main())
