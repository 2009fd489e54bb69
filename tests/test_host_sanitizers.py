"""CPU: the host halves of the two libraries -- the runtime / caching pool of libnpm_hip.so (csrc/npm_runtime.hip) and the
whole exchange shim (csrc/npm_comm.cpp) -- built with g++ -fsanitize=address,undefined against host-memory stand-ins for HIP
and RCCL (tests/hostmock/) and driven through their C ABI by tests/hostmock/sanitize_main.cpp: argument and state errors,
size classes and recycling, the out-of-memory retry, copies, events, four host threads on the pool, the exchange's event
bookkeeping (cap on pending spans, recycling), init / destroy pairs.  AddressSanitizer's leak check and the stand-ins'
live-object counters see whatever is not given back.  (GPU AddressSanitizer is not available on this pool; device code is
covered by the parity tests.)"""

import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which('g++') is None, reason='needs g++')
def test_host_code_under_address_and_ub_sanitizers(tmp_path):
    mock = os.path.join(ROOT, 'tests', 'hostmock')
    csrc = os.path.join(ROOT, 'np_modeling_amd', 'csrc')
    exe = str(tmp_path / 'sanitize_main')
    cmd = ['g++', '-std=c++17', '-g', '-O1', '-fno-omit-frame-pointer', '-fsanitize=address,undefined', '-fno-sanitize-recover=all',
           '-Wall', '-Wextra', '-Wno-unused-parameter', '-pthread',
           f'-I{mock}', f'-I{os.path.join(ROOT, "include")}', f'-I{csrc}',
           '-x', 'c++', os.path.join(csrc, 'npm_runtime.hip'), os.path.join(csrc, 'npm_comm.cpp'),
           os.path.join(mock, 'mock.cpp'), os.path.join(mock, 'sanitize_main.cpp'), '-ldl', '-o', exe]
    build = subprocess.run(cmd, capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-4000:]
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=1:abort_on_error=1', UBSAN_OPTIONS='print_stacktrace=1')
    run = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=300)
    assert run.returncode == 0 and 'host sanitizers: ok' in run.stdout, (run.stdout[-2000:], run.stderr[-6000:])
