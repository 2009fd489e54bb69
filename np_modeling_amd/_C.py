"""ctypes binding of ``libnpm_hip.so`` (C ABI in ``include/npm_hip.h``).

There is no CPU path: if the shared library is missing, or no MI355X is visible, the
first use raises.  (``import np_modeling_amd`` itself stays importable so that the
build check can import the package on a machine without a GPU.)
"""

from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, 'lib')
LIB_PATH = os.path.join(LIB_DIR, 'libnpm_hip.so')
RCCL_LIB_PATH = os.path.join(LIB_DIR, 'libnpm_rccl.so')


class NpmError(RuntimeError):
    """A C-ABI call returned non-zero (``code``: the NPM_E_* / HIP / RCCL value, None for host-side errors)."""

    def __init__(self, message: str, code=None):
        super().__init__(message)
        self.code = code


class npm_gemm(C.Structure):
    _fields_ = [
        ('trans_a', C.c_int32), ('trans_b', C.c_int32),
        ('m', C.c_int32), ('n', C.c_int32), ('k', C.c_int32),
        ('batch0', C.c_int32), ('batch1', C.c_int32),
        ('a', C.c_void_p), ('lda', C.c_int64), ('stride_a0', C.c_int64), ('stride_a1', C.c_int64),
        ('b', C.c_void_p), ('ldb', C.c_int64), ('stride_b0', C.c_int64), ('stride_b1', C.c_int64),
        ('c', C.c_void_p), ('ldc', C.c_int64), ('stride_c0', C.c_int64), ('stride_c1', C.c_int64),
        ('alpha', C.c_float),
        ('epilogue', C.c_int32),
        ('bias', C.c_void_p),
        ('residual', C.c_void_p), ('ldr', C.c_int64),
        ('aux', C.c_void_p), ('ldaux', C.c_int64),
        ('split_k', C.c_int32),
        ('rowvec', C.c_void_p),
        ('colsum', C.c_void_p),
        ('bsum', C.c_void_p),
        ('asum', C.c_void_p),
        ('rowdot', C.c_void_p), ('rowdot_scale', C.c_float),
    ]


class npm_conv2d(C.Structure):
    _fields_ = [
        ('n', C.c_int32), ('h', C.c_int32), ('w', C.c_int32),
        ('c_in', C.c_int32), ('c_out', C.c_int32), ('ksize', C.c_int32),
        ('x', C.c_void_p), ('filt', C.c_void_p), ('bias', C.c_void_p),
        ('y', C.c_void_p), ('pre', C.c_void_p),
        ('relu', C.c_int32),
    ]


class npm_mha_core(C.Structure):
    _fields_ = [
        ('batch', C.c_int32), ('heads', C.c_int32), ('seq_q', C.c_int32), ('seq_kv', C.c_int32), ('head_dim', C.c_int32),
        ('scale', C.c_float),
        ('q', C.c_void_p), ('q_pitch', C.c_int64),
        ('k', C.c_void_p), ('k_pitch', C.c_int64),
        ('v', C.c_void_p), ('v_pitch', C.c_int64),
        ('mask', C.c_void_p), ('mask_stride_b', C.c_int64), ('mask_stride_h', C.c_int64), ('mask_stride_q', C.c_int64),
        ('ctx', C.c_void_p), ('ctx_pitch', C.c_int64),
        ('lse', C.c_void_p),
        ('scores', C.c_void_p),
        ('dctx', C.c_void_p), ('dctx_pitch', C.c_int64),
        ('dq', C.c_void_p), ('dq_pitch', C.c_int64),
        ('dk', C.c_void_p), ('dk_pitch', C.c_int64),
        ('dv', C.c_void_p), ('dv_pitch', C.c_int64),
        ('tile_summary', C.c_void_p), ('summary_stride_b', C.c_int64), ('summary_stride_h', C.c_int64),
        ('summary_all_offset', C.c_int64),
        ('neg_delta', C.c_void_p), ('neg_delta_stride_b', C.c_int64), ('neg_delta_stride_h', C.c_int64),
    ]


class npm_comm_exchange_stats(C.Structure):
    _fields_ = [('bytes', C.c_ulonglong), ('allreduce_calls', C.c_int), ('waits', C.c_int),
                ('allreduce_ms', C.c_double), ('exposed_ms', C.c_double), ('last_allreduce_ms', C.c_double),
                ('dropped', C.c_int)]


EPI_BIAS, EPI_RESIDUAL, EPI_RELU_SAVE, EPI_RELU_MASK, EPI_RELU, EPI_SOFTMAX_BWD, EPI_ROWDOT = 1, 2, 4, 8, 16, 32, 128

_P, _SZ, _I64, _I32, _F = C.c_void_p, C.c_size_t, C.c_int64, C.c_int32, C.c_float

# name -> argtypes; every function returns int (0 = ok) unless listed in _SPECIAL.
SIGNATURES = {
    'npm_device_count': [C.POINTER(C.c_int)],
    'npm_init': [C.c_int],
    'npm_shutdown': [],
    'npm_device_name': [C.c_char_p, C.c_int],
    'npm_sync': [],
    'npm_malloc': [C.POINTER(_P), _SZ],
    'npm_free': [_P],
    'npm_pool_stats': [C.POINTER(_SZ), C.POINTER(_SZ)],
    'npm_pool_trim': [],
    'npm_h2d': [_P, _P, _SZ],
    'npm_d2h': [_P, _P, _SZ],
    'npm_d2d': [_P, _P, _SZ],
    'npm_fill_f32': [_P, _F, _SZ],
    'npm_event_create': [C.POINTER(_P)],
    'npm_event_destroy': [_P],
    'npm_event_record': [_P],
    'npm_event_sync': [_P],
    'npm_event_elapsed_ms': [_P, _P, C.POINTER(_F)],
    'npm_sgemm': [C.POINTER(npm_gemm)],
    'npm_set_tuning': [C.c_int, C.c_int],
    'npm_set_math': [C.c_int],
    'npm_get_math': [],
    'npm_last_math': [],
    'npm_debug_gemm_trace': [_P],
    'npm_debug_attn_trace': [_P],
    'npm_relu_fwd': [_P, _P, _SZ],
    'npm_relu_bwd': [_P, _P, _P, _SZ],
    'npm_add': [_P, _P, _P, _SZ],
    'npm_add3': [_P, _P, _P, _P, _SZ],
    'npm_axpy': [_P, _P, _F, _SZ],
    'npm_scale': [_P, _P, _F, _SZ],
    'npm_colsum': [_P, _P, _I64, _I64, _I64],
    'npm_relu_bwd_colsum': [_P, _P, _P, _P, _I64, _I64],
    'npm_attn_rowdot': [_P, _P, _P, _I64, _I64, _I64, _I64],
    'npm_softmax_fwd': [_P, _P, _I64, _I64, _F],
    'npm_softmax_bwd': [_P, _P, _P, _I64, _I64, _F],
    'npm_layernorm_fwd': [_P, _P, _P, _F, _I64, _I64, _P, _P, _P],
    'npm_layernorm_bwd': [_P, _P, _P, _P, _P, _P, _I64, _I64, _P, _P, _P],
    'npm_layernorm_dropout_fwd': [_P, _P, _F, _P, _P, _F, _I64, _I64, _P, _P, _P],
    'npm_layernorm_dropout_bwd': [_P, _P, _P, _F, _P, _P, _P, _P, _I64, _I64, _P, _P, _P],
    'npm_conv2d_fwd': [C.POINTER(npm_conv2d)],
    'npm_conv2d_bwd_x': [_P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32],
    'npm_conv2d_bwd_w': [_P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32],
    'npm_conv2d_bwd_w_relu': [_P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32],
    'npm_mha_core_supported': [C.c_int],
    'npm_mha_core_fwd': [C.POINTER(npm_mha_core)],
    'npm_mha_core_bwd': [C.POINTER(npm_mha_core)],
    'npm_mha_mask_summary': [_P, _I64, _I64, _I64, _I32, _I32, _I32, _I32, _P],
    'npm_adam_step': [_P, _P, _P, _P, _SZ, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int],
    'npm_fill_f64': [_P, C.c_double, _SZ],
    'npm_mse_fwd': [_P, _P, _SZ, C.POINTER(C.c_double)],
    'npm_mse_bwd': [_P, _P, _P, _SZ],
    'npm_xent_fwd': [_P, _P, _SZ, C.POINTER(C.c_double)],
    'npm_xent_bwd': [_P, _P, _P, _SZ],
    'npm_mask_scale': [_P, _P, _P, _SZ, _F],
    'npm_dropout_philox': [_P, _P, _P, _SZ, _F, C.c_uint64, C.c_uint64],
}
_SPECIAL = {
    'npm_abi_version': (C.c_int, []),
    'npm_last_error': (C.c_char_p, []),
    'npm_stream': (C.c_void_p, []),
    'npm_last_attn_kernel': (C.c_char_p, []),
}

COMM_SIGNATURES = {
    'npm_comm_library_path': [C.c_char_p, C.c_int],
    'npm_comm_unique_id': [C.c_char_p],
    'npm_comm_init': [C.c_char_p, C.c_int, C.c_int, _P],
    'npm_comm_rank': [C.POINTER(C.c_int), C.POINTER(C.c_int)],
    'npm_comm_allreduce_f32': [_P, _SZ, C.c_int],
    'npm_comm_broadcast_f32': [_P, _SZ, C.c_int],
    'npm_comm_wait': [],
    'npm_comm_stats_enable': [C.c_int],
    'npm_comm_stats': [C.POINTER(npm_comm_exchange_stats)],
    'npm_comm_barrier': [],
    'npm_comm_allreduce_host_f64': [C.POINTER(C.c_double), C.c_int],
    'npm_comm_destroy': [],
}
_COMM_SPECIAL = {'npm_comm_last_error': (C.c_char_p, [])}

_LIB: Optional[object] = None       # the loaded library (tests may install a host simulator here)
_COMM_LIB: Optional[object] = None
_DEVICE: Optional[int] = None
# Set by device.UpdateQueue while it holds queued parameter updates: called by lib() in front of every library call that
# is not itself part of the queue, so a launch issued between two queued updates sees them applied (program order).
_ORDER_HOOK = None


def _bind(cdll, signatures, special):
    for name, argtypes in signatures.items():
        fn = getattr(cdll, name)
        fn.argtypes = argtypes
        fn.restype = C.c_int
    for name, (restype, argtypes) in special.items():
        fn = getattr(cdll, name)
        fn.argtypes = argtypes
        fn.restype = restype
    return cdll


def build_if_missing() -> None:
    """The shared libraries are build artefacts (git-ignored).  If one is absent and the ROCm compiler is present,
    build them in tree -- still no CPU fallback: no compiler, no library.  Serialised across processes by a file
    lock (N ranks importing the package at once must not run ``make`` concurrently); a launcher calls this before
    it starts the ranks (np_modeling_amd/launch.py)."""
    if (os.path.exists(LIB_PATH) and os.path.exists(RCCL_LIB_PATH)) or os.environ.get('NPM_NO_AUTOBUILD') == '1':
        return
    import fcntl
    import shutil
    import subprocess
    hipcc = os.environ.get('HIPCC') or shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        return
    os.makedirs(LIB_DIR, exist_ok=True)
    with open(os.path.join(LIB_DIR, '.build.lock'), 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if os.path.exists(LIB_PATH) and os.path.exists(RCCL_LIB_PATH):      # another process built them meanwhile
            return
        proc = subprocess.run(['make', '-C', os.path.join(_HERE, 'csrc'), '-j4', f'HIPCC={hipcc}'],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if proc.returncode != 0:
            raise NpmError('building the HIP libraries failed:\n' + proc.stdout[-4000:])


def _build_if_missing(path: str) -> None:
    if not os.path.exists(path):
        build_if_missing()


def load_library(path: str = LIB_PATH):
    """dlopen libnpm_hip.so and declare every prototype of include/npm_hip.h."""
    if path == LIB_PATH:
        _build_if_missing(path)
    if not os.path.exists(path):
        raise NpmError(
            f'{path} is missing: build the HIP library first '
            f'(python -c "import __graft_entry__ as g; g.build()" or make -C np_modeling_amd/csrc). '
            'np_modeling_amd has no CPU fallback.')
    return _bind(C.CDLL(path, mode=C.RTLD_GLOBAL), SIGNATURES, _SPECIAL)


def load_comm_library(path: str = RCCL_LIB_PATH):
    if path == RCCL_LIB_PATH:
        _build_if_missing(path)
    if not os.path.exists(path):
        raise NpmError(f'{path} is missing: build it with make -C np_modeling_amd/csrc')
    return _bind(C.CDLL(path, mode=C.RTLD_GLOBAL), COMM_SIGNATURES, _COMM_SPECIAL)


def source_id() -> str:
    """16 hex digits over the kernel sources and headers the libraries are built from (csrc/*, include/*.h).  A profile that is
    not taken inside the measuring process (the PMC traffic figure of bench.py) records it, and whoever quotes the profile
    compares: numbers of another build are not this build's."""
    import glob
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    digest = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(root, 'np_modeling_amd', 'csrc', '*.hip')) + glob.glob(os.path.join(root, 'np_modeling_amd', 'csrc', '*.h'))
                   + glob.glob(os.path.join(root, 'np_modeling_amd', 'csrc', '*.cpp')) + glob.glob(os.path.join(root, 'include', '*.h')))
    for path in files:
        digest.update(os.path.basename(path).encode())
        with open(path, 'rb') as f:
            digest.update(f.read())
    return digest.hexdigest()[:16]


def library_is_current() -> bool:
    """Whether the built libnpm_hip.so is at least as new as every source it is made of (the Makefile's own rule)."""
    import glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(LIB_PATH):
        return False
    built = os.path.getmtime(LIB_PATH)
    sources = glob.glob(os.path.join(root, 'np_modeling_amd', 'csrc', '*.hip')) + glob.glob(os.path.join(root, 'np_modeling_amd', 'csrc', '*.h')) \
        + [os.path.join(root, 'include', 'npm_hip.h')]
    return all(os.path.getmtime(p) <= built for p in sources)


def lib(ordered: bool = True):
    """The bound, device-initialised library.  Raises if there is no library or no GPU.  ``ordered=False``: the caller
    touches nothing a queued parameter update reads or writes (device.UpdateQueue) and need not wait for the queue."""
    global _LIB, _DEVICE
    if ordered and _ORDER_HOOK is not None:
        _ORDER_HOOK()
    if _LIB is None:
        ipc_env_for_multi_rank()
        _LIB = load_library()
    if _DEVICE is None:
        ipc_env_for_multi_rank()
        count = C.c_int(0)
        _LIB.npm_device_count(C.byref(count))         # 0 devices: npm_init below reports it with the HIP error text
        device = pick_device(count.value, os.environ)
        rc = _LIB.npm_init(device)
        if rc != 0:
            msg = _LIB.npm_last_error()
            raise NpmError(f'npm_init({device}) failed with code {rc}: '
                           f'{msg.decode() if msg else "?"} -- an MI355X is required, there is no CPU fallback')
        _DEVICE = device
        for item in filter(None, os.environ.get('NPM_TUNE', '').split(',')):     # e.g. NPM_TUNE=0=2 (A/B experiments)
            knob, value = item.split('=')
            check(_LIB.npm_set_tuning(int(knob), int(value)), 'npm_set_tuning')
        if os.environ.get('NPM_MATH'):
            set_math(os.environ['NPM_MATH'])
    return _LIB


def ipc_env_for_multi_rank(env=None) -> bool:
    """One rank of several: RCCL shares device buffers between the processes of a node, and this pool's host driver only
    supports dmabuf IPC -- without ``HSA_ENABLE_IPC_MODE_LEGACY=0`` it fails with ``hipIpcGetMemHandle: invalid argument``.
    The runtime reads the variable when it initialises, so it is set (unless the user chose a value) before the device
    library is loaded / initialised, whoever launched the ranks (np_modeling_amd/launch.py exports it too).  Returns
    whether this process is one of several ranks."""
    env = os.environ if env is None else env
    if int(env.get('WORLD_SIZE', '1') or '1') <= 1:
        return False
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    return True


def pick_device(visible: int, env) -> int:
    """Which of the ``visible`` HIP devices this process drives.

    ``NPM_DEVICE`` names it outright.  Otherwise one process per GPU: launchers either show every rank all GPUs of
    the node (the self-launcher, ``torch.distributed.run``) -- then ``LOCAL_RANK`` is the device index -- or mask
    the devices per rank (``HIP_VISIBLE_DEVICES`` / ``ROCR_VISIBLE_DEVICES`` = one GPU each) -- then the only
    visible device, index 0, is this rank's, whatever ``LOCAL_RANK`` says.  A mask that leaves several devices but
    fewer than ``LOCAL_RANK + 1`` is a launch error and reported as one (two ranks would share a GPU silently)."""
    if env.get('NPM_DEVICE'):
        return int(env['NPM_DEVICE'])
    local = int(env.get('LOCAL_RANK', '0'))
    if visible <= 1:
        return 0
    if local >= visible:
        raise NpmError(f'LOCAL_RANK={local} but only {visible} HIP devices are visible: show every rank all GPUs of '
                       'the node, or exactly one each (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES), or set NPM_DEVICE')
    return local


MATH_MODES = {'f32': 0, 'bf16x3_fast': 1, 'bf16x3': 2, 'f16x2': 3}      # include/npm_hip.h NPM_MATH_*


def parity_bounds() -> dict:
    """{mode: (REL, SCALED)} -- the parity contract of each math mode as include/npm_hip.h states it (NPM_PARITY_*): the
    header is the one place the numbers live; tests/test_gpu_parity.py asserts them and bench.py reports a throughput line
    only for a mode that has them."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, 'include', 'npm_hip.h')) as f:
        found = dict(re.findall(r'#define\s+NPM_PARITY_(\w+)\s+([0-9.eE+-]+)', f.read()))
    out = {}
    for mode in MATH_MODES:
        rel, scaled = found.get('REL_' + mode.upper()), found.get('SCALED_' + mode.upper())
        if rel is not None and scaled is not None:
            out[mode] = (float(rel), float(scaled))
    return out


def current_math() -> str:
    """The math mode the library is in (``npm_get_math``: one cheap call; a mode set through ``NPM_TUNE=10=<mode>``
    or a direct ``npm_set_tuning`` is seen too).  'f32' before the library is bound."""
    if _LIB is None or _DEVICE is None:
        return 'f32'
    value = _LIB.npm_get_math()
    return next(name for name, v in MATH_MODES.items() if v == value)


def set_math(mode: str) -> None:
    """Arithmetic of the matrix products: 'f32' (exact-fp32 MFMA, default), 'bf16x3' (three-way bf16 operand split
    on the bf16 matrix pipe, fp32-class error), 'bf16x3_fast' (same, one accumulator) or 'f16x2' (two-way fp16 split with
    row scaling on the f16 matrix pipe: fp32-class error row-normwise; the bf16 split where it does not apply);
    include/npm_hip.h."""
    if mode not in MATH_MODES:
        raise ValueError(f'unknown math mode {mode!r}: expected one of {sorted(MATH_MODES)}')
    check(lib().npm_set_math(MATH_MODES[mode]), 'npm_set_math')


def get_math() -> str:
    value = lib().npm_get_math()
    return next(name for name, v in MATH_MODES.items() if v == value)


def last_math() -> str:
    """The arithmetic the most recent matrix-product launch actually ran (include/npm_hip.h npm_last_math)."""
    value = lib().npm_last_math()
    return next(name for name, v in MATH_MODES.items() if v == value)


def last_attn_kernel() -> str:
    """What the most recent fused attention call launched (include/npm_hip.h npm_last_attn_kernel)."""
    return lib().npm_last_attn_kernel().decode()


def comm_lib():
    global _COMM_LIB
    if _COMM_LIB is None:
        _COMM_LIB = load_comm_library()
    return _COMM_LIB


def check(rc: int, what: str = '') -> None:
    if rc != 0:
        msg = _LIB.npm_last_error() if _LIB is not None else b''
        raise NpmError(f'{what or "npm call"} failed ({rc}): {msg.decode() if msg else ""}', rc)


def check_comm(rc: int, what: str = '') -> None:
    if rc != 0:
        msg = _COMM_LIB.npm_comm_last_error() if _COMM_LIB is not None else b''
        raise NpmError(f'{what or "npm_comm call"} failed ({rc}): {msg.decode() if msg else ""}')


def device_index() -> Optional[int]:
    return _DEVICE


def visible_devices() -> int:
    """HIP devices this process can see (after any HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES mask)."""
    count = C.c_int(0)
    (load_library() if _LIB is None else _LIB).npm_device_count(C.byref(count))
    return count.value
