"""pytest configuration: markers, repo root on sys.path, golden-fixture loader."""

import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    with np.load(os.path.join(GOLDEN_DIR, name + '.npz')) as data:
        return {k: data[k] for k in data.files}


@pytest.fixture
def golden():
    return load_golden


def assert_close(got, ref, tol=1e-5, what=''):
    """Parity metric of this repo: |got - ref| <= tol * (|ref| + max|ref|) elementwise.

    fp32 GEMMs reorder sums, so an element's error scales with the magnitude of the tensor
    (the terms that cancelled), not with the element itself; a purely relative test would
    be meaningless on near-zero elements.  BASELINE.json's bound is 1e-4; most tests use a
    tighter tol."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, f'{what}: shape {got.shape} vs {ref.shape}'
    scale = np.abs(ref).max() if ref.size else 0.0
    np.testing.assert_allclose(got, ref, rtol=tol, atol=tol * scale + 1e-30, err_msg=what)


MATH_MODES = ['f32', 'bf16x3', 'bf16x3_fast', 'f16x2']


@pytest.fixture(params=MATH_MODES)
def math_mode(request):
    """Runs the test once per arithmetic of the matrix products (include/npm_hip.h npm_set_math); back to the
    default afterwards."""
    import np_modeling_amd
    np_modeling_amd.set_math(request.param)
    yield request.param
    np_modeling_amd.set_math('f32')


@pytest.fixture(autouse=True)
def _math_mode_back_to_default():
    """Whatever arithmetic a test selected (and however it ended), the next test starts from 'f32'."""
    yield
    from np_modeling_amd import _C
    if _C._LIB is not None and _C.current_math() != 'f32':
        _C.set_math('f32')
