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
