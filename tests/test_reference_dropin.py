"""CPU, build container only: the reference's own unchanged ``optimizer.py`` / ``loss.py`` / ``train.py`` drive the
device layers after ``np_modeling_amd.install()`` (BASELINE.json north_star: "Trainer/optimizer/loss drop in
unchanged"; reference train.py:20-46, optimizer.py:13-18,53-67, loss.py:21-39, train_test.py:14-81).

Skipped where /root/reference does not exist (the GPU box).  The flows run in a child interpreter
(tests/dropin_worker.py) so that the reference's top-level module names are bound only there."""

import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = os.environ.get('NPM_REFERENCE', '/root/reference')


@pytest.mark.skipif(not os.path.isfile(os.path.join(REFERENCE, 'train.py')),
                    reason='the reference checkout is only present in the build container')
def test_reference_modules_drive_installed_layers():
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1', NPM_REFERENCE=REFERENCE)
    env.pop('PYTHONPATH', None)
    proc = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'dropin_worker.py')], env=env,
                          capture_output=True, text=True, timeout=900)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-4000:]
    assert 'DROPIN OK' in proc.stdout
    assert 'train_mlp[sgd]' in proc.stdout and 'train_mlp[adam]' in proc.stdout
