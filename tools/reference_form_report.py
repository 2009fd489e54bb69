"""Writes the table behind tests/test_gpu_reference_form.py (run on the GPU box):
    python tools/reference_form_report.py > profiles/r04_reference_form.md"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]

import np_modeling_amd as npm          # noqa: E402
import reference_form as RF            # noqa: E402

print('# The reference\'s own assertion form on the HIP path (exact-f32 MFMA), MI355X\n')
print('`violation` = max |got - ref| / (atol + rtol |ref|) over the tensor; <= 1 passes the reference test\'s `assert_allclose`.')
print('`ref` = the NumPy reference\'s own output (tests/golden/ref_*.npz), flows and constants of the cited reference tests;')
print('`fp32 floor` = the same figure for the NumPy oracle evaluated in fp32 end to end (taken only where the product exceeds 1).\n')
print('| case | quantity | rtol | atol | product violation | fp32 floor | verdict | constants from |')
print('|---|---|---|---|---|---|---|---|')
worst = 0.0
for name, q, rtol, atol, v, v32, verdict, where in RF.table(npm):
    worst = max(worst, v)
    print(f'| {name} | {q} | {rtol:g} | {atol:g} | {v:.3f} | {"-" if v32 is None else f"{v32:.3f}"} | {verdict} | {where} |')
print(f'\nworst violation: {worst:.3f}')
