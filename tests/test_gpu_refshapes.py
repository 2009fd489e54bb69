"""GPU: the product's layers against the REFERENCE'S OWN OUTPUTS at the reference's own test shapes and at every head size
the fused attention kernels take (tests/golden/ref_*.npz; recipes and rationale in tests/refshapes.py).  In the default
exact-f32 arithmetic the tests assert that the fused kernels are what ran (npm_last_attn_kernel / npm_last_math), so
mha_fwd_kernel, mha_bwd16_kernel (head size 128 with saved scores: the BASELINE configs' path) and mha_bwd8_kernel (every
other head size, and the recomputing mode) are compared with reference arrays directly, not through the oracle; the other
choices of NPM_TUNE_ATTN_BWD16 (3: mha_bwd8_kernel always; 1 / 0: round 3's kernels) run the same fixtures in
test_reference_outputs_other_backward_kernels."""

import numpy as np
import pytest

import refshapes as R
import refshape_runner as RR

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def npm():
    import np_modeling_amd
    return np_modeling_amd


ATTENTION = [n for n, c in R.CASES.items() if c['kind'] in ('mha', 'encoder', 'decoder')]
FWD = 'mha_fwd8_kernel'            # the default forward (NPM_TUNE_ATTN_FWD8 = 2); the 4-wave mha_fwd_kernel: test_reference_outputs_other_forward


@pytest.mark.parametrize('name', ATTENTION)
def test_reference_outputs(npm, name, math_mode):
    case = R.CASES[name]
    d = case['feat'] // case['heads']
    fused = math_mode == 'f32'

    def after_forward(layer):
        for att in RR.attention_layers(layer, case):
            assert att._core is fused
        if fused:
            assert npm._C.last_attn_kernel() == f'{FWD} D={d} mask=0 scores={int(d >= 64)}', npm._C.last_attn_kernel()
            assert npm.last_math() == 'f32'

    def after_backward(layer):
        if fused:
            last = npm._C.last_attn_kernel()
            assert last == f'{"mha_bwd16_kernel" if d == 128 else "mha_bwd8_kernel"} D={d} mask=0 scores={int(d >= 64)}', last

    got, ref = RR.run(npm, name, after_forward, after_backward)
    RR.compare(got, ref, tol=1e-5)


@pytest.mark.parametrize('name', [n for n in ATTENTION if R.CASES[n]['kind'] == 'mha'])
@pytest.mark.parametrize('knob', [3, 1, 0])
def test_reference_outputs_other_backward_kernels(npm, name, knob):
    """The other choices of NPM_TUNE_ATTN_BWD16 stay selectable and run the same fixtures: 3 = mha_bwd8_kernel always,
    1 = mha_bwd16_kernel (head size 128) / the 4-wave mha_bwd_kernel, 0 = the 4-wave kernel."""
    from np_modeling_amd import _C
    d = R.CASES[name]['feat'] // R.CASES[name]['heads']
    from np_modeling_amd import device as D
    want = 'mha_bwd8_kernel' if knob == 3 else ('mha_bwd16_kernel' if (knob == 1 and d == 128) else 'mha_bwd_kernel')
    _C.check(_C.lib().npm_set_tuning(14, knob), 'npm_set_tuning')
    saved = D.ATTN_SAVE_SCORES
    D.ATTN_SAVE_SCORES = True                      # every head size with saved scores (the default keeps them from 64 up)
    try:
        got, ref = RR.run(npm, name, after_backward=lambda layer: _assert_kernel(npm, f'{want} D={d} mask=0 scores=1'))
    finally:
        D.ATTN_SAVE_SCORES = saved
        _C.check(_C.lib().npm_set_tuning(14, 2), 'npm_set_tuning')
    RR.compare(got, ref, tol=1e-5)


@pytest.mark.parametrize('name', [n for n in ATTENTION if R.CASES[n]['kind'] == 'mha'])
def test_reference_outputs_other_forward(npm, name):
    """The 4-wave 32x32x2 forward (NPM_TUNE_ATTN_FWD8 = 0) stays selectable and runs the same fixtures."""
    from np_modeling_amd import _C
    d = R.CASES[name]['feat'] // R.CASES[name]['heads']
    from np_modeling_amd import device as D
    _C.check(_C.lib().npm_set_tuning(17, 0), 'npm_set_tuning')
    saved = D.ATTN_SAVE_SCORES
    D.ATTN_SAVE_SCORES = True
    try:
        got, ref = RR.run(npm, name, after_forward=lambda layer: _assert_kernel(npm, f'mha_fwd_kernel D={d} mask=0 scores=1'))
    finally:
        D.ATTN_SAVE_SCORES = saved
        _C.check(_C.lib().npm_set_tuning(17, 2), 'npm_set_tuning')
    RR.compare(got, ref, tol=1e-5)


def _assert_kernel(npm, want):
    assert npm._C.last_attn_kernel() == want, npm._C.last_attn_kernel()


@pytest.mark.parametrize('name', [n for n in ATTENTION if R.CASES[n]['kind'] == 'mha'])
def test_reference_outputs_recomputing_backward(npm, name):
    """The memory-lean mode (NPM_ATTN_SAVE_SCORES=0): log-sum-exp only, the backward recomputes q.k."""
    from np_modeling_amd import device as D
    case = R.CASES[name]
    d = case['feat'] // case['heads']
    saved = D.ATTN_SAVE_SCORES
    D.ATTN_SAVE_SCORES = False
    try:
        def after_forward(layer):
            assert layer._core and layer._raw_scores is None
            assert npm._C.last_attn_kernel() == f'{FWD} D={d} mask=0 scores=0'

        def after_backward(layer):
            assert npm._C.last_attn_kernel() == f'mha_bwd8_kernel D={d} mask=0 scores=0'

        got, ref = RR.run(npm, name, after_forward, after_backward)
    finally:
        D.ATTN_SAVE_SCORES = saved
    RR.compare(got, ref, tol=1e-5)


@pytest.mark.parametrize('name', ['ref_conv_k3', 'ref_dense', 'ref_softmax', 'ref_layernorm'])
def test_reference_flow_of_the_single_layers(npm, name, math_mode):
    """conv_test.py:37-107 (x [64, 32, 16, 32] -> 16 channels, k 3), mlp_test.py:35-94, activations_test.py:11-32,
    normalizations_test.py:37-90: the layer's own seeded initialisation (asserted bit-equal to the reference's), dy = the MSE
    gradient, aliases of w / b taken before backward see the update."""
    got, ref = RR.run(npm, name)
    RR.compare(got, ref, tol=1e-5 if name == 'ref_conv_k3' else 2e-6)
