"""GPU: the product's layers against the REFERENCE'S OWN OUTPUTS at the reference's own test shapes and at every head size
the fused attention kernels take (tests/golden/ref_*.npz; recipes and rationale in tests/refshapes.py).  In the default
exact-f32 arithmetic the tests assert that the fused kernels are what ran (npm_last_attn_kernel / npm_last_math), so
mha_fwd_kernel, mha_bwd16_kernel (head size 128, saved scores) and mha_bwd_kernel (the other head sizes, and the
recomputing mode) are compared with reference arrays directly, not through the oracle."""

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


@pytest.mark.parametrize('name', ATTENTION)
def test_reference_outputs(npm, name, math_mode):
    case = R.CASES[name]
    d = case['feat'] // case['heads']
    fused = math_mode == 'f32'

    def after_forward(layer):
        for att in RR.attention_layers(layer, case):
            assert att._core is fused
        if fused:
            assert npm._C.last_attn_kernel().startswith(f'mha_fwd_kernel D={d} mask=0 scores=1'), npm._C.last_attn_kernel()
            assert npm.last_math() == 'f32'

    def after_backward(layer):
        if fused:
            last = npm._C.last_attn_kernel()
            assert last == (f'mha_bwd16_kernel D={d} mask=0 scores=1' if d == 128 else f'mha_bwd_kernel D={d} mask=0 scores=1'), last

    got, ref = RR.run(npm, name, after_forward, after_backward)
    RR.compare(got, ref, tol=1e-5)


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
            assert npm._C.last_attn_kernel() == f'mha_fwd_kernel D={d} mask=0 scores=0'

        def after_backward(layer):
            assert npm._C.last_attn_kernel() == f'mha_bwd_kernel D={d} mask=0 scores=0'

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
