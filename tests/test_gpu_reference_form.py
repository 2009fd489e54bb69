"""GPU: the reference's own assertion form, with its own constants, on the reference tests' own flows
(tests/reference_form.py).  north_star: "the repo's Jax/Flax-gated layer tests pass" -- JAX is not in this image, so the
partner of each comparison is the NumPy reference's output instead of Flax's; the bound is the reference test's."""

import pytest

import reference_form as RF

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def npm():
    import np_modeling_amd
    return np_modeling_amd


@pytest.mark.parametrize('name', list(RF.FORMS))
def test_reference_assertion_form(npm, name):
    rows = RF.table(npm, [name])
    bad = [r for r in rows if r[6] == 'FAILS']
    assert not bad, bad
