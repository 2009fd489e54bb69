"""CPU: the oracle against the reference's outputs at the reference's OWN test shapes and at the head sizes the fused
attention kernels take (tests/golden/ref_*.npz, recipes in tests/refshapes.py, written by oracle/make_golden.py from the
real reference).  Also checks that the rebuilt inputs / parameters are the ones the fixtures were made from (CRC-32)."""

import numpy as np
import pytest

import refshapes as R
from conftest import load_golden
from oracle import np_oracle as O


def close_scaled(a, b, tol=1e-5, what=''):
    b = np.asarray(b, dtype=np.float64)
    np.testing.assert_allclose(np.asarray(a, dtype=np.float64), b, rtol=tol, atol=tol * np.abs(b).max(), err_msg=what)


def close_grad(grads, g, k, tol=1e-5):
    """A parameter gradient against the fixture's.  The key bias's gradient is identically zero in exact arithmetic
    (a constant added to every key shifts each softmax row uniformly), so what both sides hold is rounding noise: it is
    compared on the scale of the query bias's gradient instead of its own."""
    if k.endswith('bk'):
        ref = np.asarray(g['grad_' + k], dtype=np.float64)
        scale = np.abs(g['grad_' + k[:-2] + 'bq']).max()
        assert np.abs(ref).max() <= tol * scale
        np.testing.assert_allclose(np.asarray(grads[k], dtype=np.float64), ref, rtol=0, atol=tol * scale, err_msg=k)
    else:
        close_scaled(grads[k], g['grad_' + k], tol, what=k)


def _inputs(name):
    case, g = R.CASES[name], load_golden(name)
    inp = R.draw_inputs(case)
    for k, v in inp.items():
        assert R.crc(v) == int(g[k + '_crc']), f'{name}: regenerated {k} differs from the one the fixture was made from'
    return case, g, inp


MHA = [n for n, c in R.CASES.items() if c['kind'] == 'mha']


@pytest.mark.parametrize('name', MHA)
@pytest.mark.parametrize('verbatim', [True, False])
def test_mha(name, verbatim):
    case, g, inp = _inputs(name)
    p = R.bound_params(case)
    assert R.crc(np.concatenate([v.ravel() for v in p.values()])) == int(g['params_crc'])
    out, cache = O.mha_fwd(p, inp['query'], inp.get('kv'), verbatim=verbatim)
    close_scaled(out, g['out'], what='out')
    dy = R.mse_grad(g['out'], inp['targets'])
    assert R.crc(dy) == int(g['dy_crc'])
    (dq, dk, dv), grads = O.mha_bwd(p, cache, dy, verbatim=verbatim)
    close_scaled(dq, g['dquery'], what='dquery')
    close_scaled(dk, g['dkey'], what='dkey')
    close_scaled(dv, g['dvalue'], what='dvalue')
    for n in R.MHA_NAMES:
        close_grad(grads, g, n)
        if case.get('updated'):
            close_scaled(O.sgd_step(p[n], grads[n], case['lr']), g[n + '1'], what=n + '1')
    # the softmaxes these fixtures exercise are not one-hot (the point of the 1/sqrt(fan_in) rebinding)
    assert np.median(cache['scores'].max(axis=-1)) < 0.5


@pytest.mark.parametrize('name', ['ref_encoder_prenorm', 'ref_encoder_postnorm'])
def test_encoder(name):
    case, g, inp = _inputs(name)
    p = R.bound_params(case)
    nf = case['norm_first']
    out, cache = O.encoder_fwd(p, inp['query'], nf, eps=R.LN_EPS)
    close_scaled(out, g['out'], what='out')
    dy = R.mse_grad(g['out'], inp['targets'])
    assert R.crc(dy) == int(g['dy_crc'])
    dx, grads = O.encoder_bwd(p, cache, dy, nf, eps=R.LN_EPS)
    close_scaled(dx, g['dx'], what='dx')
    assert len(grads) == 16 and not any(k.startswith('grad_') for k in g)      # parameter gradients: encoder_*.npz (small shape)


@pytest.mark.parametrize('name', ['ref_decoder_prenorm', 'ref_decoder_postnorm'])
def test_decoder(name):
    case, g, inp = _inputs(name)
    p = R.bound_params(case)
    nf = case['norm_first']
    out, cache = O.decoder_fwd(p, inp['query'], inp['kv'], nf, eps=R.LN_EPS)
    close_scaled(out, g['out'], what='out')
    dy = R.mse_grad(g['out'], inp['targets'])
    (dq, dkv), grads = O.decoder_bwd(p, cache, dy, nf, eps=R.LN_EPS)
    close_scaled(dq, g['dq'], what='dq')
    close_scaled(dkv, g['dkv'], what='dkv')
    assert len(grads) == 26 and not any(k.startswith('grad_') for k in g)      # parameter gradients: decoder_*.npz (small shape)


def test_conv_reference_shape():
    """conv_test.py:37-107: x [64, 32, 16, 32] -> 16 channels, k 3, the layer's own initialisation."""
    case, g, inp = _inputs('ref_conv_k3')
    y, pre = O.conv_layer_fwd(inp['x'], g['w0'], g['b0'])
    close_scaled(y, g['y'], tol=1e-6, what='y')
    dy = R.mse_grad(g['y'], inp['targets'])
    assert R.crc(dy) == int(g['dy_crc'])
    dx, dw, db = O.conv_layer_bwd(inp['x'], g['w0'], pre, dy)
    close_scaled(dx[list(R.CONV_DX_SAMPLES)], g['dx_samples'], tol=1e-6, what='dx samples')
    close_scaled(dx.sum(axis=0), g['dx_batch_sum'], tol=1e-6, what='dx summed over the batch')
    close_scaled(dw, g['dw'], tol=1e-6, what='dw')
    close_scaled(db, g['db'], tol=1e-6, what='db')
    close_scaled(O.sgd_step(g['w0'], dw, case['lr']), g['w1'], tol=1e-6)
    close_scaled(O.sgd_step(g['b0'], db, case['lr']), g['b1'], tol=1e-6)
