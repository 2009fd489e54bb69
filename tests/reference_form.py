"""The reference's OWN assertion form applied to this build: ``|got - ref| <= atol + rtol * |ref|`` with the constants of
each reference test, on fixtures that follow each test's flow (tests/refshapes.py; dy is the MSE gradient, so gradients are
O(1e-5) and the absolute term dominates them -- that is how the reference tests are written).

``ref`` here is the NumPy reference's output (tests/golden/ref_*.npz) where the reference's tests have JAX/Flax (absent
from this image): the bound is the same, the comparison partner is the reference itself.

``violation`` = max over elements of |got - ref| / (atol + rtol |ref|): <= 1 passes the reference's assertion.  Where the
fp32 product exceeds 1, the same figure is taken for the ORACLE evaluated in fp32 end to end (oracle/np_oracle.py
``compute_in(np.float32)``: another legitimate fp32 implementation with another summation order, standing in for the fp32
JAX side of the reference's tests); the test then asserts product <= 2 x that noise floor."""

import numpy as np

import refshapes as R
from conftest import load_golden

# case -> (where the constants are in the reference, default (rtol, atol), {quantity: (rtol, atol) or None for default})
FORMS = {
    'ref_dense': ('layers/mlp_test.py:13,59,92-94', (1e-6, 1e-6), dict(y=None, dx=None, w1=None, b1=None)),
    'ref_conv_k3': ('layers/conv_test.py:13,71,105-107', (1e-6, 1e-6), dict(y=(1e-6, 3e-5), dx_samples=None, w1=None, b1=None)),
    'ref_softmax': ('layers/activations_test.py:32', (1e-5, 1e-5), dict(dx=None)),
    'ref_layernorm': ('layers/normalizations_test.py:38-39,60,73-90', (1e-6, 1e-6), dict(z=None, gamma1=None, beta1=None, dx=None)),
    'ref_mha_self_d16': ('layers/utils.py:13, layers/attentions_test.py:49,76-85', (1e-5, 2e-5),
                         {'out': None, 'dquery+dkey+dvalue': None, **{n + '1': None for n in R.MHA_NAMES}}),
    'ref_mha_self_d32': ('layers/utils.py:13', (1e-5, 2e-5), {'out': None, 'dquery+dkey+dvalue': None}),
    'ref_mha_cross_d64': ('layers/utils.py:13', (1e-5, 2e-5), {'out': None, 'dquery': None, 'dkey+dvalue': None}),
    'ref_mha_self_d128': ('layers/utils.py:13', (1e-5, 2e-5), {'out': None, 'dquery+dkey+dvalue': None}),
    'ref_encoder_prenorm': ('layers/transformer_test.py:99-100,139,156', (1e-5, 1e-5), dict(out=None, dx=None)),
    'ref_encoder_postnorm': ('layers/transformer_test.py:99-100,139,156', (1e-5, 1e-5), dict(out=None, dx=None)),
    'ref_decoder_prenorm': ('layers/transformer_test.py:160-161,203,218-219', (1e-5, 1e-5), dict(out=None, dq=None, dkv=None)),
    'ref_decoder_postnorm': ('layers/transformer_test.py:160-161,203,218-219', (1e-5, 1e-5), dict(out=None, dq=None, dkv=None)),
}


def violation(got, ref, rtol, atol) -> float:
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape
    return float((np.abs(got - ref) / (atol + rtol * np.abs(ref))).max())


def oracle_fp32(name):
    """The quantities of FORMS[name] from the oracle evaluated in fp32 end to end, on the same inputs."""
    from oracle import np_oracle as O
    case, g = R.CASES[name], load_golden(name)
    inp = R.draw_inputs(case)
    f = np.float32
    out = {}
    with O.compute_in(np.float32):
        kind = case['kind']
        if kind == 'dense':
            y, pre = O.dense_fwd(inp['x'], g['w0'], g['b0'])
            dx, dw, db = O.dense_bwd(inp['x'], g['w0'], pre, R.mse_grad(g['y'], inp['targets']))
            out.update(y=y, dx=dx, w1=O.sgd_step(g['w0'], dw, case['lr']), b1=O.sgd_step(g['b0'], db, case['lr']))
        elif kind == 'conv':
            y, pre = O.conv_layer_fwd(inp['x'], g['w0'], g['b0'])
            dx, dw, db = O.conv_layer_bwd(inp['x'], g['w0'], pre, R.mse_grad(g['y'], inp['targets']))
            out.update(y=y, dx_samples=dx[list(R.CONV_DX_SAMPLES)], w1=O.sgd_step(g['w0'], dw, case['lr']),
                       b1=O.sgd_step(g['b0'], db, case['lr']))
        elif kind == 'softmax':
            y = O.softmax_fwd(inp['x'])
            out.update(dx=O.softmax_bwd(y, R.mse_grad(g['y'], inp['targets'])))
        elif kind == 'layernorm':
            p = R.bound_params(case)
            z, cache = O.layernorm_fwd(inp['x'], p['gamma'], p['beta'], f(R.LN_EPS))
            dx, dgamma, dbeta = O.layernorm_bwd(inp['x'], p['gamma'], f(R.LN_EPS), cache, R.mse_grad(g['z'], inp['targets']))
            out.update(z=z, dx=dx, gamma1=O.sgd_step(p['gamma'], dgamma, case['lr']), beta1=O.sgd_step(p['beta'], dbeta, case['lr']))
        else:
            p = R.bound_params(case)
            dy = R.mse_grad(g['out'], inp['targets'])
            if kind == 'mha':
                o, cache = O.mha_fwd(p, inp['query'], inp.get('kv'))
                (dq, dk, dv), grads = O.mha_bwd(p, cache, dy)
                out['out'] = o
                if 'kv' in inp:
                    out['dquery'], out['dkey+dvalue'] = dq, dk + dv
                else:
                    out['dquery+dkey+dvalue'] = dq + dk + dv
                for n in R.MHA_NAMES:
                    out[n + '1'] = O.sgd_step(p[n], grads[n], case['lr'])
            elif kind == 'encoder':
                o, cache = O.encoder_fwd(p, inp['query'], case['norm_first'], eps=f(R.LN_EPS))
                dx, _ = O.encoder_bwd(p, cache, dy, case['norm_first'], eps=f(R.LN_EPS))
                out.update(out=o, dx=dx)
            else:
                o, cache = O.decoder_fwd(p, inp['query'], inp['kv'], case['norm_first'], eps=f(R.LN_EPS))
                (dq, dkv), _ = O.decoder_bwd(p, cache, dy, case['norm_first'], eps=f(R.LN_EPS))
                out.update(out=o, dq=dq, dkv=dkv)
    for k, v in out.items():
        assert np.asarray(v).dtype == np.float32, (name, k, np.asarray(v).dtype)
    return out


def table(npm, names=None):
    """Rows (case, quantity, rtol, atol, product violation, fp32-oracle violation or None, verdict) for every asserted quantity."""
    import refshape_runner as RR
    rows = []
    for name in names or FORMS:
        where, (rtol0, atol0), quantities = FORMS[name]
        got, ref = RR.run(npm, name)
        floor = None
        for q, tol in quantities.items():
            rtol, atol = tol or (rtol0, atol0)
            v = violation(got[q], ref[q], rtol, atol)
            v32 = None
            if v > 1.0:
                floor = floor or oracle_fp32(name)
                v32 = violation(floor[q], ref[q], rtol, atol)
            verdict = 'passes' if v <= 1.0 else ('within 2x of the fp32 floor' if v <= 2.0 * v32 else 'FAILS')
            rows.append((name, q, rtol, atol, v, v32, verdict, where))
    return rows
