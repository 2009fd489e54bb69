"""Runs the PRODUCT's layers over a tests/refshapes.py case and returns what the reference's tests look at, next to the
reference's own arrays from tests/golden/ref_*.npz.  Used on the GPU (tests/test_gpu_refshapes.py,
tests/test_gpu_reference_form.py) and on the host simulator (tests/test_host_logic.py)."""

import copy

import numpy as np

import refshapes as R
from conftest import load_golden


def build_layer(npm, case):
    L = npm.layers
    if case['kind'] == 'mha':
        return L.MultiHeadAttention(num_heads=case['heads'])
    if case['kind'] == 'encoder':
        return L.TransformerEncoder(num_heads=case['heads'], hidden_units=case['hidden'], norm_first=case['norm_first'])
    if case['kind'] == 'decoder':
        return L.TransformerDecoder(num_heads=case['heads'], hidden_units=case['hidden'], norm_first=case['norm_first'])
    if case['kind'] == 'dense':
        return L.Dense(units=case['units'])
    if case['kind'] == 'softmax':
        return L.Softmax()
    if case['kind'] == 'layernorm':
        return L.LayerNormalization()
    return L.Conv2D(channels=case['channels'], kernel_size=case['k'])


def attention_layers(layer, case):
    if case['kind'] == 'mha':
        return [layer]
    if case['kind'] == 'encoder':
        return [layer._self_attention]
    return [layer._self_attention, layer._cross_attention]


def run(npm, name, after_forward=None, after_backward=None):
    """Returns (got, ref): dicts name -> array of everything the fixture holds a reference value for.
    ``after_forward(layer)`` / ``after_backward(layer)`` are called right after the respective product call (the GPU
    tests assert there which kernel ran)."""
    case, g = R.CASES[name], load_golden(name)
    inp = R.draw_inputs(case)
    for k, v in inp.items():
        assert R.crc(v) == int(g[k + '_crc']), f'{name}: regenerated {k} is not the array the fixture was made from'
    layer = build_layer(npm, case)
    got, ref = {}, {}
    if case['kind'] == 'conv':
        y = layer(inp['x'])                                     # np.random.seed(0) + the two draws above: the reference's init
        np.testing.assert_array_equal(np.asarray(layer.w), g['w0'])
        np.testing.assert_array_equal(np.asarray(layer.b), g['b0'])
        got['y'], ref['y'] = np.asarray(y), g['y']
        dy = R.mse_grad(g['y'], inp['targets'])
        assert R.crc(dy) == int(g['dy_crc'])
        rec, recorder = copy.deepcopy(layer), R.GradRecorder()
        dx = np.asarray(rec(dy, backprop=True, optimizer_=recorder))
        got.update(dx_samples=dx[list(R.CONV_DX_SAMPLES)], dx_batch_sum=dx.astype(np.float64).sum(axis=0),
                   dw=recorder.grads[(id(rec), '_w')], db=recorder.grads[(id(rec), '_b')])
        w, b = layer.w, layer.b                                 # aliases taken before backward see the update (conv_test.py:57-58)
        layer(dy, backprop=True, learning_rate=case['lr'])
        got.update(w1=np.asarray(w), b1=np.asarray(b))
        for k in ('dx_samples', 'dx_batch_sum', 'dw', 'db', 'w1', 'b1'):
            ref[k] = g[k]
        return got, ref
    if case['kind'] == 'dense':                                 # mlp_test.py:35-94
        y = layer(inp['x'])
        lin = layer.linear
        w, b = lin.w, lin.b                                     # aliases taken before backward (mlp_test.py:50-51)
        np.testing.assert_array_equal(np.asarray(w), g['w0'])
        np.testing.assert_array_equal(np.asarray(b), g['b0'])
        dy = R.mse_grad(g['y'], inp['targets'])
        assert R.crc(dy) == int(g['dy_crc'])
        rec, recorder = copy.deepcopy(layer), R.GradRecorder()
        rec(dy, backprop=True, optimizer_=recorder)
        got.update(y=np.asarray(y), dw=recorder.grads[(id(rec.linear), '_w')], db=recorder.grads[(id(rec.linear), '_b')])
        got['dx'] = np.asarray(layer(dy, backprop=True, learning_rate=case['lr']))
        got.update(w1=np.asarray(w), b1=np.asarray(b))
        for k in got:
            ref[k] = g[k]
        return got, ref
    if case['kind'] == 'softmax':                               # activations_test.py:11-32
        got['y'] = np.asarray(layer(inp['x']))
        dy = R.mse_grad(g['y'], inp['targets'])
        assert R.crc(dy) == int(g['dy_crc'])
        got['dx'] = np.asarray(layer(dy, backprop=True))
        return got, {k: g[k] for k in got}
    if case['kind'] == 'layernorm':                             # normalizations_test.py:37-90
        layer(inp['x'])
        params = R.bound_params(case)
        layer._gamma, layer._beta, layer._epsilon = params['gamma'].copy(), params['beta'].copy(), R.LN_EPS
        got['z'] = np.asarray(layer(inp['x']))
        dz = R.mse_grad(g['z'], inp['targets'])
        assert R.crc(dz) == int(g['dy_crc'])
        rec, recorder = copy.deepcopy(layer), R.GradRecorder()
        rec(dz, backprop=True, optimizer_=recorder)
        got.update(dgamma=recorder.grads[(id(rec), '_gamma')], dbeta=recorder.grads[(id(rec), '_beta')])
        got['dx'] = np.asarray(layer(dz, backprop=True, learning_rate=case['lr']))
        got.update(gamma1=np.asarray(layer._gamma), beta1=np.asarray(layer._beta))
        return got, {k: g[k] for k in got}
    args = (inp['query'], inp['kv']) if 'kv' in inp else (inp['query'],)
    layer(*args)                                                # lazy initialisation (draws discarded)
    params = R.bound_params(case)
    assert R.crc(np.concatenate([v.ravel() for v in params.values()])) == int(g['params_crc'])
    R.bind(layer, case, params)
    got['out'], ref['out'] = np.asarray(layer(*args)), g['out']
    if after_forward:
        after_forward(layer)
    dy = R.mse_grad(g['out'], inp['targets'])                   # the reference's dy: same backward input on both sides
    assert R.crc(dy) == int(g['dy_crc'])
    rec, recorder = copy.deepcopy(layer), R.GradRecorder()      # attentions_test.py:72: the test deep-copies the layer
    grads_in = rec(dy, backprop=True, optimizer_=recorder)
    if after_backward:
        after_backward(rec)
    if case['kind'] == 'mha':
        names = ('dquery', 'dkey', 'dvalue')
        if 'kv' not in inp:                                     # one tensor fed three times: only the sum is defined
            got['dquery+dkey+dvalue'] = sum(np.asarray(a, dtype=np.float64) for a in grads_in)
            ref['dquery+dkey+dvalue'] = sum(g[n].astype(np.float64) for n in names)
        else:
            got['dquery'], ref['dquery'] = np.asarray(grads_in[0]), g['dquery']
            got['dkey+dvalue'] = np.asarray(grads_in[1], dtype=np.float64) + np.asarray(grads_in[2])
            ref['dkey+dvalue'] = g['dkey'].astype(np.float64) + g['dvalue']
    elif case['kind'] == 'encoder':
        got['dx'], ref['dx'] = np.asarray(grads_in), g['dx']
    else:
        got['dq'], ref['dq'] = np.asarray(grads_in[0]), g['dq']
        got['dkv'], ref['dkv'] = np.asarray(grads_in[1]), g['dkv']
    for k, v in recorder.named(rec, case).items():
        if 'grad_' + k in g:                                    # stored for the MHA cases (the composites: small-shape fixtures)
            got['grad_' + k], ref['grad_' + k] = v, g['grad_' + k]
    if case.get('updated'):
        upd = copy.deepcopy(layer)
        upd(dy, backprop=True, learning_rate=case['lr'])
        for n in R.MHA_NAMES:
            got[n + '1'], ref[n + '1'] = np.asarray(getattr(upd, '_' + n)), g[n + '1']
    return got, ref


def compare(got, ref, tol=1e-5):
    """The repo's scaled metric (tests/conftest.py assert_close) on every entry; the key biases' gradient (zero in exact
    arithmetic) on the scale of the query biases'."""
    from conftest import assert_close
    for k in ref:
        if k.endswith('bk') and k.startswith('grad_'):
            scale = np.abs(ref[k[:-2] + 'bq']).max()
            np.testing.assert_allclose(np.asarray(got[k], dtype=np.float64), ref[k], rtol=0, atol=tol * scale, err_msg=k)
        else:
            assert_close(got[k], ref[k], tol=tol, what=k)
