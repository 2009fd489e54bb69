"""GPU: the device layers behind the reference's Layer API, against the reference's own
outputs (tests/golden, produced by oracle/make_golden.py) and against the oracle at the
shapes the reference's tests use."""

import copy

import numpy as np
import pytest

from conftest import assert_close, load_golden
from oracle import np_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def npm():
    import np_modeling_amd
    return np_modeling_amd


def rand(shape):
    return np.random.normal(size=shape).astype(np.float32)


# ---- Dense / Linear ---------------------------------------------------------------------
@pytest.mark.parametrize('name', ['dense', 'linear'])
def test_dense_golden(npm, name, math_mode):
    """Flow of reference layers/mlp_test.py:35-94: aliases of w and b taken before backward
    observe the in-place SGD update."""
    g = load_golden(name)
    L = npm.layers
    layer = L.Dense(units=16) if name == 'dense' else L.Linear(units=16)
    np.random.seed(0)
    x = rand([64, 32])
    y = layer(x)
    np.testing.assert_array_equal(x, g['x'])
    lin = layer.linear if name == 'dense' else layer
    w, b = lin.w, lin.b
    np.testing.assert_array_equal(np.asarray(w), g['w0'])       # same seeded draws as the reference
    np.testing.assert_array_equal(np.asarray(b), g['b0'])
    assert_close(y, g['y'], tol=2e-6)
    assert y.shape == (64, 16)
    dx = layer(g['dy'], backprop=True, learning_rate=float(g['lr']))
    assert dx.shape == (64, 32)
    assert_close(dx, g['dx'], tol=2e-6)
    assert_close(w, g['w1'], tol=2e-6)                           # alias sees the update
    assert_close(b, g['b1'], tol=2e-6)
    assert lin.w is w


def test_call_protocol(npm):
    L = npm.layers
    layer = L.Dense(units=8)
    x = rand([4, 6])
    layer(x)
    with pytest.raises(ValueError):
        layer(rand([4, 8]), backprop=True, learning_rate=0.1, optimizer_=npm.optimizer.SGDOptimizer(0.1))
    with pytest.raises(AssertionError):
        L.Dense(units=3).linear                                  # property asserts _initialized
    # Linear backward is 2-D only, like reference mlp.py:33
    lin = L.Linear(units=5)
    lin(rand([2, 3, 4]))
    with pytest.raises(AssertionError):
        lin(rand([2, 3, 5]), backprop=True, learning_rate=0.1)
    # first call may be a backprop call: initialize still fires (layer.py:33-35)
    relu = L.ReLU()
    assert relu._initialized is False
    relu(x)
    assert relu._initialized is True


def test_array_likes_assigned_into_private_attrs(npm):
    """Weight binders assign plain arrays into _w/_b (reference layers/utils.py:77-88)."""
    L = npm.layers
    layer = L.Dense(units=16)
    x = rand([10, 12])
    layer(x)
    w = rand([12, 16])
    b = rand([16])
    layer._linear._w = w.tolist()
    layer._linear._b = b
    y = layer(x)
    want, pre = O.dense_fwd(x.astype(np.float64), w.astype(np.float64), b.astype(np.float64))
    assert_close(y, want, tol=2e-6)
    dy = rand([10, 16])
    dx = layer(dy, backprop=True, learning_rate=0.5)
    wdx, wdw, wdb = O.dense_bwd(x.astype(np.float64), w.astype(np.float64), pre, dy.astype(np.float64))
    assert_close(dx, wdx, tol=2e-6)
    assert_close(layer.linear.w, w - 0.5 * wdw, tol=2e-6)
    assert_close(layer.linear.b, b - 0.5 * wdb, tol=2e-6)


def test_reference_shaped_optimizer_contract(npm):
    """An optimizer written like the reference's (getattr -> `v -= lr * g` -> setattr, fp64
    host moments for Adam) drives the device layers unchanged."""
    class HostSGD:
        def __init__(self, lr):
            self.lr = lr

        def update(self, obj, attribute, gradient):
            variable = getattr(obj, attribute)
            variable -= self.lr * gradient
            setattr(obj, attribute, variable)

    L = npm.layers
    np.random.seed(3)
    a, b = L.Dense(units=7), L.Dense(units=7)
    x, dy = rand([9, 5]), rand([9, 7])
    np.random.seed(4)
    a(x)
    np.random.seed(4)
    b(x)
    a(dy, backprop=True, optimizer_=HostSGD(0.1))
    b(dy, backprop=True, optimizer_=npm.optimizer.SGDOptimizer(0.1))
    np.testing.assert_array_equal(np.asarray(a.linear.w), np.asarray(b.linear.w))
    # Adam: same numbers as the oracle's restatement of reference optimizer.py:53-67
    np.random.seed(4)
    c = L.Dense(units=7)
    y = c(x)
    w0, b0 = np.asarray(c.linear.w).copy(), np.asarray(c.linear.b).copy()
    adam = npm.optimizer.AdamOptimizer(1e-2)
    c(dy, backprop=True, optimizer_=adam)
    _, pre = O.dense_fwd(x, w0, b0)
    _, dw, db = O.dense_bwd(x, w0, pre, dy)
    assert_close(c.linear.w, O.adam_step(w0, dw, {}, 1e-2), tol=2e-6)
    assert_close(c.linear.b, O.adam_step(b0, db, {}, 1e-2), tol=2e-6)


# ---- activations ---------------------------------------------------------------------------
def test_relu_golden(npm):
    g = load_golden('relu')
    relu = npm.layers.ReLU()
    y = relu(g['x'])
    np.testing.assert_array_equal(np.asarray(y), g['y'])
    dx = relu.backward(g['dy'])                       # no optimizer argument, as the reference
    np.testing.assert_array_equal(np.asarray(dx), g['dx'].astype(np.float32))
    with pytest.raises(AssertionError):
        relu.backward(g['dy'][:3])


def test_softmax_layer(npm):
    """reference layers/activations_test.py:11-32: [128, 128], backward via __call__ without optimizer."""
    np.random.seed(0)
    sm = npm.layers.Softmax()
    x, dy = rand([128, 128]), rand([128, 128])
    y = sm(x)
    assert_close(y, O.softmax_fwd(x.astype(np.float64)), tol=2e-6)
    dx = sm(dy, backprop=True)
    assert_close(dx, O.softmax_bwd(np.asarray(y), dy, verbatim=True), tol=5e-6)
    g = load_golden('softmax')
    sm3 = npm.layers.Softmax()
    assert_close(sm3(g['x']), g['y'], tol=2e-6)
    assert_close(sm3(g['dy'], backprop=True), g['dx'], tol=5e-6)


# ---- LayerNormalization ---------------------------------------------------------------------
@pytest.mark.parametrize('name', ['layernorm_2d', 'layernorm_3d'])
def test_layernorm_golden(npm, name):
    g = load_golden(name)
    eps = float(g['eps'])
    ln = npm.layers.LayerNormalization(epsilon=eps)
    np.random.seed(0)
    x = rand(g['x'].shape) * 2.0 + 0.5
    z = ln(x)
    np.testing.assert_array_equal(np.asarray(ln._gamma), g['gamma0'])
    np.testing.assert_array_equal(np.asarray(ln._beta), g['beta0'])
    assert_close(z, g['z'], tol=3e-6)
    gamma, beta = ln._gamma, ln._beta
    dx = ln(g['dz'], backprop=True, learning_rate=float(g['lr']))
    assert_close(dx, g['dx'], tol=5e-6)
    assert_close(gamma, g['gamma1'], tol=3e-6)
    assert_close(beta, g['beta1'], tol=3e-6)


def test_layernorm_rebinding_like_flax_binder(npm):
    """reference layers/utils.py:62-68 rebinds _gamma/_beta/_epsilon after the first call."""
    ln = npm.layers.LayerNormalization()
    x = rand([32, 128])
    ln(x)
    assert hasattr(ln, '_gamma') and hasattr(ln, '_beta') and hasattr(ln, '_epsilon')
    ln._gamma = np.ones(128, dtype=np.float32)
    ln._beta = np.zeros(128, dtype=np.float32)
    ln._epsilon = 1e-6
    z = ln(x)
    want, _ = O.layernorm_fwd(x.astype(np.float64), np.ones(128), np.zeros(128), 1e-6)
    assert_close(z, want, tol=3e-6)


def test_dropout_identity_and_mask(npm):
    L = npm.layers
    d0 = L.DropOut(0.0)
    x = npm.as_device(rand([8, 8]))
    assert d0(x) is x                                           # same object (normalizations.py:23)
    assert d0.backward(x) is x
    d = L.DropOut(0.5)
    np.random.seed(0)
    xs = rand([128, 32])
    np.random.seed(1)
    y = np.asarray(d(xs))
    np.random.seed(1)
    mask = np.random.binomial(n=1, p=0.5, size=xs.size).reshape(xs.shape)
    np.testing.assert_allclose(y, np.where(mask, xs / 0.5, 0.0), rtol=1e-6)
    np.testing.assert_allclose(np.asarray(d.backward(xs)), np.where(mask, xs / 0.5, 0.0), rtol=1e-6)


@pytest.mark.parametrize('shape,p', [([128, 32], 0.5), ([7, 13, 5], 0.1), ([1000003], 0.9), ([3], 0.25)])
def test_dropout_device_rng(npm, shape, p):
    """Masks drawn on the device (Philox4x32-10): bit-equal to the oracle's restatement of the generator (itself pinned by
    the published known-answer vectors), `_mask` readable like the reference's (normalizations_test.py:15-30),
    backward applies the same mask, successive calls draw independent masks, a seed reproduces a run."""
    L = npm.layers
    keep = 1 - p
    rng = np.random.default_rng(0)
    x = rng.standard_normal(shape).astype(np.float32)
    try:
        npm.set_dropout_rng('device', seed=0x1234567811223344)
        d = L.DropOut(p)
        y = np.asarray(d(x))
        want_mask = O.dropout_philox_mask(x.size, keep, 0x1234567811223344, 0).reshape(shape)
        np.testing.assert_array_equal(d._mask != 0, want_mask)
        np.testing.assert_array_equal(y, np.where(want_mask, x / np.float32(keep), 0).astype(np.float32))      # normalizations.py:23: x / keep_prob
        dy = rng.standard_normal(shape).astype(np.float32)
        np.testing.assert_array_equal(np.asarray(d.backward(dy)),
                                      np.where(want_mask, dy / np.float32(keep), 0).astype(np.float32))
        y2 = np.asarray(d(x))                                    # next call: next offset
        np.testing.assert_array_equal(d._mask != 0, O.dropout_philox_mask(x.size, keep, 0x1234567811223344, 1).reshape(shape))
        npm.set_dropout_rng('device', seed=0x1234567811223344)   # same seed: same run
        np.testing.assert_array_equal(np.asarray(L.DropOut(p)(x)), y)
        if x.size > 1000:
            assert abs((d._mask != 0).mean() - keep) < 0.01 and not np.array_equal(y, y2)
    finally:
        npm.set_dropout_rng('host')
    np.random.seed(3)                                            # back on the host generator
    h = L.DropOut(p)
    h(x)
    np.random.seed(3)
    np.testing.assert_array_equal(h._mask, np.random.binomial(n=1, p=keep, size=x.size).reshape(shape))


def test_encoder_with_device_dropout(npm):
    """drop_rate > 0 inside the encoder (fused composition: the dropouts ride inside the LayerNorm kernels) with
    device-drawn masks: the step equals the oracle's encoder evaluated with those very masks."""
    rng = np.random.default_rng(2)
    x = rng.standard_normal([2, 12, 32]).astype(np.float32)
    dy = rng.standard_normal([2, 12, 32]).astype(np.float32)
    try:
        npm.set_dropout_rng('device', seed=99)
        np.random.seed(0)
        enc = npm.layers.TransformerEncoder(num_heads=2, hidden_units=48, norm_first=True, drop_rate=0.2)
        out = np.asarray(enc(x))
        dx = np.asarray(enc(dy, backprop=True, learning_rate=0.0))
        m1, m2 = enc._dropout1._mask != 0, enc._dropout2._mask != 0
        assert m1.shape == (2, 12, 32) and m2.shape == (24, 32) and 0.6 < m1.mean() < 0.95
    finally:
        npm.set_dropout_rng('host')
    # the same computation with dropout off and the masks applied by hand around the norm inputs (pre-norm: x -> drop -> norm)
    p = {}
    att = enc._self_attention
    for n in O.MHA_PARAM_NAMES:
        p['att_' + n] = np.asarray(getattr(att, '_' + n)).astype(np.float64)
    for tag, norm in (('n1', enc._norm1), ('n2', enc._norm2)):
        p[tag + '_gamma'], p[tag + '_beta'] = (np.asarray(getattr(norm, a)).astype(np.float64) for a in ('_gamma', '_beta'))
    p['d1_w'], p['d1_b'] = (np.asarray(getattr(enc._dense1.linear, a)).astype(np.float64) for a in ('_w', '_b'))
    p['d2_w'], p['d2_b'] = (np.asarray(getattr(enc._dense2, a)).astype(np.float64) for a in ('_w', '_b'))
    want, cache = O.encoder_fwd(p, x.astype(np.float64), True, drop=(m1, m2, 0.8))
    assert_close(out, want, tol=1e-5)
    want_dx, _ = O.encoder_bwd(p, cache, dy.astype(np.float64), True)
    assert_close(dx, want_dx, tol=1e-5)
    assert enc._fused


# ---- MultiHeadAttention ------------------------------------------------------------------------
_MHA = ['wq', 'wk', 'wv', 'wo', 'bq', 'bk', 'bv', 'bo']


@pytest.mark.parametrize('name', ['mha_self', 'mha_cross'])
def test_mha_golden(npm, name, math_mode):
    g = load_golden(name)
    layer = npm.layers.MultiHeadAttention(num_heads=int(g['heads']))
    kv = g.get('kv')
    out = layer(g['query'], kv) if kv is not None else layer(g['query'])
    for n in _MHA:                                             # bind the reference's parameters
        assert getattr(layer, '_' + n).shape == g[n + '0'].shape
        setattr(layer, '_' + n, g[n + '0'])
    out = layer(g['query'], kv) if kv is not None else layer(g['query'])
    assert_close(out, g['out'], tol=1e-5)
    layer2 = copy.deepcopy(layer)                              # attentions_test.py:72
    dq, dk, dv = layer2(g['dy'], backprop=True, learning_rate=float(g['lr']))
    assert_close(dq, g['dquery'], tol=1e-5)
    assert_close(dk, g['dkey'], tol=1e-5)
    assert_close(dv, g['dvalue'], tol=1e-5)
    for n in _MHA:
        assert_close(getattr(layer2, '_' + n), g[n + '1'], tol=1e-5, what=n)
        np.testing.assert_array_equal(np.asarray(getattr(layer, '_' + n)), g[n + '0'])   # deepcopy isolated


def test_mha_reference_test_shape_vs_oracle(npm, math_mode):
    """B16, Sq32, F128, H8 -- reference layers/attentions_test.py:13-85 (self-attention)."""
    np.random.seed(0)
    layer = npm.layers.MultiHeadAttention(num_heads=8)
    query = rand([16, 32, 128])
    out = layer(query)
    p = {n: np.asarray(getattr(layer, '_' + n)).astype(np.float64) / (4.0 if n[0] == 'w' else 1.0) for n in _MHA}
    for n in _MHA:
        setattr(layer, '_' + n, p[n].astype(np.float32))
        p[n] = p[n].astype(np.float32).astype(np.float64)
    out = layer(query)
    want, cache = O.mha_fwd(p, query.astype(np.float64))
    assert_close(out, want, tol=1e-5)
    dy = rand([16, 32, 128]) * 0.01
    dq, dk, dv = layer(dy, backprop=True, learning_rate=0.01)
    (wq_, wk_, wv_), grads = O.mha_bwd(p, cache, dy.astype(np.float64))
    assert_close(np.asarray(dq) + np.asarray(dk) + np.asarray(dv), wq_ + wk_ + wv_, tol=1e-5)
    for n in _MHA:
        assert_close(getattr(layer, '_' + n), p[n] - 0.01 * grads[n], tol=1e-5, what=n)


@pytest.mark.parametrize('rowdot', [True, False])
@pytest.mark.parametrize('cross', [False, True])
def test_mha_head_size_128_row_terms_from_the_dctx_gemm(npm, monkeypatch, rowdot, cross):
    """Head size 128 (the C4 / C5 head): the attention backward's row terms come out of the epilogue of the dctx = dy wo GEMM
    (NPM_EPI_ROWDOT) by default, or from the pass inside npm_mha_core_bwd (NPM_ATTN_ROWDOT=0); both against the oracle, self-
    and cross-attention (Sq != Skv, ragged against the 32-query tiles)."""
    from np_modeling_amd import _C, device as D
    monkeypatch.setattr(D, 'ATTN_ROWDOT', rowdot)
    np.random.seed(1)
    layer = npm.layers.MultiHeadAttention(num_heads=2)
    query = rand([3, 72, 256])
    kv = rand([3, 200, 256]) if cross else None
    args = (query, kv) if cross else (query,)
    layer(*args)
    p = {n: np.asarray(getattr(layer, '_' + n)).astype(np.float64) / (16.0 if n[0] == 'w' else 1.0) for n in _MHA}
    for n in _MHA:
        getattr(layer, '_' + n).set(p[n].astype(np.float32))
        p[n] = p[n].astype(np.float32).astype(np.float64)
    out = layer(*args)
    want, cache = O.mha_fwd(p, query.astype(np.float64), None if kv is None else kv.astype(np.float64))
    assert_close(out, want, tol=1e-5)
    dy = rand([3, 72, 256]) * 0.01
    calls = []
    inner = D.gemm
    monkeypatch.setattr(D, 'gemm', lambda *a, **k: (calls.append(k.get('rowdot') is not None), inner(*a, **k))[1])
    dq, dk, dv = layer(dy, backprop=True, learning_rate=0.01)
    assert any(calls) == rowdot and _C.last_attn_kernel().startswith('mha_bwd16_kernel' if not cross else 'mha_bwd')
    (wq_, wk_, wv_), grads = O.mha_bwd(p, cache, dy.astype(np.float64))
    assert_close(dq, wq_, tol=1e-5)
    assert_close(np.asarray(dk) + np.asarray(dv), wk_ + wv_, tol=1e-5)
    for n in _MHA:
        assert_close(getattr(layer, '_' + n), p[n] - 0.01 * grads[n], tol=1e-5, what=n)


def test_mha_deepcopy_after_packed_forward(npm):
    """attentions_test.py:72 on the PACKED self-attention path (default-initialised parameters, one q/k/v GEMM):
    the cached k and v are views into the packed buffer; a deep copy made between forward and backward must give
    the original layer's backward bit for bit, and the copied encoder likewise."""
    np.random.seed(5)
    layer = npm.layers.MultiHeadAttention(num_heads=4)
    x, dy = rand([3, 20, 64]), rand([3, 20, 64])
    layer(x)
    assert layer._packed
    twin = copy.deepcopy(layer)
    assert twin._params_adjacent()
    got = [np.asarray(g) for g in twin(dy, backprop=True, learning_rate=0.1)]
    want = [np.asarray(g) for g in layer(dy, backprop=True, learning_rate=0.1)]
    for a, b in zip(got, want):
        assert np.isfinite(a).all()
        np.testing.assert_array_equal(a, b)
    for n in _MHA:
        np.testing.assert_array_equal(np.asarray(getattr(twin, '_' + n)), np.asarray(getattr(layer, '_' + n)))
    enc = npm.layers.TransformerEncoder(num_heads=4, hidden_units=96, norm_first=True)
    enc(x)
    twin = copy.deepcopy(enc)
    np.testing.assert_array_equal(np.asarray(twin(dy, backprop=True, learning_rate=0.1)),
                                  np.asarray(enc(dy, backprop=True, learning_rate=0.1)))


def test_mha_mask_conventions(npm):
    layer = npm.layers.MultiHeadAttention(num_heads=2)
    q = rand([2, 4, 8])
    layer(q)
    with pytest.raises(NotImplementedError):             # head size 4: no fused kernel, and masks exist only there
        layer(q, mask=np.ones([2, 2, 4, 4], dtype=bool))
    with pytest.raises(AssertionError):
        npm.layers.MultiHeadAttention(num_heads=3)(rand([2, 4, 8]))     # 8 % 3 != 0
