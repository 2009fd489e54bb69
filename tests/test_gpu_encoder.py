"""GPU: TransformerEncoder / TransformerDecoder against the reference's outputs and the oracle,
plus the Trainer flows of reference train_test.py."""

import numpy as np
import pytest

from conftest import assert_close, load_golden
from oracle import np_oracle as O

pytestmark = pytest.mark.gpu


class GradRecorder:
    """An optimizer that records gradients instead of applying them (parameters stay fixed)."""

    def __init__(self):
        self.grads = {}

    def update(self, obj, attribute, gradient):
        self.grads[(type(obj).__name__, attribute, id(obj))] = gradient

_MAP = dict(att_wq=('_self_attention', '_wq'), att_wk=('_self_attention', '_wk'),
            att_wv=('_self_attention', '_wv'), att_wo=('_self_attention', '_wo'),
            att_bq=('_self_attention', '_bq'), att_bk=('_self_attention', '_bk'),
            att_bv=('_self_attention', '_bv'), att_bo=('_self_attention', '_bo'),
            n1_gamma=('_norm1', '_gamma'), n1_beta=('_norm1', '_beta'),
            n2_gamma=('_norm2', '_gamma'), n2_beta=('_norm2', '_beta'),
            d1_w=('_dense1._linear', '_w'), d1_b=('_dense1._linear', '_b'),
            d2_w=('_dense2', '_w'), d2_b=('_dense2', '_b'))


@pytest.fixture(scope='module')
def npm():
    import np_modeling_amd
    return np_modeling_amd


def rand(shape):
    return np.random.normal(size=shape).astype(np.float32)


def _sub(enc, path):
    obj = enc
    for part in path.split('.'):
        obj = getattr(obj, part)
    return obj


def _params(enc):
    return {k: np.asarray(getattr(_sub(enc, path), attr)) for k, (path, attr) in _MAP.items()}


@pytest.mark.parametrize('name', ['encoder_prenorm', 'encoder_postnorm'])
def test_encoder_golden(npm, name, math_mode):
    """Seeded like oracle/make_golden.py: the lazily drawn parameters must equal the
    reference's bit for bit, then forward / dx / every updated parameter must match."""
    g = load_golden(name)
    nf = bool(g['norm_first'])
    np.random.seed(0)
    enc = npm.layers.TransformerEncoder(num_heads=int(g['heads']), hidden_units=int(g['hidden']), norm_first=nf)
    qkv = rand(g['qkv'].shape)
    out = enc(qkv)
    for k, v in _params(enc).items():
        np.testing.assert_array_equal(v, g[k + '__0'], err_msg=k)
    assert_close(out, g['out'], tol=1e-5)
    dx = enc(g['dy'], backprop=True, learning_rate=float(g['lr']))
    assert_close(dx, g['dx'], tol=1e-5)
    for k, v in _params(enc).items():
        assert_close(v, g[k + '__1'], tol=1e-5, what=k)


@pytest.mark.parametrize('name', ['encoder_dropout_prenorm', 'encoder_dropout_postnorm'])
def test_encoder_with_dropout_golden(npm, name, math_mode):
    """The reference's seeded run with drop_rate = 0.1 (transformer.py:13,22-23,35-36,40-41,49-50,55-56; masks from
    np.random.binomial, normalizations.py:14-30): same seed -> the same parameters and the same masks, bit for bit; output,
    dx and all 16 updated parameters match -- through the FUSED composition: each DropOut applied inside the LayerNorm
    kernels behind it (npm_layernorm_dropout_fwd / _bwd), no mask pass, no standalone add."""
    g = load_golden(name)
    nf = bool(g['norm_first'])
    np.random.seed(0)
    enc = npm.layers.TransformerEncoder(num_heads=int(g['heads']), hidden_units=int(g['hidden']), norm_first=nf,
                                        drop_rate=float(g['drop_rate']))
    qkv = rand(g['qkv'].shape)
    out = enc(qkv)
    assert enc._fused
    for k, v in _params(enc).items():
        np.testing.assert_array_equal(v, g[k + '__0'], err_msg=k)
    np.testing.assert_array_equal(np.asarray(enc._dropout1._mask), g['mask1'])
    np.testing.assert_array_equal(np.asarray(enc._dropout2._mask), g['mask2'])
    assert_close(out, g['out'], tol=1e-5)
    dx = enc(g['dy'], backprop=True, learning_rate=float(g['lr']))
    assert_close(dx, g['dx'], tol=1e-5)
    for k, v in _params(enc).items():
        assert_close(v, g[k + '__1'], tol=1e-5, what=k)


@pytest.mark.parametrize('rows,d,residual', [(64, 1024, True), (37, 128, False), (5, 4096, True), (130, 72, True), (3, 2048, False)])
def test_layernorm_dropout_kernels_equal_the_composed_calls(npm, rows, d, residual):
    """npm_layernorm_dropout_fwd / _bwd (include/npm_hip.h) equal npm_mask_scale -> npm_layernorm_fwd and
    npm_layernorm_bwd -> npm_mask_scale (-> npm_add) to rounding: z, mean, rstd, dx, dgamma, dbeta.  Row lengths the row-in-registers
    kernels do not take are refused (the layer composes the calls then)."""
    from np_modeling_amd import _C
    D = npm.device
    rng = np.random.default_rng(rows + d)
    x = D.from_host(rng.standard_normal([rows, d]).astype(np.float32))
    dz = D.from_host(rng.standard_normal([rows, d]).astype(np.float32))
    res = D.from_host(rng.standard_normal([rows, d]).astype(np.float32)) if residual else None
    gamma, beta = D.from_host(rng.standard_normal([d]).astype(np.float32)), D.from_host(rng.standard_normal([d]).astype(np.float32))
    keep = 0.75
    mask_host = (rng.random([rows, d]) < keep).astype(np.uint8)
    mask = D.bytes_from_host(mask_host)
    # composed
    xd = D.empty([rows, d])
    _C.check(_C.lib().npm_mask_scale(x.ptr, mask.ptr, xd.ptr, x.size, keep))
    z0, mean0, rstd0 = D.layernorm_fwd(xd, gamma, beta, 1e-3)
    dg0, db0 = D.empty([d]), D.empty([d])
    inner = D.layernorm_bwd(dz, xd, mean0, rstd0, gamma, dg0, db0)
    dx0 = D.empty([rows, d])
    _C.check(_C.lib().npm_mask_scale(inner.ptr, mask.ptr, dx0.ptr, x.size, keep))
    if residual:
        dx0 = D.add(dx0, res)
    # fused
    z1, mean1, rstd1 = D.layernorm_fwd(x, gamma, beta, 1e-3, drop=(mask, keep))
    dg1, db1 = D.empty([d]), D.empty([d])
    dx1 = D.layernorm_bwd(dz, x, mean1, rstd1, gamma, dg1, db1, residual=res, drop=(mask, keep))
    # (not bitwise: the row sums  sum g yhat  / sum (x - mean)^2  are sums of products, which hipcc contracts into fused multiply-adds
    #  its own way in every template instance -- measured: 3 % of dx one ulp apart at [64, 1024])
    for name, a, b in (('z', z1, z0), ('mean', mean1, mean0), ('rstd', rstd1, rstd0), ('dx', dx1, dx0), ('dgamma', dg1, dg0), ('dbeta', db1, db0)):
        assert_close(a.numpy(), b.numpy(), tol=5e-7, what=name)
    np.testing.assert_array_equal(dx1.numpy()[mask_host == 0], res.numpy()[mask_host == 0] if residual else 0)    # dropped positions: exactly the residual
    # against the definition
    want = np.where(mask_host != 0, x.numpy() / np.float32(keep), np.float32(0))
    wz, _ = O.layernorm_fwd(want.astype(np.float64), gamma.numpy().astype(np.float64), beta.numpy().astype(np.float64), 1e-3)
    assert_close(z1, wz, tol=3e-6)
    assert not D.layernorm_dropout_supported(70) and not D.layernorm_dropout_supported(8192)
    with pytest.raises(_C.NpmError):
        D.layernorm_fwd(D.zeros([4, 70]), D.zeros([70]), D.zeros([70]), 1e-3, drop=(D.bytes_from_host(np.ones([4, 70], dtype=np.uint8)), keep))


@pytest.mark.parametrize('norm_first', [True, False])
def test_encoder_reference_test_shape(npm, norm_first, math_mode):
    """B16, S32, F128, H8, U256 (reference layers/transformer_test.py:98-156), weights scaled
    down like a trained model so activations stay O(1)."""
    np.random.seed(0)
    enc = npm.layers.TransformerEncoder(num_heads=8, hidden_units=256, norm_first=norm_first)
    qkv = rand([16, 32, 128])
    enc(qkv)
    p = {}
    for k, (path, attr) in _MAP.items():
        arr = np.asarray(getattr(_sub(enc, path), attr))
        if arr.ndim > 1:
            arr = (arr / np.sqrt(arr.shape[-1] if k.startswith('att_w') else arr.shape[0])).astype(np.float32)
        setattr(_sub(enc, path), attr, arr)
        p[k] = arr.astype(np.float64)
    out = enc(qkv)
    want, cache = O.encoder_fwd(p, qkv.astype(np.float64), norm_first)
    assert_close(out, want, tol=1e-5)
    dy = rand([16, 32, 128]) * 0.1
    dx = enc(dy, backprop=True, learning_rate=1e-3)
    want_dx, grads = O.encoder_bwd(p, cache, dy.astype(np.float64), norm_first)
    assert_close(dx, want_dx, tol=1e-5)
    for k, v in _params(enc).items():
        assert_close(v, p[k] - 1e-3 * grads[k], tol=1e-5, what=k)


@pytest.mark.parametrize('norm_first', [True, False])
def test_encoder_fused_equals_unfused(npm, norm_first):
    """The epilogue-fused composition and the literal reference order agree to rounding."""
    np.random.seed(1)
    enc = npm.layers.TransformerEncoder(num_heads=4, hidden_units=64, norm_first=norm_first)
    qkv, dy = rand([5, 12, 32]), rand([5, 12, 32])
    out = np.asarray(enc(qkv))
    unf = np.asarray(enc._forward_unfused(npm.as_device(qkv)))
    assert_close(out, unf, tol=2e-6)

    class Recorder:
        def __init__(self):
            self.grads = {}

        def update(self, obj, attribute, gradient):
            self.grads[(id(obj), attribute)] = np.asarray(gradient).copy()

    from np_modeling_amd import parallel
    r1, r2 = Recorder(), Recorder()
    enc(qkv)
    with parallel.grad_scope(0) as scope:
        dx1 = np.asarray(enc._backward_fused(npm.as_device(dy), r1, scope))
    enc(qkv)
    with parallel.grad_scope(0) as scope:
        dx2 = np.asarray(enc._backward_unfused(npm.as_device(dy), r2, scope))
    assert_close(dx1, dx2, tol=5e-6)
    assert r1.grads.keys() == r2.grads.keys() and len(r1.grads) == 16
    bq_scale = max(np.abs(v).max() for k, v in r1.grads.items() if k[1] == '_bq')
    for key in r1.grads:
        if key[1] == '_bk':
            # zero in real arithmetic (every row of datt sums to 0): what either composition stores is rounding noise
            assert np.abs(r1.grads[key]).max() < 1e-4 * bq_scale and np.abs(r2.grads[key]).max() < 1e-4 * bq_scale
            continue
        assert_close(r1.grads[key], r2.grads[key], tol=5e-6, what=str(key))


_DEC = dict(n1_gamma=('_norm1', '_gamma'), n1_beta=('_norm1', '_beta'), n2_gamma=('_norm2', '_gamma'),
            n2_beta=('_norm2', '_beta'), n3_gamma=('_norm3', '_gamma'), n3_beta=('_norm3', '_beta'),
            d1_w=('_dense1._linear', '_w'), d1_b=('_dense1._linear', '_b'), d2_w=('_dense2', '_w'), d2_b=('_dense2', '_b'))
for _n in ('wq', 'wk', 'wv', 'wo', 'bq', 'bk', 'bv', 'bo'):
    _DEC['sa_' + _n] = ('_self_attention', '_' + _n)
    _DEC['ca_' + _n] = ('_cross_attention', '_' + _n)


@pytest.mark.parametrize('name', ['decoder_prenorm', 'decoder_postnorm'])
def test_decoder_golden(npm, name, math_mode):
    """TransformerDecoder (reference transformer.py:95-203, test transformer_test.py:159-219): seeded
    parameters equal the reference's, then output, (dq, dkv) and all 26 updated parameters match."""
    g = load_golden(name)
    np.random.seed(0)
    dec = npm.layers.TransformerDecoder(num_heads=int(g['heads']), hidden_units=int(g['hidden']),
                                        norm_first=bool(g['norm_first']))
    q, kv = rand(g['q'].shape), rand(g['kv'].shape)
    out = dec(q, kv)
    for k, (path, attr) in _DEC.items():
        np.testing.assert_array_equal(np.asarray(getattr(_sub(dec, path), attr)), g[k + '__0'], err_msg=k)
    # tol 1e-4 (BASELINE.json's bound): two near-one-hot softmaxes amplify fp32 summation-order noise,
    # the reference's own einsum order is no closer to exact (see tests/test_oracle_golden.py::test_decoder)
    assert_close(out, g['out'], tol=1e-4)
    dq, dkv = dec(g['dy'], backprop=True, learning_rate=float(g['lr']))
    assert_close(dq, g['dq'], tol=1e-4)
    assert_close(dkv, g['dkv'], tol=1e-4)
    for k, (path, attr) in _DEC.items():
        assert_close(getattr(_sub(dec, path), attr), g[k + '__1'], tol=1e-4, what=k)


@pytest.mark.parametrize('name', ['decoder_dropout_prenorm', 'decoder_dropout_postnorm'])
def test_decoder_with_dropout_golden(npm, name, math_mode):
    """The reference's seeded decoder run with drop_rate = 0.1 (transformer.py:98,111-113; normalizations.py:14-30): same seed
    -> the same 26 parameters and the same three masks bit for bit, then output, (dq, dkv) and all updated parameters --
    through the fused composition (the dropouts inside the LayerNorm kernels)."""
    g = load_golden(name)
    np.random.seed(0)
    dec = npm.layers.TransformerDecoder(num_heads=int(g['heads']), hidden_units=int(g['hidden']),
                                        norm_first=bool(g['norm_first']), drop_rate=float(g['drop_rate']))
    q, kv = rand(g['q'].shape), rand(g['kv'].shape)
    out = dec(q, kv)
    assert dec._fused
    for k, (path, attr) in _DEC.items():
        np.testing.assert_array_equal(np.asarray(getattr(_sub(dec, path), attr)), g[k + '__0'], err_msg=k)
    for i, d in enumerate((dec._dropout1, dec._dropout2, dec._dropout3), start=1):
        np.testing.assert_array_equal(np.asarray(d._mask), g[f'mask{i}'])
    assert_close(out, g['out'], tol=1e-4)
    dq, dkv = dec(g['dy'], backprop=True, learning_rate=float(g['lr']))
    assert_close(dq, g['dq'], tol=1e-4)
    assert_close(dkv, g['dkv'], tol=1e-4)
    for k, (path, attr) in _DEC.items():
        assert_close(getattr(_sub(dec, path), attr), g[k + '__1'], tol=1e-4, what=k)


@pytest.mark.parametrize('norm_first', [True, False])
def test_decoder_fused_equals_unfused(npm, norm_first):
    """The fused decoder composition (residuals and the dkey + dvalue / dq + dk + dv sums in GEMM epilogues and the
    LayerNorm backward, reference transformer.py:120-203) against the literal composition from standalone kernels,
    at O(1) activations (the golden fixture's unscaled weights saturate the softmaxes)."""
    from np_modeling_amd import parallel
    D = npm.device
    rng = np.random.default_rng(4)
    np.random.seed(7)
    dec = npm.layers.TransformerDecoder(num_heads=4, hidden_units=96, norm_first=norm_first)
    q = rng.standard_normal([3, 20, 64]).astype(np.float32)
    kv = rng.standard_normal([3, 37, 64]).astype(np.float32)
    dy = rng.standard_normal([3, 20, 64]).astype(np.float32)
    dec(q, kv)
    for path, attrs in (('_self_attention', ('_wq', '_wk', '_wv', '_wo')), ('_cross_attention', ('_wq', '_wk', '_wv', '_wo')),
                        ('_dense1._linear', ('_w',)), ('_dense2', ('_w',))):
        for attr in attrs:
            arr = getattr(_sub(dec, path), attr)
            arr.set(np.asarray(arr) / np.float32(8.0))
    out = np.asarray(dec(q, kv))
    r1 = GradRecorder()
    dq1, dkv1 = (np.asarray(g) for g in dec(dy, backprop=True, optimizer_=r1))
    ref = np.asarray(dec._forward_unfused(D.as_device(q), D.as_device(kv)))
    assert_close(out, ref, tol=3e-6)
    r2 = GradRecorder()
    with parallel.grad_scope(0) as scope:
        dq2, dkv2 = (np.asarray(g) for g in dec._backward_unfused(D.as_device(dy), r2, scope))
    assert_close(dq1, dq2, tol=1e-5)
    assert_close(dkv1, dkv2, tol=1e-5)
    assert len(r1.grads) == 26 and r1.grads.keys() == r2.grads.keys()
    for key in r1.grads:
        a, b = np.asarray(r1.grads[key]), np.asarray(r2.grads[key])
        if key[1] == '_bk':      # exactly zero in real arithmetic (rows of datt sum to 0): rounding noise in both
            assert np.abs(a).max() < 1e-4 and np.abs(b).max() < 1e-4
            continue
        assert_close(a, b, tol=1e-5, what=str(key[:2]))


def test_losses_on_device(npm):
    """reference loss_test.py:15-66: MSE and cross-entropy forward / backward on device outputs."""
    g = load_golden('losses')
    mse = npm.loss.MSELoss()
    value = mse(npm.as_device(g['y']), g['t'])
    np.testing.assert_allclose(value, g['mse'], rtol=1e-6)
    grad = mse(backprop=True)
    assert isinstance(grad, npm.DeviceArray)
    assert_close(grad, g['mse_grad'], tol=1e-6)
    np.testing.assert_allclose(npm.loss.MSELoss()(g['y'], g['t']), g['mse'], rtol=1e-6)      # host inputs
    ce = npm.loss.CrossEntropyLoss()
    np.testing.assert_allclose(ce(npm.as_device(g['prob']), g['onehot']), g['ce'], rtol=1e-6)
    assert_close(ce(backprop=True), g['ce_grad'], tol=1e-6)


def test_softmax_cross_entropy_chain_on_device(npm):
    """reference loss_test.py:49-66, the composed flow in its own call form: ``ce(softmax(y), t)`` and
    ``softmax(ce(y, t, backprop=True), backprop=True)`` -- the loss's backward ignores the arguments it is called
    with and uses what its forward cached; everything stays on the device between the two layers."""
    g = load_golden('softmax_ce')
    ce, softmax = npm.loss.CrossEntropyLoss(), npm.layers.Softmax()
    y = npm.as_device(g['y'])
    prob = softmax(y)
    assert isinstance(prob, npm.DeviceArray)
    assert_close(prob, g['prob'], tol=1e-6)
    np.testing.assert_allclose(ce(prob, g['targets']), g['ce'], rtol=1e-6)
    dprob = ce(y, g['targets'], backprop=True)
    assert isinstance(dprob, npm.DeviceArray)
    assert_close(dprob, g['dprob'], tol=1e-6)
    dy = softmax(dprob, backprop=True)
    assert_close(dy, g['dy'], tol=2e-6)
    prob64 = g['prob'].astype(np.float64)
    assert_close(dy, prob64 * g['targets'].astype(np.float64).sum(axis=-1, keepdims=True) - g['targets'], tol=2e-6)


def test_adam_on_device_matches_reference_numerics(npm):
    """Three Adam steps on device-resident fp64 moments against the oracle's restatement of
    reference optimizer.py:53-67 (epsilon inside the sqrt, bias correction)."""
    rng = np.random.default_rng(0)
    w0 = rng.standard_normal((37, 29)).astype(np.float32)
    var = npm.as_device(w0)
    adam = npm.optimizer.AdamOptimizer(1e-2)
    holder = type('Holder', (), {})()
    holder._w = var
    state, want = {}, w0.copy()
    for step in range(3):
        g = rng.standard_normal((37, 29)).astype(np.float32)
        adam.update(holder, '_w', npm.as_device(g))
        want = O.adam_step(want, g, state, 1e-2)
        assert_close(holder._w, want, tol=1e-6)
    assert holder._w is var


@pytest.mark.parametrize('opt', ['sgd', 'adam'])
def test_trainer_mlp_trajectory(npm, opt, capsys):
    """reference train_test.py:14-49: 10 training steps + eval; the printed losses are the
    reference's own (tests/golden/train_mlp_*.npz)."""
    import re
    g = load_golden('train_mlp_' + opt)
    np.random.seed(0)
    feats = [16, 32, 64, 32, 16]
    stack = [npm.layers.Dense(units=f, name=f'layer_{i}') for i, f in enumerate(feats)]
    x = np.random.uniform(0.0, 1.0, size=[128, 16]).astype(np.float32)
    t = np.random.uniform(0.0, 1.0, size=[128, 16]).astype(np.float32)
    optim = npm.optimizer.AdamOptimizer(1e-4) if opt == 'adam' else npm.optimizer.SGDOptimizer(1e-4)
    trainer = npm.train.Trainer(stack)
    trainer.train(inputs=x, targets=t, steps=10, optimizer_=optim)
    trainer.eval(inputs=x, targets=t)
    losses = [float(v) for v in re.findall(r'Loss:\s+([0-9.eE+-]+)', capsys.readouterr().out)]
    np.testing.assert_allclose(losses, g['losses'], rtol=5e-5)
    for i, layer in enumerate(stack):
        assert_close(layer.linear.w, g[f'w{i}'], tol=1e-5)
