"""CPU: the oracle (both flavours) against fixtures produced by the real reference.

Fixtures come from oracle/make_golden.py (imports /root/reference in the build
container).  Tolerances: the oracle repeats the reference's NumPy expressions, so
results agree to rounding; closed forms differ from the Jacobian forms by a few ulp
of fp32-scaled values (SURVEY.md 8c measured 1.3e-8 / 3.5e-7)."""

import numpy as np
import pytest

from oracle import np_oracle as O
from conftest import load_golden

TIGHT = dict(rtol=1e-6, atol=1e-6)


def close(a, b, **kw):
    tol = dict(TIGHT)
    tol.update(kw)
    np.testing.assert_allclose(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64), **tol)


def close_scaled(a, b, tol=1e-5):
    """|a-b| <= tol * (|b| + max|b|): the GEMM-reordering flavour differs from the
    reference's einsum order by fp32 rounding of O(max|b|) * eps per element."""
    b = np.asarray(b, dtype=np.float64)
    np.testing.assert_allclose(np.asarray(a, dtype=np.float64), b, rtol=tol, atol=tol * np.abs(b).max())


@pytest.mark.parametrize('name,relu', [('dense', True), ('linear', False)])
def test_dense(name, relu):
    g = load_golden(name)
    if relu:
        y, pre = O.dense_fwd(g['x'], g['w0'], g['b0'])
        dx, dw, db = O.dense_bwd(g['x'], g['w0'], pre, g['dy'])
    else:
        y = O.linear_fwd(g['x'], g['w0'], g['b0'])
        dx, dw, db = O.linear_bwd(g['x'], g['w0'], g['dy'])
    close(y, g['y'])
    close(dx, g['dx'])
    close(O.sgd_step(g['w0'], dw, float(g['lr'])), g['w1'])
    close(O.sgd_step(g['b0'], db, float(g['lr'])), g['b1'])


def test_relu_passes_gradient_at_zero():
    g = load_golden('relu')
    close(O.relu_fwd(g['x']), g['y'])
    dx = O.relu_bwd(g['x'], g['dy'])
    close(dx, g['dx'])
    assert np.all(dx[0, :5] == g['dy'][0, :5])      # x == 0 -> gradient passes (>=)


@pytest.mark.parametrize('verbatim', [True, False])
def test_softmax(verbatim):
    g = load_golden('softmax')
    y = O.softmax_fwd(g['x'])
    close(y, g['y'])
    close(O.softmax_bwd(y, g['dy'], verbatim=verbatim), g['dx'], atol=1e-7)


@pytest.mark.parametrize('name', ['layernorm_2d', 'layernorm_3d'])
@pytest.mark.parametrize('verbatim', [True, False])
def test_layernorm(name, verbatim):
    g = load_golden(name)
    eps = float(g['eps'])
    z, cache = O.layernorm_fwd(g['x'], g['gamma0'], g['beta0'], eps)
    close(z, g['z'])
    dx, dgamma, dbeta = O.layernorm_bwd(g['x'], g['gamma0'], eps, cache, g['dz'], verbatim=verbatim)
    close(dx, g['dx'], atol=5e-6)
    close(O.sgd_step(g['gamma0'], dgamma, float(g['lr'])), g['gamma1'])
    close(O.sgd_step(g['beta0'], dbeta, float(g['lr'])), g['beta1'])


@pytest.mark.parametrize('name', ['conv_k1', 'conv_k3', 'conv_k5'])
def test_conv(name):
    g = load_golden(name)
    y, pre = O.conv_layer_fwd(g['x'], g['w0'], g['b0'])
    close(y, g['y'])
    dx, dw, db = O.conv_layer_bwd(g['x'], g['w0'], pre, g['dy'])
    close(dx, g['dx'])
    close(O.sgd_step(g['w0'], dw, float(g['lr'])), g['w1'])
    close(O.sgd_step(g['b0'], db, float(g['lr'])), g['b1'])


@pytest.mark.parametrize('name', ['mha_self', 'mha_cross'])
@pytest.mark.parametrize('verbatim', [True, False])
def test_mha(name, verbatim):
    g = load_golden(name)
    p = {n: g[n + '0'] for n in O.MHA_PARAM_NAMES}
    kv = g.get('kv')
    out, cache = O.mha_fwd(p, g['query'], kv, verbatim=verbatim)
    close_scaled(out, g['out'])
    (dq, dk, dv), grads = O.mha_bwd(p, cache, g['dy'], verbatim=verbatim)
    close_scaled(dq, g['dquery'])
    close_scaled(dk, g['dkey'])
    close_scaled(dv, g['dvalue'])
    for n in O.MHA_PARAM_NAMES:
        close_scaled(O.sgd_step(p[n], grads[n], float(g['lr'])), g[n + '1'])


@pytest.mark.parametrize('name', ['encoder_prenorm', 'encoder_postnorm'])
@pytest.mark.parametrize('verbatim', [True, False])
def test_encoder(name, verbatim):
    g = load_golden(name)
    nf = bool(g['norm_first'])
    p = {k[:-3]: v for k, v in g.items() if k.endswith('__0')}
    out, cache = O.encoder_fwd(p, g['qkv'], nf, verbatim=verbatim)
    close_scaled(out, g['out'])
    dx, grads = O.encoder_bwd(p, cache, g['dy'], nf, verbatim=verbatim)
    close_scaled(dx, g['dx'])
    for k, grad in grads.items():
        close_scaled(O.sgd_step(p[k], grad, float(g['lr'])), g[k + '__1'])


@pytest.mark.parametrize('name', ['encoder_dropout_prenorm', 'encoder_dropout_postnorm'])
@pytest.mark.parametrize('verbatim', [True, False])
def test_encoder_with_dropout(name, verbatim):
    """drop_rate = 0.1 (reference transformer.py:13,22-23 and normalizations.py:14-30): with the masks the reference drew, the
    oracle reproduces its output, input gradient and every updated parameter."""
    g = load_golden(name)
    nf = bool(g['norm_first'])
    p = {k[:-3]: v for k, v in g.items() if k.endswith('__0')}
    drop = (g['mask1'], g['mask2'], 1.0 - float(g['drop_rate']))
    assert 0.8 < g['mask1'].mean() < 0.97 and g['mask2'].shape == (g['qkv'].shape[0] * g['qkv'].shape[1], g['qkv'].shape[2])
    out, cache = O.encoder_fwd(p, g['qkv'], nf, verbatim=verbatim, drop=drop)
    close_scaled(out, g['out'])
    dx, grads = O.encoder_bwd(p, cache, g['dy'], nf, verbatim=verbatim)
    close_scaled(dx, g['dx'])
    for k, grad in grads.items():
        close_scaled(O.sgd_step(p[k], grad, float(g['lr'])), g[k + '__1'])
    # and the masks are the global generator's draws at their place in the lazy-initialisation order
    np.random.seed(0)
    qkv = np.random.normal(size=g['qkv'].shape).astype(np.float32)
    np.testing.assert_array_equal(qkv, g['qkv'])
    keep = 1.0 - float(g['drop_rate'])
    if nf:                                   # pre-norm: dropout1 runs first of all (transformer.py:35), then norm1 draws gamma, beta
        m1 = np.random.binomial(n=1, p=keep, size=qkv.size).reshape(qkv.shape)
        np.testing.assert_array_equal(m1, g['mask1'])


@pytest.mark.parametrize('norm_first', [True, False])
def test_encoder_init_draw_order(norm_first):
    """Seeded param draws reproduce the reference's lazy-initialisation order."""
    g = load_golden('encoder_prenorm' if norm_first else 'encoder_postnorm')
    # make_golden draws qkv first (rand) and the params lazily at the first call:
    np.random.seed(0)
    qkv = np.random.normal(size=g['qkv'].shape).astype(np.float32)
    p = O.encoder_init_ordered(int(g['heads']), int(g['hidden']), qkv.shape[-1], norm_first)
    np.testing.assert_array_equal(qkv, g['qkv'])
    for k, v in p.items():
        np.testing.assert_array_equal(v, g[k + '__0'], err_msg=k)


@pytest.mark.parametrize('opt', ['sgd', 'adam'])
def test_train_mlp_trajectory(opt):
    """train_test.py MLP flow: 10 steps + eval; losses are the reference's printed values."""
    g = load_golden('train_mlp_' + opt)
    np.random.seed(0)
    feats = [16, 32, 64, 32, 16]
    x = np.random.uniform(0.0, 1.0, size=[128, 16]).astype(np.float32)
    t = np.random.uniform(0.0, 1.0, size=[128, 16]).astype(np.float32)
    np.testing.assert_array_equal(x, g['x'])
    params, states = [], []
    losses = []
    lr = float(g['lr'])
    for step in range(11):
        acts, pres = [x], []
        for i, f in enumerate(feats):
            if step == 0:
                w = O.random_init([acts[-1].shape[-1], f])
                b = O.random_init([f])
                params.append([w, b])
                states.append([{}, {}])
            y, pre = O.dense_fwd(acts[-1], *params[i])
            acts.append(y)
            pres.append(pre)
        losses.append(O.mse_fwd(acts[-1], t))
        if step == 10:
            break
        dy = O.mse_bwd(acts[-1], t)
        for i in reversed(range(len(feats))):
            dy, dw, db = O.dense_bwd(acts[i], params[i][0], pres[i], dy)
            if opt == 'sgd':
                params[i][0] = O.sgd_step(params[i][0], dw, lr)
                params[i][1] = O.sgd_step(params[i][1], db, lr)
            else:
                params[i][0] = O.adam_step(params[i][0], dw, states[i][0], lr)
                params[i][1] = O.adam_step(params[i][1], db, states[i][1], lr)
    np.testing.assert_allclose(losses, g['losses'], rtol=2e-5)
    for i in range(len(feats)):
        close(params[i][0], g[f'w{i}'], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('name', ['decoder_prenorm', 'decoder_postnorm'])
@pytest.mark.parametrize('verbatim', [True, False])
def test_decoder(name, verbatim):
    g = load_golden(name)
    nf = bool(g['norm_first'])
    p = {k[:-3]: v for k, v in g.items() if k.endswith('__0')}
    # Two attention stages on unscaled N(0,1) parameters: the softmaxes are near one-hot and
    # amplify the fp32 rounding of the reference's own einsum projections, so the GEMM-ordered
    # flavour differs from the reference by up to 4e-5 of the tensor scale (the verbatim flavour,
    # same summation order, agrees to 1e-5).  BASELINE.json's bound is 1e-4.
    tol = 1e-5 if verbatim else 1e-4
    out, cache = O.decoder_fwd(p, g['q'], g['kv'], nf, verbatim=verbatim)
    close_scaled(out, g['out'], tol)
    (dq, dkv), grads = O.decoder_bwd(p, cache, g['dy'], nf, verbatim=verbatim)
    close_scaled(dq, g['dq'], tol)
    close_scaled(dkv, g['dkv'], tol)
    assert len(grads) == 26
    for k, grad in grads.items():
        close_scaled(O.sgd_step(p[k], grad, float(g['lr'])), g[k + '__1'], tol)


@pytest.mark.parametrize('name', ['decoder_dropout_prenorm', 'decoder_dropout_postnorm'])
def test_decoder_with_dropout(name):
    """drop_rate = 0.1 in the decoder (transformer.py:98,111-113): with the three masks the reference drew, the oracle reproduces
    its output, (dq, dkv) and all 26 updated parameters (same-order flavour; tolerances as in test_decoder)."""
    g = load_golden(name)
    nf = bool(g['norm_first'])
    p = {k[:-3]: v for k, v in g.items() if k.endswith('__0')}
    drop = (g['mask1'], g['mask2'], g['mask3'], 1.0 - float(g['drop_rate']))
    assert g['mask3'].shape == (g['q'].shape[0] * g['q'].shape[1], g['q'].shape[2]) and 0.8 < g['mask2'].mean() < 0.97
    out, cache = O.decoder_fwd(p, g['q'], g['kv'], nf, verbatim=True, drop=drop)
    close_scaled(out, g['out'], 1e-5)
    (dq, dkv), grads = O.decoder_bwd(p, cache, g['dy'], nf, verbatim=True)
    close_scaled(dq, g['dq'], 1e-5)
    close_scaled(dkv, g['dkv'], 1e-5)
    assert len(grads) == 26
    for k, grad in grads.items():
        close_scaled(O.sgd_step(p[k], grad, float(g['lr'])), g[k + '__1'], 1e-5)


def test_losses():
    g = load_golden('losses')
    np.testing.assert_allclose(O.mse_fwd(g['y'], g['t']), g['mse'], rtol=1e-6)
    close(O.mse_bwd(g['y'], g['t']), g['mse_grad'])
    np.testing.assert_allclose(O.xent_fwd(g['prob'], g['onehot']), g['ce'], rtol=1e-6)
    close(O.xent_bwd(g['prob'], g['onehot']), g['ce_grad'])


@pytest.mark.parametrize('verbatim', [True, False])
def test_softmax_cross_entropy_chain(verbatim):
    """loss_test.py:49-66: ce(softmax(y), t) forward, softmax.backward(ce.backward()) backward; the closed form
    of the composition is the textbook softmax(y) * sum(t) - t."""
    g = load_golden('softmax_ce')
    prob = O.softmax_fwd(g['y'])
    close(prob, g['prob'])
    np.testing.assert_allclose(O.xent_fwd(prob, g['targets']), g['ce'], rtol=1e-6)
    dprob = O.xent_bwd(prob, g['targets'])
    close(dprob, g['dprob'], rtol=1e-5, atol=1e-4)
    dy = O.softmax_bwd(prob, dprob, verbatim=verbatim)
    close(dy, g['dy'], rtol=1e-5, atol=1e-6)
    close(dy, prob * g['targets'].sum(axis=-1, keepdims=True) - g['targets'], rtol=1e-5, atol=1e-6)


def test_attention_core_restatements_agree():
    """The one-shot attention core (attentions.py:103-112) and the blockwise online-softmax forward the reference
    derives in attentions_test.py:194-246 are the same function; the core composed with the projections is
    mha_fwd / mha_bwd, which the golden MHA fixtures pin."""
    rng = np.random.default_rng(0)
    q = rng.standard_normal([2, 70, 3, 16])
    k, v = rng.standard_normal([2, 45, 3, 16]), rng.standard_normal([2, 45, 3, 16])
    scale = 0.25
    ctx, lse, probs = O.attention_core_fwd(q, k, v, scale)
    ctx2, lse2 = O.attention_core_fwd_blockwise(q, k, v, scale)
    np.testing.assert_allclose(ctx2, ctx, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(lse2, lse, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(probs.sum(-1), 1.0, atol=1e-12)
    # against the golden self-attention fixture: the middle of mha_fwd / mha_bwd IS the core
    g = load_golden('mha_self')
    p = {n: g[n + '0'].astype(np.float64) for n in O.MHA_PARAM_NAMES}
    out, cache = O.mha_fwd(p, g['query'].astype(np.float64))
    core_ctx, _, core_probs = O.attention_core_fwd(cache['q'], cache['k'], cache['v'], 1 / np.sqrt(cache['q'].shape[-1]))
    np.testing.assert_allclose(core_probs, cache['scores'], atol=1e-12)
    np.testing.assert_allclose(core_ctx.transpose(0, 2, 1, 3), cache['values'], atol=1e-12)
    # the gradient against central differences of the forward (moderate scores: the random operands above)
    dctx = rng.standard_normal(ctx.shape)
    dq, dk, dv = O.attention_core_bwd(q, k, v, probs, dctx, scale)
    eps = 1e-6
    for which, grad, idx in ((0, dq, (1, 3, 2, 5)), (2, dv, (0, 7, 1, 2)), (1, dk, (1, 4, 0, 1))):
        def loss(delta):
            ops = [q.copy(), k.copy(), v.copy()]
            ops[which][idx] += delta
            return (O.attention_core_fwd(*ops, scale)[0] * dctx).sum()
        np.testing.assert_allclose((loss(eps) - loss(-eps)) / (2 * eps), grad[idx], rtol=1e-5, atol=1e-8)


def test_philox_known_answers():
    """Philox4x32-10 known-answer vectors published with the generator (Random123 kat_vectors): they pin the oracle's
    restatement, which in turn pins the device kernel's masks bit for bit (tests/test_gpu_rowops.py)."""
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for counter, key, want in kat:
        got = O.philox4x32_10(np.array([counter], dtype=np.uint32), key)[0]
        assert tuple(int(v) for v in got) == want
    mask = O.dropout_philox_mask(100003, 0.75, seed=1234, offset=7)
    assert mask.shape == (100003,) and abs(mask.mean() - 0.75) < 0.01
    assert not np.array_equal(mask, O.dropout_philox_mask(100003, 0.75, seed=1234, offset=8))
    assert O.dropout_philox_mask(10, 1.0, 0, 0).all()


def test_f16x2_split_arithmetic():
    """The scaled two-way fp16 split of the product's opt-in NPM_MATH_F16X2 mode, restated in NumPy: with ideal
    accumulation its error against fp64, relative to the largest element of the output row, is the representation's
    2^-22-class error alone (rms 3e-8, worst 1.3e-7 at K = 1024; NumPy's own fp32 matmul: rms 1.3e-7, worst 8.6e-7) -- the
    same for Gaussian rows, rows 2^40 apart and elements 2^30 apart inside a row; zero rows and columns are exact."""
    rng = np.random.default_rng(0)
    m, n, k = 96, 80, 1024
    base = rng.standard_normal((m, k), dtype=np.float32)
    b = (rng.standard_normal((k, n), dtype=np.float32) / 32).astype(np.float32)
    cases = {'gaussian': base,
             'rows 2^40 apart': (base * np.exp2(rng.integers(-20, 21, size=(m, 1)))).astype(np.float32),
             'elements 2^30 apart': (base * np.exp2(-rng.integers(0, 31, size=(1, k)))).astype(np.float32)}
    for name, a in cases.items():
        ref = a.astype(np.float64) @ b.astype(np.float64)
        err = np.abs(O.gemm_f16x2(a, b) - ref) / np.abs(ref).max(axis=1, keepdims=True)
        f32 = np.abs((a @ b).astype(np.float64) - ref) / np.abs(ref).max(axis=1, keepdims=True)
        assert err.max() < 2e-7 and np.sqrt((err ** 2).mean()) < 5e-8, (name, err.max())
        assert np.sqrt((err ** 2).mean()) < 0.5 * np.sqrt((f32 ** 2).mean()), name
    a = base.copy()
    a[3] = 0
    bb = b.copy()
    bb[:, 5] = 0
    c = O.gemm_f16x2(a, bb)
    assert np.all(c[3] == 0) and np.all(c[:, 5] == 0)
    s = O.f16x2_scales(base, 1)
    assert np.all((np.abs(base).max(axis=1, keepdims=True) * s >= 2.0 ** 13) & (np.abs(base).max(axis=1, keepdims=True) * s < 2.0 ** 14))

