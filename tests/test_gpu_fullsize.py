"""GPU: parity at BASELINE.json's FULL sizes through size-independent properties.

The oracle cannot run a 256 x 512 x 1024 encoder step in seconds, but every op on the path is
per-sample (LayerNorm and softmax per row, attention per (batch, head)); only parameter gradients
sum over the batch.  So at full size:
  * the output / input-gradient of any batch slice equals the same slice run ALONE, which the oracle
    checks at a size it finishes in seconds (batch 2);
  * parameter gradients are additive over batch chunks (full batch == sum of two halves);
  * GEMM checksums: (A B) 1 == A (B 1) at the C2 size;
  * softmax rows sum to 1, and the fused composition equals the literal reference order.
"""

import numpy as np
import pytest

from conftest import assert_close
from oracle import np_oracle as O

pytestmark = pytest.mark.gpu

B, S, F, H, U = 256, 512, 1024, 8, 4096


@pytest.fixture(scope='module')
def npm():
    import np_modeling_amd
    return np_modeling_amd


@pytest.fixture(params=['f32', 'bf16x3', 'f16x2'])
def exact_modes(request, npm):
    """The full-size identities hold at the same tolerances under the exact-f32 MFMA and under the unbiased
    split-bf16 mode ('bf16x3_fast' drifts by its documented bias in the 4096-term checksums)."""
    npm.set_math(request.param)
    yield request.param
    npm.set_math('f32')


class GradRecorder:
    """An optimizer that records gradients instead of applying them (parameters stay fixed)."""

    def __init__(self):
        self.grads = {}

    def update(self, obj, attribute, gradient):
        self.grads[(type(obj).__name__, attribute, id(obj))] = gradient


def _encoder(npm, rng):
    enc = npm.layers.TransformerEncoder(num_heads=H, hidden_units=U, norm_first=True)
    enc(npm.as_device(np.zeros([1, 8, F], dtype=np.float32)))
    att = enc._self_attention
    p = {}

    def put(obj, attr, key, scale):
        arr = np.asarray(getattr(obj, attr))
        new = (np.clip(rng.standard_normal(arr.shape), -1, 1) * scale).astype(np.float32)
        getattr(obj, attr).set(new)
        p[key] = new.astype(np.float64)

    for n in ('wq', 'wk', 'wv', 'wo'):
        put(att, '_' + n, 'att_' + n, 1 / np.sqrt(F))
    for n in ('bq', 'bk', 'bv', 'bo'):
        put(att, '_' + n, 'att_' + n, 1.0)
    put(enc._norm1, '_gamma', 'n1_gamma', 1.0); put(enc._norm1, '_beta', 'n1_beta', 1.0)
    put(enc._norm2, '_gamma', 'n2_gamma', 1.0); put(enc._norm2, '_beta', 'n2_beta', 1.0)
    put(enc._dense1._linear, '_w', 'd1_w', 1 / np.sqrt(F)); put(enc._dense1._linear, '_b', 'd1_b', 1.0)
    put(enc._dense2, '_w', 'd2_w', 1 / np.sqrt(U)); put(enc._dense2, '_b', 'd2_b', 1.0)
    return enc, p


def test_encoder_full_size_slices_and_additivity(npm, exact_modes):
    """BASELINE configs[4] per-GPU shard: d_model 1024, 8 heads, seq 512, batch 256, U = 4096."""
    D = npm.device
    rng = np.random.default_rng(0)
    enc, p = _encoder(npm, rng)
    x = rng.standard_normal([B, S, F], dtype=np.float32)
    dy = rng.standard_normal([B, S, F], dtype=np.float32) * np.float32(0.01)
    dx_, ddy = D.from_host(x), D.from_host(dy)

    rec_full = GradRecorder()
    out = enc(dx_)
    att = enc._self_attention
    lse_full = att._lse if att._core else None          # kept by the fused attention core instead of the probabilities
    dx = enc(ddy, backprop=True, optimizer_=rec_full)
    # (1) a batch slice of the full-size run == that slice alone, checked by the oracle
    sl = slice(100, 102)
    want_out, cache = O.encoder_fwd(p, x[sl].astype(np.float64), True)
    want_dx, _ = O.encoder_bwd(p, cache, dy[sl].astype(np.float64), True)
    out_host = out.reshape(B, S * F).numpy()[sl].reshape(2, S, F)
    dx_host = dx.reshape(B, S * F).numpy()[sl].reshape(2, S, F)
    assert_close(out_host, want_out, tol=1e-5)
    if lse_full is not None:       # log-sum-exp of every (head, query) row of the sliced samples
        scaled = np.einsum('bqhd,bkhd->bhqk', cache['att']['q'], cache['att']['k']) / np.sqrt(F // H)
        top = scaled.max(-1)
        want_lse = top + np.log(np.exp(scaled - top[..., None]).sum(-1))
        np.testing.assert_allclose(lse_full.numpy()[sl], want_lse, rtol=0, atol=2e-5)
    else:                          # split-bf16 modes compose attention from GEMMs: every probability row sums to 1
        probs = att._attention_scores
        rows = probs.size // S
        total = D.empty([rows, 4])
        D.gemm(rows, 4, S, D.Mat(probs.reshape(rows, S), S), D.Mat(D.full([S, 4], 1.0), 4), D.Mat(total, 4))
        np.testing.assert_allclose(total.numpy()[:, 0], 1.0, rtol=0, atol=5e-6)
    # ReLU's derivative is discontinuous: a hidden pre-activation within fp32 rounding of 0 can take the other
    # branch than in the fp64 oracle and changes that row's dx by one w1 column.  Such rows (a handful out of 1024
    # at this size) are excluded; every other row must agree.
    near_zero = (np.abs(cache['d1_pre']) < 5e-6).any(axis=1).reshape(2, S)      # ~4096 * 0.27 * 1e-5 = 1 % of rows
    assert near_zero.mean() < 0.05
    assert_close(dx_host[~near_zero], want_dx[~near_zero], tol=1e-5)
    full = {k: np.asarray(v) for k, v in rec_full.grads.items()}
    assert len(full) == 16

    # (2) parameter gradients are additive over batch chunks
    acc = {}
    for lo, hi in ((0, B // 2), (B // 2, B)):
        rec = GradRecorder()
        enc(D.from_host(x[lo:hi]))
        enc(D.from_host(dy[lo:hi]), backprop=True, optimizer_=rec)
        for k, v in rec.grads.items():
            acc[k] = acc.get(k, 0) + np.asarray(v).astype(np.float64)
    bq_scale = max(np.abs(v).max() for k, v in full.items() if k[1] == '_bq')
    for k in full:
        if k[1] == '_bk':
            # the key-bias gradient is exactly zero in real arithmetic (every row of datt sums to 0, so adding a
            # constant to all keys changes nothing): what is stored is rounding noise, held against dbq's scale
            assert np.abs(full[k]).max() < 1e-4 * bq_scale and np.abs(acc[k]).max() < 1e-4 * bq_scale
            continue
        assert_close(full[k], acc[k], tol=2e-5, what=str(k[:2]))



def test_encoder_full_size_fused_equals_unfused(npm):
    D = npm.device
    from np_modeling_amd import parallel
    rng = np.random.default_rng(1)
    enc, _ = _encoder(npm, rng)
    x = D.from_host(rng.standard_normal([64, S, F], dtype=np.float32))
    dy = D.from_host(rng.standard_normal([64, S, F], dtype=np.float32) * np.float32(0.01))
    out = enc(x).numpy()
    r1 = GradRecorder()
    dx1 = enc(dy, backprop=True, optimizer_=r1).numpy()
    ref = enc._forward_unfused(x).numpy()
    assert_close(out, ref, tol=3e-6)
    r2 = GradRecorder()
    with parallel.grad_scope(0) as scope:
        dx2 = enc._backward_unfused(dy, r2, scope).numpy()
    assert_close(dx1, dx2, tol=1e-5)
    g1 = {k: np.asarray(v) for k, v in r1.grads.items()}
    g2 = {k: np.asarray(v) for k, v in r2.grads.items()}
    assert len(g1) == 16 and g1.keys() == g2.keys()
    bq_scale = max(np.abs(v).max() for k, v in g1.items() if k[1] == '_bq')
    for k in g1:
        if k[1] == '_bk':      # rounding noise around an exact zero in both compositions (see the additivity test)
            assert np.abs(g1[k]).max() < 1e-4 * bq_scale and np.abs(g2[k]).max() < 1e-4 * bq_scale
            continue
        # two fp32 computations with different summation orders over K = 32768 rows, EACH within 1e-5 of the exact value
        # (tests/test_gpu_parity.py: the worst parameter gradient, dwq, sits at 6e-6): they may differ by the sum
        assert_close(g1[k], g2[k], tol=2e-5, what=str(k[:2]))


def test_encoder_full_size_with_dropout(npm):
    """The headline shapes (d_model 1024, 8 heads, seq 512, U 4096; batch 64 here) with drop_rate 0.1 and masks drawn on the
    device: the fused composition (both DropOuts applied inside the LayerNorm kernels, reference transformer.py:35-36,49-50 and
    normalizations.py:21-30) against (1) the oracle on two samples of the batch, evaluated with the very masks the device drew,
    and (2) the literal composition (DropOut layer -> norm -> ... -> standalone adds) under the same seed: same masks, outputs,
    input gradient and parameter gradients equal to rounding."""
    D = npm.device
    from np_modeling_amd import parallel
    rng = np.random.default_rng(6)
    b = 64
    x = rng.standard_normal([b, S, F], dtype=np.float32)
    dy = rng.standard_normal([b, S, F], dtype=np.float32) * np.float32(0.01)
    runs = {}
    try:
        for fused in (True, False):
            npm.set_dropout_rng('device', seed=4242)
            enc = npm.layers.TransformerEncoder(num_heads=H, hidden_units=U, norm_first=True, drop_rate=0.1)
            enc(npm.as_device(np.zeros([1, 8, F], dtype=np.float32)))          # lazy init (consumes Philox offsets 0, 1)
            prng = np.random.default_rng(60)
            p = {}
            for key, (obj, attr, scale) in dict(
                    att_wq=(enc._self_attention, '_wq', 1 / np.sqrt(F)), att_wk=(enc._self_attention, '_wk', 1 / np.sqrt(F)),
                    att_wv=(enc._self_attention, '_wv', 1 / np.sqrt(F)), att_wo=(enc._self_attention, '_wo', 1 / np.sqrt(F)),
                    att_bq=(enc._self_attention, '_bq', 1.0), att_bk=(enc._self_attention, '_bk', 1.0),
                    att_bv=(enc._self_attention, '_bv', 1.0), att_bo=(enc._self_attention, '_bo', 1.0),
                    n1_gamma=(enc._norm1, '_gamma', 1.0), n1_beta=(enc._norm1, '_beta', 1.0),
                    n2_gamma=(enc._norm2, '_gamma', 1.0), n2_beta=(enc._norm2, '_beta', 1.0),
                    d1_w=(enc._dense1._linear, '_w', 1 / np.sqrt(F)), d1_b=(enc._dense1._linear, '_b', 1.0),
                    d2_w=(enc._dense2, '_w', 1 / np.sqrt(U)), d2_b=(enc._dense2, '_b', 1.0)).items():
                arr = getattr(obj, attr)
                new = (np.clip(prng.standard_normal(arr.shape), -1, 1) * scale).astype(np.float32)
                arr.set(new)
                p[key] = new.astype(np.float64)
            rec = GradRecorder()
            if fused:
                out = enc(D.from_host(x))
                assert enc._fused
                dx = enc(D.from_host(dy), backprop=True, optimizer_=rec)
            else:
                out = enc._forward_unfused(D.from_host(x))
                with parallel.grad_scope(0) as scope:
                    dx = enc._backward_unfused(D.from_host(dy), rec, scope)
            masks = (np.asarray(enc._dropout1._mask) != 0, np.asarray(enc._dropout2._mask) != 0)
            tags = {id(enc._self_attention): 'att', id(enc._norm1): 'n1', id(enc._norm2): 'n2', id(enc._dense1._linear): 'd1',
                    id(enc._dense2): 'd2'}
            runs[fused] = (out.numpy(), dx.numpy(), masks, {(tags[k[2]], k[1]): np.asarray(v) for k, v in rec.grads.items()}, p)
    finally:
        npm.set_dropout_rng('host')
    out_f, dx_f, masks_f, grads_f, p = runs[True]
    out_u, dx_u, masks_u, grads_u, _ = runs[False]
    for a, c in zip(masks_f, masks_u):
        np.testing.assert_array_equal(a, c)
        assert abs(a.mean() - 0.9) < 1e-3
    assert_close(out_f, out_u, tol=3e-6)
    assert_close(dx_f, dx_u, tol=1e-5)
    assert grads_f.keys() == grads_u.keys() and len(grads_f) == 16
    bq_scale = max(np.abs(v).max() for k, v in grads_f.items() if k[1] == '_bq')
    for k in grads_f:
        if k[1] == '_bk':          # exactly zero in real arithmetic: rounding noise (see the additivity test above)
            assert np.abs(grads_f[k]).max() < 1e-4 * bq_scale
            continue
        assert_close(grads_f[k], grads_u[k], tol=2e-5, what=str(k))
    # the oracle on two samples, with the device's masks
    sl = slice(30, 32)
    m1, m2 = masks_f[0][sl], masks_f[1].reshape(b, S, F)[sl].reshape(2 * S, F)
    want, cache = O.encoder_fwd(p, x[sl].astype(np.float64), True, drop=(m1, m2, 0.9))
    want_dx, _ = O.encoder_bwd(p, cache, dy[sl].astype(np.float64), True)
    assert_close(out_f[sl], want, tol=1e-5)
    near_zero = (np.abs(cache['d1_pre']) < 5e-6).any(axis=1).reshape(2, S)
    assert near_zero.mean() < 0.05
    assert_close(dx_f[sl][~near_zero], want_dx[~near_zero], tol=1e-5)


def test_dense_c2_checksum(npm, exact_modes):
    """BASELINE configs[1]: Dense(4096 -> 4096) + ReLU, batch 4096: (x w + b) 1 == x (w 1) + sum(b) on the
    pre-activation, ReLU bit-exact from it, and the backward identities dw 1 = x^T (g 1), 1^T dx = (1^T g) w^T."""
    D = npm.device
    n = 4096
    rng = np.random.default_rng(2)
    layer = npm.layers.Dense(units=n)
    x = rng.standard_normal([n, n], dtype=np.float32)
    layer(D.from_host(np.zeros([1, n], dtype=np.float32)))
    w = (rng.standard_normal([n, n], dtype=np.float32) / 64).astype(np.float32)
    b = rng.standard_normal([n], dtype=np.float32)
    layer.linear._w.set(w)
    layer.linear._b.set(b)
    y = layer(D.from_host(x))
    pre = layer._activation._x.numpy()
    np.testing.assert_array_equal(y.numpy(), np.maximum(pre, 0))
    w64, x64 = w.astype(np.float64), x.astype(np.float64)
    assert_close(pre.astype(np.float64).sum(axis=1), x64 @ w64.sum(axis=1) + b.astype(np.float64).sum(), tol=2e-6)
    dy = rng.standard_normal([n, n], dtype=np.float32)
    rec = GradRecorder()
    dx = layer(D.from_host(dy), backprop=True, optimizer_=rec).numpy()
    g = np.where(pre >= 0, dy, 0).astype(np.float64)
    grads = {k[1]: np.asarray(v).astype(np.float64) for k, v in rec.grads.items()}
    assert_close(grads['_b'], g.sum(axis=0), tol=2e-6)
    assert_close(grads['_w'].sum(axis=1), x64.T @ g.sum(axis=1), tol=2e-6)
    assert_close(dx.astype(np.float64).sum(axis=0), g.sum(axis=0) @ w64.T, tol=2e-6)


def test_mha_c4_slice(npm, exact_modes):
    """BASELINE configs[3]: MHA d_model 1024, 8 heads, seq 512, batch 256 -- two samples of the full-size run
    against the oracle (attention is per sample), parameter gradients additive over halves."""
    D = npm.device
    rng = np.random.default_rng(3)
    layer = npm.layers.MultiHeadAttention(num_heads=H)
    layer(D.from_host(np.zeros([1, 8, F], dtype=np.float32)))
    p = {}
    for n in O.MHA_PARAM_NAMES:
        arr = getattr(layer, '_' + n)
        new = (np.clip(rng.standard_normal(arr.shape), -1, 1) * (1 / np.sqrt(F) if n[0] == 'w' else 1.0)).astype(np.float32)
        arr.set(new)
        p[n] = new.astype(np.float64)
    q = rng.standard_normal([B, S, F], dtype=np.float32)
    dy = rng.standard_normal([B, S, F], dtype=np.float32) * np.float32(0.01)
    out = layer(D.from_host(q))
    rec = GradRecorder()
    dq, dk, dv = layer(D.from_host(dy), backprop=True, optimizer_=rec)
    sl = slice(7, 9)
    want, cache = O.mha_fwd(p, q[sl].astype(np.float64))
    (wq_, wk_, wv_), _ = O.mha_bwd(p, cache, dy[sl].astype(np.float64))
    assert_close(out.reshape(B, S * F).numpy()[sl].reshape(2, S, F), want, tol=1e-5)
    got = sum(a.reshape(B, S * F).numpy()[sl].reshape(2, S, F).astype(np.float64) for a in (dq, dk, dv))
    assert_close(got, wq_ + wk_ + wv_, tol=1e-5)
    full = {k[1]: np.asarray(v) for k, v in rec.grads.items()}
    acc = {}
    for lo, hi in ((0, B // 2), (B // 2, B)):
        r = GradRecorder()
        layer(D.from_host(q[lo:hi]))
        layer(D.from_host(dy[lo:hi]), backprop=True, optimizer_=r)
        for k, v in r.grads.items():
            acc[k[1]] = acc.get(k[1], 0) + np.asarray(v).astype(np.float64)
    for k in full:
        if k == '_bk':      # exactly zero in real arithmetic (rows of datt sum to 0): rounding noise, held against dbq's scale
            bound = 1e-4 * np.abs(full['_bq']).max()
            assert np.abs(full[k]).max() < bound and np.abs(acc[k]).max() < bound
            continue
        assert_close(full[k], acc[k], tol=2e-5, what=k)


@pytest.mark.parametrize('heads,causal', [(8, True), (16, False), (16, True)])
def test_mha_c4_shape_masked_and_head_size_64(npm, heads, causal):
    """C4's dimensions (d_model 1024, seq 512, batch 256) under a CAUSAL mask -- the tile summary at work at full size: a
    prebuilt device mask, tiles skipped in forward and backward (mha_bwd8_kernel) -- and with 16 heads of 64 (the 8-wave
    backward at head size 64): two samples of the full-size run against the oracle, the input gradient, and the parameter
    gradients additive over batch halves."""
    D = npm.device
    rng = np.random.default_rng(heads + causal)
    layer = npm.layers.MultiHeadAttention(num_heads=heads)
    layer(D.from_host(np.zeros([1, 8, F], dtype=np.float32)))
    p = {}
    for n in O.MHA_PARAM_NAMES:
        arr = getattr(layer, '_' + n)
        new = (np.clip(rng.standard_normal(arr.shape), -1, 1) * (1 / np.sqrt(F) if n[0] == 'w' else 1.0)).astype(np.float32)
        arr.set(new)
        p[n] = new.astype(np.float64)
    q = rng.standard_normal([B, S, F], dtype=np.float32)
    dy = rng.standard_normal([B, S, F], dtype=np.float32) * np.float32(0.01)
    tri = np.tril(np.ones([S, S], dtype=bool))
    kw = lambda b: dict(mask=D.AttnMask(tri[None, None], b, heads, S, S)) if causal else {}
    out = layer(D.from_host(q), **kw(B))
    assert layer._core and npm._C.last_attn_kernel().startswith((f'mha_fwd8_kernel D={F // heads} mask={int(causal)}',
                                                                 f'mha_fwd_kernel D={F // heads} mask={int(causal)}'))
    rec = GradRecorder()
    dq, dk, dv = layer(D.from_host(dy), backprop=True, optimizer_=rec)
    assert npm._C.last_attn_kernel().startswith('mha_bwd8_kernel' if (causal or heads == 16) else 'mha_bwd16_kernel')
    sl = slice(101, 103)
    full_mask = np.broadcast_to(tri, (2, heads, S, S)) if causal else None
    want, cache = O.mha_fwd(p, q[sl].astype(np.float64), mask=full_mask)
    (wq_, wk_, wv_), _ = O.mha_bwd(p, cache, dy[sl].astype(np.float64))
    assert_close(out.reshape(B, S * F).numpy()[sl].reshape(2, S, F), want, tol=1e-5)
    got = sum(a.reshape(B, S * F).numpy()[sl].reshape(2, S, F).astype(np.float64) for a in (dq, dk, dv))
    assert_close(got, wq_ + wk_ + wv_, tol=1e-5)
    full = {k[1]: np.asarray(v) for k, v in rec.grads.items()}
    acc = {}
    for lo, hi in ((0, B // 2), (B // 2, B)):
        r = GradRecorder()
        layer(D.from_host(q[lo:hi]), **kw(hi - lo))
        layer(D.from_host(dy[lo:hi]), backprop=True, optimizer_=r)
        for k, v in r.grads.items():
            acc[k[1]] = acc.get(k[1], 0) + np.asarray(v).astype(np.float64)
    for k in full:
        if k == '_bk':
            bound = 1e-4 * np.abs(full['_bq']).max()
            assert np.abs(full[k]).max() < bound and np.abs(acc[k]).max() < bound
            continue
        assert_close(full[k], acc[k], tol=2e-5, what=k)


def test_dense_c1(npm, exact_modes):
    """BASELINE configs[0]: Dense(512 -> 512), batch 64 -- the whole layer against the oracle."""
    D = npm.device
    rng = np.random.default_rng(4)
    layer = npm.layers.Dense(units=512)
    x = rng.standard_normal([64, 512], dtype=np.float32)
    dy = rng.standard_normal([64, 512], dtype=np.float32)
    layer(D.from_host(np.zeros([1, 512], dtype=np.float32)))
    w = (np.clip(rng.standard_normal([512, 512]), -1, 1) / np.sqrt(512)).astype(np.float32)
    b = np.clip(rng.standard_normal([512]), -1, 1).astype(np.float32)
    layer.linear._w.set(w)
    layer.linear._b.set(b)
    y = layer(x)
    want_y, pre = O.dense_fwd(x.astype(np.float64), w.astype(np.float64), b.astype(np.float64))
    assert_close(y, want_y, tol=2e-6)
    rec = GradRecorder()
    dx = layer(dy, backprop=True, optimizer_=rec)
    wdx, wdw, wdb = O.dense_bwd(x.astype(np.float64), w.astype(np.float64), pre, dy.astype(np.float64))
    grads = {k[1]: np.asarray(v) for k, v in rec.grads.items()}
    safe = ~(np.abs(pre) < 2e-6).any(axis=1)            # rows whose ReLU branch cannot differ from the fp64 oracle's
    assert safe.mean() > 0.9
    assert_close(np.asarray(dx)[safe], wdx[safe], tol=2e-6)
    if safe.all():
        assert_close(grads['_w'], wdw, tol=2e-6)
        assert_close(grads['_b'], wdb, tol=2e-6)


def test_conv_c3_slice(npm):
    """BASELINE configs[2] at FULL size: Conv2D(64 -> 128, k = 3) on x[256, 224, 224, 64] (y is 6.6 GB, 12.8 M
    pixels: the 32-bit descriptor arithmetic of csrc/npm_conv.hip meets its largest offsets here).  Convolution is
    per sample, so the first, a middle and the LAST sample (highest addresses) of the full-size run are compared
    whole -- image-border rows and columns included -- with the oracle at batch 1 (reference conv.py:44-61,74-194);
    the filter and bias gradients, which sum over all 12.8 M pixels, must equal the sum over two batch halves run
    on their own (other base addresses, other pixel indices), and a one-hot dy probe pins the filter gradient of
    the last sample to the oracle.  Runs under the exact-f32 MFMA and the split-bf16 mode on the same data."""
    D = npm.device
    n, hw, c0, c1, k = 256, 224, 64, 128, 3
    rng = np.random.default_rng(5)
    layer = npm.layers.Conv2D(channels=c1, kernel_size=k)
    layer(D.from_host(np.zeros([1, 4, 4, c0], dtype=np.float32)))
    w = (np.clip(rng.standard_normal([k, k, c0, c1]), -1, 1) / np.sqrt(k * k * c0)).astype(np.float32)
    b = np.clip(rng.standard_normal([c1]), -1, 1).astype(np.float32)
    layer._w.set(w)
    layer._b.set(b)
    # distinct data in every sample, drawn one sample at a time (bounded host memory)
    x_dev, dy_dev = D.empty([n, hw, hw, c0]), D.empty([n, hw, hw, c1])
    picks = (0, 131, n - 1)
    kept = {}
    per_x, per_y = hw * hw * c0, hw * hw * c1
    for s in range(n):
        xs = rng.standard_normal([hw, hw, c0], dtype=np.float32)
        ds = rng.standard_normal([hw, hw, c1], dtype=np.float32) * np.float32(0.01)
        x_dev.flat_view(s * per_x, [hw, hw, c0]).set(xs)
        dy_dev.flat_view(s * per_y, [hw, hw, c1]).set(ds)
        if s in picks:
            kept[s] = (xs, ds)
    w64, b64 = w.astype(np.float64), b.astype(np.float64)
    want = {}
    for s in picks:
        xs, ds = kept[s]
        want_y, want_pre = O.conv_layer_fwd(xs[None].astype(np.float64), w64, b64)
        want_dx, _, _ = O.conv_layer_bwd(xs[None].astype(np.float64), w64, want_pre, ds[None].astype(np.float64))
        # ReLU's derivative is discontinuous: a pre-activation within fp32 rounding of 0 may take the other branch
        # than in the fp64 oracle, which changes dx in its k x k neighbourhood; those pixels are left out
        near = (np.abs(want_pre[0]) < 4e-6).any(axis=-1)
        grown = near.copy()
        for di in (-1, 0, 1):
            for dj in (-1, 0, 1):
                grown |= np.roll(np.roll(near, di, axis=0), dj, axis=1)
        assert grown.mean() < 0.02
        # the border rows / columns stay part of the comparison
        assert not grown[0].all() and not grown[-1].all() and not grown[:, 0].all() and not grown[:, -1].all()
        want[s] = (want_pre, want_dx, grown)
    probe = np.zeros([hw, hw, c1], dtype=np.float32)
    probe[:, :, 5] = 1.0
    want_probe = O.conv2d_grad_w(probe[None].astype(np.float64), kept[n - 1][0][None].astype(np.float64), k)

    try:
        for mode in ('f32', 'bf16x3'):
            npm.set_math(mode)
            rec = GradRecorder()
            y = layer(x_dev)
            pre = layer._activation._x
            dx = layer(dy_dev, backprop=True, optimizer_=rec)
            full = {key[1]: np.asarray(v).astype(np.float64) for key, v in rec.grads.items()}
            for s in picks:
                want_pre, want_dx, grown = want[s]
                got_pre = pre.flat_view(s * per_y, [1, hw, hw, c1]).numpy()
                got_y = y.flat_view(s * per_y, [1, hw, hw, c1]).numpy()
                assert_close(got_pre, want_pre, tol=3e-6, what=f'{mode} pre[{s}]')
                np.testing.assert_array_equal(got_y, np.maximum(got_pre, 0))
                got_dx = dx.flat_view(s * per_x, [1, hw, hw, c0]).numpy()
                assert_close(got_dx[0][~grown], want_dx[0][~grown], tol=3e-6, what=f'{mode} dx[{s}]')
            del y, dx, pre
            # filter / bias gradients: full batch == sum of the two halves run separately
            acc = {}
            half = n // 2
            for lo in (0, half):
                r = GradRecorder()
                layer(x_dev.flat_view(lo * per_x, [half, hw, hw, c0]))
                layer(dy_dev.flat_view(lo * per_y, [half, hw, hw, c1]), backprop=True, optimizer_=r)
                for key, v in r.grads.items():
                    acc[key[1]] = acc.get(key[1], 0) + np.asarray(v).astype(np.float64)
            assert_close(full['_w'], acc['_w'], tol=2e-5, what=f'{mode} dw additivity')
            assert_close(full['_b'], acc['_b'], tol=2e-5, what=f'{mode} db additivity')
            # an independent anchor for the filter gradient at the highest addresses: with dy = 1 on ONE channel of
            # every pixel of the LAST sample (0 elsewhere), dw is the shifted-window sum of that sample's x
            gprobe = D.zeros([n, hw, hw, c1])
            gprobe.flat_view((n - 1) * per_y, [hw, hw, c1]).set(probe)
            dwp = D.empty([k, k, c0, c1])
            npm._C.check(npm._C.lib().npm_conv2d_bwd_w(gprobe.ptr, x_dev.ptr, dwp.ptr, n, hw, hw, c0, c1, k))
            assert_close(dwp, want_probe, tol=2e-5, what=f'{mode} dw probe (last sample)')
            del gprobe
    finally:
        npm.set_math('f32')


_DEC_PARAMS = dict(n1_gamma=('_norm1', '_gamma'), n1_beta=('_norm1', '_beta'), n2_gamma=('_norm2', '_gamma'),
                   n2_beta=('_norm2', '_beta'), n3_gamma=('_norm3', '_gamma'), n3_beta=('_norm3', '_beta'),
                   d1_w=('_dense1._linear', '_w'), d1_b=('_dense1._linear', '_b'), d2_w=('_dense2', '_w'), d2_b=('_dense2', '_b'))
for _n in O.MHA_PARAM_NAMES:
    _DEC_PARAMS['sa_' + _n] = ('_self_attention', '_' + _n)
    _DEC_PARAMS['ca_' + _n] = ('_cross_attention', '_' + _n)


def _decoder(npm, rng, norm_first):
    """TransformerDecoder at d_model 1024 / 8 heads / U 4096 with O(1) activations: weight matrices drawn like the
    reference's initializer and scaled by 1 / sqrt(fan_in), biases and LayerNorm parameters as drawn."""
    dec = npm.layers.TransformerDecoder(num_heads=H, hidden_units=U, norm_first=norm_first)
    dec(npm.as_device(np.zeros([1, 8, F], dtype=np.float32)), npm.as_device(np.zeros([1, 8, F], dtype=np.float32)))
    p = {}
    for key, (path, attr) in _DEC_PARAMS.items():
        obj = dec
        for part in path.split('.'):
            obj = getattr(obj, part)
        arr = getattr(obj, attr)
        fan_in = U if key == 'd2_w' else F
        scale = 1 / np.sqrt(fan_in) if arr.ndim > 1 else 1.0
        new = (np.clip(rng.standard_normal(arr.shape), -1, 1) * scale).astype(np.float32)
        arr.set(new)
        p[key] = new.astype(np.float64)
    return dec, p


@pytest.mark.parametrize('norm_first', [True, False])
def test_decoder_slice(npm, norm_first):
    """TransformerDecoder (reference transformer.py:95-203) AT SIZE: batch 64, Sq 512, Skv 1024 (cross-attention with
    Sq != Skv at head size 128 through the fused core, inside the layer), d_model 1024, 8 heads, U 4096.
      * two samples of the full-size run against the oracle at O(1) activations (every op is per sample);
      * the 26 parameter gradients are additive over batch halves;
      * the fused composition (residuals / dkey + dvalue / dq + dk + dv in GEMM epilogues and the LayerNorm backward)
        equals the literal reference order built from standalone kernels."""
    from np_modeling_amd import parallel
    D = npm.device
    b, sq, skv = 64, 512, 1024
    rng = np.random.default_rng(11 + int(norm_first))
    dec, p = _decoder(npm, rng, norm_first)
    q = rng.standard_normal([b, sq, F], dtype=np.float32)
    kv = rng.standard_normal([b, skv, F], dtype=np.float32)
    dy = rng.standard_normal([b, sq, F], dtype=np.float32) * np.float32(0.01)
    dq_, dkv_, ddy = D.from_host(q), D.from_host(kv), D.from_host(dy)

    rec_full = GradRecorder()
    out = dec(dq_, dkv_)
    assert dec._cross_attention._core and dec._self_attention._core            # the fused attention core ran
    gq, gkv = dec(ddy, backprop=True, optimizer_=rec_full)
    out_host = out.numpy()
    gq_host, gkv_host = gq.numpy(), gkv.numpy()
    full = {k: np.asarray(v).astype(np.float64) for k, v in rec_full.grads.items()}
    assert len(full) == 26

    # (1) a batch slice of the full-size run == that slice alone, by the oracle
    sl = slice(30, 32)
    want_out, cache = O.decoder_fwd(p, q[sl].astype(np.float64), kv[sl].astype(np.float64), norm_first)
    assert_close(out_host[sl], want_out, tol=1e-5)
    # ReLU's derivative is discontinuous: a hidden pre-activation within fp32 rounding of 0 may take the other branch
    # than in the fp64 oracle, and in a decoder that one row's gradient reaches every key through the cross-attention.
    # So the oracle's backward runs on the kernel's OWN branch decisions -- after checking that the kernel's
    # pre-activations are the oracle's and that every differing decision sits inside the rounding band.
    pre_gpu = dec._dense1._activation._x.flat_view(sl.start * sq * U, [2 * sq, U]).numpy()
    assert_close(pre_gpu, cache['d1_pre'], tol=1e-5)
    flips = (pre_gpu >= 0) != (cache['d1_pre'] >= 0)
    assert flips.mean() < 1e-4 and (np.abs(cache['d1_pre'][flips]) < 5e-6).all()
    cache['d1_pre'] = pre_gpu.astype(np.float64)
    (want_dq, want_dkv), _ = O.decoder_bwd(p, cache, dy[sl].astype(np.float64), norm_first)
    assert_close(gq_host[sl], want_dq, tol=1e-5)
    assert_close(gkv_host[sl], want_dkv, tol=1e-5)

    # (2) parameter gradients are additive over batch halves
    acc = {}
    for lo, hi in ((0, b // 2), (b // 2, b)):
        rec = GradRecorder()
        dec(D.from_host(q[lo:hi]), D.from_host(kv[lo:hi]))
        dec(D.from_host(dy[lo:hi]), backprop=True, optimizer_=rec)
        for k, v in rec.grads.items():
            acc[k] = acc.get(k, 0) + np.asarray(v).astype(np.float64)
    bq_scale = max(np.abs(v).max() for k, v in full.items() if k[1] == '_bq')
    for k in full:
        if k[1] == '_bk':              # exactly zero in real arithmetic (rows of datt sum to 0): rounding noise
            assert np.abs(full[k]).max() < 1e-4 * bq_scale and np.abs(acc[k]).max() < 1e-4 * bq_scale
            continue
        assert_close(full[k], acc[k], tol=2e-5, what=str(k[:2]))

    # (3) fused == unfused, on a quarter of the batch
    n = b // 4
    xq, xkv, xdy = D.from_host(q[:n]), D.from_host(kv[:n]), D.from_host(dy[:n])
    o1 = dec(xq, xkv).numpy()
    r1 = GradRecorder()
    dq1, dkv1 = (g.numpy() for g in dec(xdy, backprop=True, optimizer_=r1))
    o2 = dec._forward_unfused(xq, xkv).numpy()
    assert_close(o1, o2, tol=3e-6)
    r2 = GradRecorder()
    with parallel.grad_scope(0) as scope:
        dq2, dkv2 = (g.numpy() for g in dec._backward_unfused(xdy, r2, scope))
    assert_close(dq1, dq2, tol=1e-5)
    assert_close(dkv1, dkv2, tol=1e-5)
    assert r1.grads.keys() == r2.grads.keys()
    for k in r1.grads:
        a, c2 = np.asarray(r1.grads[k]), np.asarray(r2.grads[k])
        if k[1] == '_bk':
            assert np.abs(a).max() < 1e-4 * bq_scale and np.abs(c2).max() < 1e-4 * bq_scale
            continue
        assert_close(a, c2, tol=1e-5, what=str(k[:2]))


# ---- weight gradients at BASELINE's FULL contraction length, against fp64 directly ------------------------------------
# The additivity checks above compare the kernels with themselves; the tests below hold the SHIPPED split-K path (split
# counts, slab reduction, fused bias sums) against the oracle's arithmetic in fp64 at the real K: dw = x^T dy with
# K = B S = 131 072 (reference layers/mlp.py:34-35, layers/attentions.py:167-188) and Conv2D's filter gradient with
# K = 256 x 224 x 224 = 12.8 M pixels (layers/conv.py:54-56,185-194).  The fp64 side runs in chunks over K (bounded host
# memory); metric: the GEMM tests', |got - ref| <= tol (|ref| + max |ref|).  Tolerance: the SCALED bound of the parity contract
# (include/npm_hip.h NPM_PARITY_SCALED_*, 1e-5), not the 2e-6 of the short-K GEMM tests: an fp32 accumulator that takes
# 43 691 products in a row (K / 3 splits) carries eps * n / sqrt(2) of a term's magnitude -- measured 7.8e-6 of the largest
# element for dense1's dw in the exact-f32 mode (profiles/r06_full_k_weight_gradients.log); the split modes add 16 products per
# step and sit lower.  (The reference's own sgemm accumulates in fp32 too: reference layers/mlp.py:35.)
FULL_K_TOL = {'f32': 1e-5, 'bf16x3': 1e-5, 'f16x2': 1e-5}


@pytest.mark.parametrize('m,n,which', [(1024, 4096, 'bsum'), (4096, 1024, 'bsum'), (3072, 1024, 'asum'), (1024, 1024, 'asum')])
def test_weight_gradient_gemm_at_full_k_against_fp64(npm, exact_modes, m, n, which):
    """npm_sgemm TN at (M, N, K) = the encoder's weight-gradient shapes: dense1 / dense2 (bias gradient = column sums of dy:
    bsum), the packed q / k / v projection (one product, M = 3 F, bias gradient = column sums of the FIRST operand: asum) and
    the output projection."""
    D = npm.device
    k = B * S
    rng = np.random.default_rng(m * 7 + n)
    a = rng.standard_normal([k, m], dtype=np.float32)
    b = rng.standard_normal([k, n], dtype=np.float32) * np.float32(0.01)
    want = np.zeros([m, n])
    step = 8192
    for lo in range(0, k, step):
        want += a[lo:lo + step].astype(np.float64).T @ b[lo:lo + step].astype(np.float64)
    src = b if which == 'bsum' else a
    want_sum = src.astype(np.float64).sum(axis=0)
    c = D.full([m, n], np.nan)
    sums = D.full([src.shape[1]], np.nan)
    D.gemm(m, n, k, D.Mat(D.from_host(a), m), D.Mat(D.from_host(b), n), D.Mat(c, n), trans_a=True, **{which + '_out': sums})
    assert npm._C.last_math() == exact_modes
    tol = FULL_K_TOL[exact_modes]
    got = c.numpy().astype(np.float64)
    print(f'full-K dw {m}x{n}x{k} {exact_modes}: scaled error {np.abs(got - want).max() / np.abs(want).max():.2e}, '
          f'{which} {np.abs(sums.numpy() - want_sum).max() / np.abs(want_sum).max():.2e}')
    assert_close(got, want, tol=tol, what=f'dw {m}x{n}x{k} {exact_modes}')
    assert_close(sums, want_sum, tol=tol, what=f'{which} {exact_modes}')
    # the same product again: bitwise reproducible (fixed-order slab reduction, no float atomics)
    c2 = D.full([m, n], np.nan)
    D.gemm(m, n, k, D.Mat(D.from_host(a), m), D.Mat(D.from_host(b), n), D.Mat(c2, n), trans_a=True, **{which + '_out': sums})
    np.testing.assert_array_equal(c2.numpy(), c.numpy())


def test_conv_c3_filter_gradient_at_full_size_against_fp64(npm):
    """C3's dw and db over ALL 256 samples (K = 12.8 M pixels per tap) against the oracle's tap-by-tap products in fp64
    (oracle conv2d_grad_w = reference conv.py:185-194; nine dgemms per chunk of samples, summed over the chunks).  The ReLU
    mask comes from the device's own pre-activation (read back), so no branch of activations.py:19 can differ between the
    two sides.  Exact-f32 MFMA and the split-bf16 mode."""
    D = npm.device
    n, hw, c0, c1, k = 256, 224, 64, 128, 3
    rng = np.random.default_rng(11)
    layer = npm.layers.Conv2D(channels=c1, kernel_size=k)
    layer(D.from_host(np.zeros([1, 4, 4, c0], dtype=np.float32)))
    layer._w.set((np.clip(rng.standard_normal([k, k, c0, c1]), -1, 1) / np.sqrt(k * k * c0)).astype(np.float32))
    layer._b.set(np.clip(rng.standard_normal([c1]), -1, 1).astype(np.float32))
    per_x, per_y = hw * hw * c0, hw * hw * c1
    x_dev, dy_dev = D.empty([n, hw, hw, c0]), D.empty([n, hw, hw, c1])
    # 16 drawn samples; sample s = one of them, shifted along the width and scaled: all 256 differ, drawn in seconds
    base_x = rng.standard_normal([16, hw, hw, c0], dtype=np.float32)
    base_d = rng.standard_normal([16, hw, hw, c1], dtype=np.float32) * np.float32(0.01)
    for s in range(n):
        shift, gain = s // 16, np.float32(0.5 + s / n)
        x_dev.flat_view(s * per_x, [hw, hw, c0]).set(np.roll(base_x[s % 16], shift, axis=1) * gain)
        dy_dev.flat_view(s * per_y, [hw, hw, c1]).set(np.roll(base_d[s % 16], -shift, axis=0) * gain)
    del base_x, base_d
    try:
        for mode in ('f32', 'bf16x3'):
            npm.set_math(mode)
            rec = GradRecorder()
            layer(x_dev)
            pre = layer._activation._x
            layer(dy_dev, backprop=True, optimizer_=rec)
            got = {key[1]: np.asarray(v).astype(np.float64) for key, v in rec.grads.items()}
            want_w, want_b = np.zeros([k, k, c0, c1]), np.zeros([c1])
            chunk = 8
            for lo in range(0, n, chunk):
                xs = x_dev.flat_view(lo * per_x, [chunk, hw, hw, c0]).numpy().astype(np.float64)
                ds = dy_dev.flat_view(lo * per_y, [chunk, hw, hw, c1]).numpy()
                ps = pre.flat_view(lo * per_y, [chunk, hw, hw, c1]).numpy()
                g = np.where(ps >= 0, ds, np.float32(0)).astype(np.float64)            # activations.py:19 on the device's own pre
                want_w += O.conv2d_grad_w(g, xs, k)
                want_b += g.sum(axis=(0, 1, 2))
            print(f'C3 full dw {mode}: scaled error {np.abs(got["_w"] - want_w).max() / np.abs(want_w).max():.2e}, '
                  f'db {np.abs(got["_b"] - want_b).max() / np.abs(want_b).max():.2e}')
            assert_close(got['_w'], want_w, tol=FULL_K_TOL[mode], what=f'{mode} dw, all {n} samples')
            assert_close(got['_b'], want_b, tol=FULL_K_TOL[mode], what=f'{mode} db, all {n} samples')
            del pre
    finally:
        npm.set_math('f32')
