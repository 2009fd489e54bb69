"""TEST INFRASTRUCTURE ONLY -- NumPy restatement of the reference's layer math.

This is the CPU oracle of the hot path.  It is a
*functional* restatement (pure functions over explicit caches) of the algorithms
in levendlee/np-modeling; every function cites the reference file:line it follows.

Two flavours of every backward that the reference writes with an explicit
Jacobian or a non-BLAS einsum:

* ``verbatim=True``  -- the reference's own formula (explicit Jacobians, einsum
  contractions, fp64 temporaries).  O(rows * n^2) memory: small shapes only.
* ``verbatim=False`` -- the algebraically identical closed form / single-GEMM
  view (what the HIP kernels implement, and what is timed as the CPU baseline at
  shapes where the Jacobians do not fit in memory).

Pinning: ``oracle/make_golden.py`` imports the real reference (in the build
container only) and writes ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``
checks both flavours here against those fixtures.  The reference itself stores no
golden vectors (its tests compare against JAX/Flax live), so the fixtures generated
from the reference's own execution are the pin.

dtype notes (NumPy >= 2, as measured in SURVEY.md section 8a): Linear stays fp32;
MultiHeadAttention is promoted to fp64 from the 1/sqrt(key_dim) scaling onward;
Softmax/LayerNorm backward and all of Conv2D are fp64.  The functions below follow
the same NumPy expressions so the promotions happen in the same places.
"""

from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np

Array = np.ndarray

# The dtype of the spots where the reference's NumPy expressions promote to fp64 (module docstring).  ``None`` keeps the
# reference's promotions; ``compute_in(np.float32)`` evaluates the same formulas in fp32 end to end -- the noise floor of a
# legitimate fp32 implementation, which tests/test_gpu_reference_form.py sets beside the product's error.
COMPUTE_DTYPE = None


class compute_in:
    def __init__(self, dtype):
        self.dtype = dtype

    def __enter__(self):
        global COMPUTE_DTYPE
        self.saved, COMPUTE_DTYPE = COMPUTE_DTYPE, self.dtype
        return self

    def __exit__(self, *exc):
        global COMPUTE_DTYPE
        COMPUTE_DTYPE = self.saved


def _wide(a) -> Array:
    return np.asarray(a, dtype=COMPUTE_DTYPE or np.float64)


# --------------------------------------------------------------------------- #
# initializer (reference layers/layer.py:57-60)
# --------------------------------------------------------------------------- #
def random_init(shape) -> Array:
    """N(0,1) clipped to [-1, 1], float32, drawn from the *global* np.random state."""
    draw = np.random.normal(size=shape).astype(np.float32)
    return np.clip(draw, -1.0, 1.0)


# --------------------------------------------------------------------------- #
# Linear / Dense (reference layers/mlp.py:21-40, 70-77)
# --------------------------------------------------------------------------- #
def linear_fwd(x: Array, w: Array, b: Array) -> Array:
    """y = x @ w + b   (mlp.py:23-24)."""
    return np.matmul(x, w) + b


def linear_bwd(x: Array, w: Array, dy: Array) -> Tuple[Array, Array, Array]:
    """db = sum_0 dy ; dw = x^T dy ; dx = dy w^T   (mlp.py:34-36).  2-D only."""
    assert dy.shape == (x.shape[0], w.shape[1])
    db = dy.sum(axis=0)
    dw = np.matmul(x.T, dy)
    dx = np.matmul(dy, w.T)
    return dx, dw, db


def relu_fwd(x: Array) -> Array:
    """max(x, 0)   (activations.py:15)."""
    return np.maximum(x, 0.0)


def relu_bwd(x_pre: Array, dy: Array) -> Array:
    """where(x >= 0, dy, 0) -- note the >=: gradient passes at x == 0 (activations.py:19)."""
    assert dy.shape == x_pre.shape
    return np.where(x_pre >= 0.0, dy, 0.0)


def dense_fwd(x, w, b):
    """Dense = Linear + ReLU (mlp.py:70-72).  Returns (y, pre_activation)."""
    pre = linear_fwd(x, w, b)
    return relu_fwd(pre), pre


def dense_bwd(x, w, pre, dy):
    """(mlp.py:74-77)."""
    return linear_bwd(x, w, relu_bwd(pre, dy))


# --------------------------------------------------------------------------- #
# Softmax (reference layers/activations.py:22-45)
# --------------------------------------------------------------------------- #
def softmax_fwd(x: Array) -> Array:
    """Max-shifted softmax over the last axis (activations.py:26-29)."""
    shifted = np.exp(x - x.max(axis=-1, keepdims=True))
    return shifted / shifted.sum(axis=-1, keepdims=True)


def softmax_bwd(y: Array, dy: Array, verbatim: bool = False) -> Array:
    """dx_b = sum_a dy_a * y_a (delta_ab - y_b)   (activations.py:32-45).

    verbatim: builds the [..., n, n] Jacobian with an fp64 identity exactly as the
    reference does.  closed form: dx = y * (dy - sum(dy * y)).
    """
    if verbatim:
        rank = y.ndim
        n = y.shape[-1]
        jac = np.expand_dims(np.eye(n), axis=tuple(range(rank - 1)))
        jac = jac - np.expand_dims(y, axis=rank - 1)
        jac = jac * np.expand_dims(y, axis=rank)
        return np.einsum('...a,...ba->...b', dy, jac)
    y64 = _wide(y)
    dy64 = _wide(dy)
    inner = (dy64 * y64).sum(axis=-1, keepdims=True)
    return y64 * (dy64 - inner)


# --------------------------------------------------------------------------- #
# LayerNormalization (reference layers/normalizations.py:43-75)
# --------------------------------------------------------------------------- #
def layernorm_fwd(x: Array, gamma: Array, beta: Array, eps: float):
    """Biased variance, eps inside the sqrt (normalizations.py:45-48).

    Returns (z, cache) with cache = (mean, var, yhat)."""
    mean = x.mean(axis=-1, keepdims=True)
    var = x.var(axis=-1, keepdims=True)
    yhat = (x - mean) / np.sqrt(var + eps)
    return gamma * yhat + beta, (mean, var, yhat)


def layernorm_bwd(x, gamma, eps, cache, dz, verbatim: bool = False):
    """Returns (dx, dgamma, dbeta)   (normalizations.py:50-75).

    verbatim: explicit [rows, d, d] Jacobian (fp64 because np.eye is fp64).
    closed form: dx = rstd * (g - mean(g) - yhat * mean(g * yhat)), g = dz * gamma.
    """
    mean, var, yhat = cache
    rank = x.ndim
    batch_dims = tuple(range(rank - 1))
    d = x.shape[-1]
    dbeta = dz.sum(axis=batch_dims)
    dgamma = (dz * yhat).sum(axis=batch_dims)
    g = dz * gamma
    if verbatim:
        centered = x - mean
        spread = var + eps
        dvar_dx = 2.0 * centered / d
        jac = (np.expand_dims(spread, rank) ** -0.5 *
               np.expand_dims(np.eye(d) - np.array(1.0 / d), batch_dims) -
               0.5 * np.expand_dims(spread, rank) ** -1.5 *
               np.expand_dims(dvar_dx, rank) * np.expand_dims(centered, rank - 1))
        dx = np.einsum('...a,...ab->...b', g, jac)
    else:
        g64 = _wide(g)
        yh64 = _wide(yhat)
        rstd = 1.0 / np.sqrt(_wide(var) + _wide(eps))
        dx = rstd * (g64 - g64.mean(axis=-1, keepdims=True) -
                     yh64 * (g64 * yh64).mean(axis=-1, keepdims=True))
    return dx, dgamma, dbeta


# --------------------------------------------------------------------------- #
# Conv2D, NHWC x HWIO, SAME, stride 1, odd k (reference layers/conv.py:74-194)
# --------------------------------------------------------------------------- #
def _pad_same(x: Array, k: int) -> Array:
    n, h, w, c = x.shape
    p = k // 2
    padded = np.zeros([n, h + k - 1, w + k - 1, c], dtype=COMPUTE_DTYPE or np.float64)          # fp64, as conv.py:97
    padded[:, p:h + p, p:w + p, :] = x
    return padded


def conv2d_fwd(x: Array, filt: Array) -> Array:
    """Sum over the k*k taps of shifted [NHW,C0] @ [C0,C1] matmuls (conv.py:97-105)."""
    n, h, w, c0 = x.shape
    k, k2, fc0, c1 = filt.shape
    assert k == k2 and fc0 == c0 and k % 2 == 1
    xp = _pad_same(x, k)
    out = np.zeros([n, h, w, c1], dtype=COMPUTE_DTYPE or np.float64)
    for i in range(k):
        for j in range(k):
            tap = xp[:, i:i + h, j:j + w, :].reshape(n * h * w, c0)
            out += np.matmul(tap, filt[i, j]).reshape(n, h, w, c1)
    return out


def conv2d_grad_x(dy: Array, filt: Array) -> Array:
    """conv of dy with the spatially flipped, channel-transposed filter (conv.py:130,153)."""
    flipped = np.transpose(filt[::-1, ::-1, :, :], [0, 1, 3, 2])
    return conv2d_fwd(dy, flipped)


def conv2d_grad_w(dy: Array, x: Array, k: int) -> Array:
    """dw[i,j] = shifted(x)^T @ dy  per tap (conv.py:185-194)."""
    n, h, w, c1 = dy.shape
    c0 = x.shape[-1]
    assert dy.shape[:3] == x.shape[:3] and k % 2 == 1
    xp = _pad_same(x, k)
    rhs = dy.reshape(n * h * w, c1)
    dw = np.zeros([k, k, c0, c1], dtype=COMPUTE_DTYPE or np.float64)
    for i in range(k):
        for j in range(k):
            lhs = xp[:, i:i + h, j:j + w, :].reshape(n * h * w, c0).T
            dw[i, j] += np.matmul(lhs, rhs)
    return dw


def conv_layer_fwd(x, filt, bias):
    """relu(conv(x, w) + b) (conv.py:44-48).  Returns (y, pre_activation)."""
    pre = conv2d_fwd(x, filt) + bias
    return relu_fwd(pre), pre


def conv_layer_bwd(x, filt, pre, dy):
    """Returns (dx, dw, db) (conv.py:50-61)."""
    assert dy.shape[:3] == x.shape[:3] and dy.shape[3] == filt.shape[3]
    g = relu_bwd(pre, dy)
    db = g.sum(axis=(0, 1, 2))
    dw = conv2d_grad_w(g, x, filt.shape[0])
    dx = conv2d_grad_x(g, filt)
    return dx, dw, db


# --------------------------------------------------------------------------- #
# MultiHeadAttention (reference layers/attentions.py:67-199)
# --------------------------------------------------------------------------- #
MHA_PARAM_NAMES = ('wq', 'wk', 'wv', 'wo', 'bq', 'bk', 'bv', 'bo')


def mha_init(num_heads: int, features: int, value_features: Optional[int] = None) -> Dict[str, Array]:
    """Draw order wq, wk, wv, wo, bq, bk, bv, bo (attentions.py:46-65)."""
    h = num_heads
    dk = features // h
    dv = (value_features or features) // h
    return dict(
        wq=random_init([h, dk, h * dk]), wk=random_init([h, dk, h * dk]),
        wv=random_init([h, dv, h * dv]), wo=random_init([h * dk, h, dv]),
        bq=random_init([h, dk]), bk=random_init([h, dk]),
        bv=random_init([h, dv]), bo=random_init([h * dk]))


def mha_fwd(p: Dict[str, Array], query: Array, key: Optional[Array] = None,
            value: Optional[Array] = None, verbatim: bool = False, mask: Optional[Array] = None):
    """Projections, scaled QK^T, softmax, PV, output projection (attentions.py:88-120).

    verbatim uses the reference's einsum strings; otherwise the same contractions are
    written as flat GEMMs + batched matmuls (BLAS) -- identical math.
    Returns (out, cache)."""
    key = query if key is None else key
    value = key if value is None else value
    h, dk, _ = p['wq'].shape
    dv = p['wv'].shape[1]
    b, sq, _ = query.shape
    skv = key.shape[1]
    if verbatim:
        q = np.einsum('...ab,cdb->...acd', query, p['wq']) + p['bq']
        k = np.einsum('...ab,cdb->...acd', key, p['wk']) + p['bk']
        v = np.einsum('...ab,cdb->...acd', value, p['wv']) + p['bv']
        att = np.einsum('...abc,...dbc->...bad', q, k)
    else:
        q = (query.reshape(b * sq, -1) @ p['wq'].reshape(h * dk, -1).T).reshape(b, sq, h, dk) + p['bq']
        k = (key.reshape(b * skv, -1) @ p['wk'].reshape(h * dk, -1).T).reshape(b, skv, h, dk) + p['bk']
        v = (value.reshape(b * skv, -1) @ p['wv'].reshape(h * dv, -1).T).reshape(b, skv, h, dv) + p['bv']
        att = np.matmul(q.transpose(0, 2, 1, 3), k.transpose(0, 2, 3, 1))
    scaled = (1.0 / np.sqrt(dk)) * att            # np.float64 scalar -> fp64 under NumPy 2
    if COMPUTE_DTYPE is not None:
        scaled = scaled.astype(COMPUTE_DTYPE)
    if mask is not None:                          # attentions.py:105-107 as written (see attention_core_fwd)
        scaled = np.where(mask, scaled, float('-inf'))
    scores = softmax_fwd(scaled)                  # [b, h, sq, skv]
    if verbatim:
        values = np.einsum('...abc,...cad->...abd', scores, v)
        out = np.einsum('...abc,...dac->...bd', values, p['wo']) + p['bo']
    else:
        values = np.matmul(scores, v.transpose(0, 2, 1, 3))      # [b, h, sq, dv]
        flat = values.transpose(0, 2, 1, 3).reshape(b * sq, h * dv)
        out = (flat @ p['wo'].reshape(-1, h * dv).T).reshape(b, sq, -1) + p['bo']
    cache = dict(query=query, key=key, value=value, q=q, k=k, v=v, scores=scores, values=values)
    return out, cache


def mha_bwd(p: Dict[str, Array], cache: Dict[str, Array], dy: Array, verbatim: bool = False):
    """Returns ((dquery, dkey, dvalue), grads) with grads keyed like the params
    (attentions.py:122-199)."""
    h, dk, _ = p['wq'].shape
    dv_dim = p['wv'].shape[1]
    query, key, value = cache['query'], cache['key'], cache['value']
    q, k, v, scores, values = cache['q'], cache['k'], cache['v'], cache['scores'], cache['values']
    b, sq, feat = dy.shape
    skv = key.shape[1]
    g: Dict[str, Array] = {}
    g['bo'] = dy.sum(axis=(0, 1))
    if verbatim:
        g['wo'] = np.einsum('...abc,...bd->...dac', values, dy).sum(axis=0)
        dvalues = np.einsum('...ab,bcd->...cad', dy, p['wo'])
        dscores = np.einsum('...abc,...dac->...abd', dvalues, v)
        dv = np.einsum('...abc,...abd->...cad', scores, dvalues)
    else:
        flat_values = values.transpose(0, 2, 1, 3).reshape(b * sq, h * dv_dim)
        dy2 = dy.reshape(b * sq, feat)
        g['wo'] = (dy2.T @ flat_values).reshape(feat, h, dv_dim)
        dvalues = (dy2 @ p['wo'].reshape(feat, h * dv_dim)).reshape(b, sq, h, dv_dim).transpose(0, 2, 1, 3)
        vt = v.transpose(0, 2, 1, 3)                                  # [b, h, skv, dv]
        dscores = np.matmul(dvalues, vt.transpose(0, 1, 3, 2))
        dv = np.matmul(scores.transpose(0, 1, 3, 2), dvalues).transpose(0, 2, 1, 3)
    dscaled = softmax_bwd(scores, dscores, verbatim=verbatim)
    datt = dscaled / np.sqrt(dk)
    if COMPUTE_DTYPE is not None:
        datt = datt.astype(COMPUTE_DTYPE)
    if verbatim:
        dq = np.einsum('...abc,...cad->...bad', datt, k)
        dk_ = np.einsum('...abc,...bad->...dbc', q, datt)
        w_eq, x_eq = '...ab,...acd->...cdb', '...abc,bcd->...ad'
        g['wq'] = np.einsum(w_eq, query, dq).sum(axis=0)
        dquery = np.einsum(x_eq, dq, p['wq'])
        g['wk'] = np.einsum(w_eq, key, dk_).sum(axis=0)
        dkey = np.einsum(x_eq, dk_, p['wk'])
        g['wv'] = np.einsum(w_eq, value, dv).sum(axis=0)
        dvalue = np.einsum(x_eq, dv, p['wv'])
    else:
        dq = np.matmul(datt, k.transpose(0, 2, 1, 3)).transpose(0, 2, 1, 3)          # [b, sq, h, dk]
        dk_ = np.matmul(datt.transpose(0, 1, 3, 2), q.transpose(0, 2, 1, 3)).transpose(0, 2, 1, 3)

        def in_proj_bwd(x_in, dproj, w, width):
            rows = x_in.shape[0] * x_in.shape[1]
            dflat = dproj.reshape(rows, h * width)
            dw = (dflat.T @ x_in.reshape(rows, -1)).reshape(w.shape)
            dx = (dflat @ w.reshape(h * width, -1)).reshape(x_in.shape)
            return dw, dx

        g['wq'], dquery = in_proj_bwd(query, dq, p['wq'], dk)
        g['wk'], dkey = in_proj_bwd(key, dk_, p['wk'], dk)
        g['wv'], dvalue = in_proj_bwd(value, dv, p['wv'], dv_dim)
    g['bq'] = dq.sum(axis=(0, 1))
    g['bk'] = dk_.sum(axis=(0, 1))
    g['bv'] = dv.sum(axis=(0, 1))
    return (dquery, dkey, dvalue), g


# --------------------------------------------------------------------------- #
# Attention core: scores -> softmax -> context and its gradient, on [B, S, H, D] q/k/v
# (the middle of reference layers/attentions.py: 103-112 forward, 146-162 backward)
# --------------------------------------------------------------------------- #
def attention_core_fwd(q: Array, k: Array, v: Array, scale: float, mask: Optional[Array] = None):
    """q [B,Sq,H,D], k/v [B,Skv,H,D] -> (ctx [B,Sq,H,Dv], lse [B,H,Sq], probabilities [B,H,Sq,Skv]).

    attentions.py:103-104 (QK^T einsum, 1/sqrt(Dk)), :105-107 (``np.where(mask, scaled, -inf)`` -- the line is
    unreachable for real masks in the reference because of the ``if mask:`` in front of it; restated as written),
    :108 (max-shifted softmax), :112 (PV).  lse = log sum_j exp(scaled_ij), what a fused kernel keeps instead of
    the probabilities."""
    scaled = scale * np.einsum('bqhd,bkhd->bhqk', q, k)
    if mask is not None:
        scaled = np.where(mask, scaled, -np.inf)
    top = scaled.max(axis=-1, keepdims=True)
    shifted = np.exp(scaled - top)
    total = shifted.sum(axis=-1, keepdims=True)
    probs = shifted / total
    ctx = np.einsum('bhqk,bkhd->bqhd', probs, v)
    return ctx, (top + np.log(total))[..., 0], probs


def attention_core_bwd(q: Array, k: Array, v: Array, probs: Array, dctx: Array, scale: float):
    """(dq, dk, dv) from the probabilities and dctx [B,Sq,H,Dv] (attentions.py:146-162; masked positions carry
    probability 0 and therefore no gradient -- the reference raises NotImplementedError there)."""
    dprobs = np.einsum('bqhd,bkhd->bhqk', dctx, v)
    dv = np.einsum('bhqk,bqhd->bkhd', probs, dctx)
    dscaled = softmax_bwd(probs, dprobs) * scale
    dq = np.einsum('bhqk,bkhd->bqhd', dscaled, k)
    dk = np.einsum('bhqk,bqhd->bkhd', dscaled, q)
    return dq, dk, dv


def attention_core_fwd_blockwise(q: Array, k: Array, v: Array, scale: float, q_block: int = 32, kv_block: int = 32):
    """The blockwise online-softmax forward the reference derives in layers/attentions_test.py:194-246 (running
    maximum m_i, running sum l_i, accumulator rescaled by l_i / l_i_new), restated for [B, S, H, D] operands of any
    length; returns (ctx, lse).  It is the algorithm of csrc/npm_attn.hip's forward kernel, checked here against
    the one-shot form above."""
    b, sq, h, d = q.shape
    skv = k.shape[1]
    ctx = np.zeros([b, sq, h, v.shape[3]])
    lse = np.zeros([b, h, sq])
    for q0 in range(0, sq, q_block):
        tq = q[:, q0:q0 + q_block]
        nq = tq.shape[1]
        m_i = np.full([b, h, nq], -np.inf)
        l_i = np.zeros([b, h, nq])
        acc = np.zeros([b, nq, h, v.shape[3]])
        for k0 in range(0, skv, kv_block):
            tk, tv = k[:, k0:k0 + kv_block], v[:, k0:k0 + kv_block]
            p_ij = scale * np.einsum('bqhd,bkhd->bhqk', tq, tk)
            m_new = np.maximum(p_ij.max(axis=3), m_i)
            l_i = l_i * np.exp(m_i - m_new)
            p_ij = np.exp(p_ij - m_new[..., None])
            l_new = p_ij.sum(axis=3) + l_i
            p_ij = p_ij / l_new[..., None]
            acc = acc * np.transpose((l_i / l_new)[..., None], [0, 2, 1, 3])
            m_i, l_i = m_new, l_new
            acc = acc + np.einsum('bhqk,bkhd->bqhd', p_ij, tv)
        ctx[:, q0:q0 + q_block] = acc
        lse[:, :, q0:q0 + q_block] = m_i + np.log(l_i)
    return ctx, lse


# --------------------------------------------------------------------------- #
# TransformerEncoder (reference layers/transformer.py:29-92), drop_rate == 0
# --------------------------------------------------------------------------- #
def encoder_init(num_heads: int, hidden_units: int, features: int) -> Dict[str, Array]:
    """Parameter draw order of a first forward call on a fresh encoder.

    pre-norm  : norm1(gamma,beta), MHA(8), norm2, dense1(w,b), dense2(w,b)
    post-norm : MHA(8), norm1, dense1, dense2, norm2
    Use ``encoder_init_ordered`` for the order-faithful version; this helper draws
    in the pre-norm order."""
    return encoder_init_ordered(num_heads, hidden_units, features, norm_first=True)


def encoder_init_ordered(num_heads, hidden_units, features, norm_first: bool) -> Dict[str, Array]:
    """Lazy initialisation happens at each sub-layer's first ``__call__``
    (layer.py:33-35), so the global-RNG draw order follows transformer.py:35-57."""
    p: Dict[str, Array] = {}

    def norm(tag):
        p[f'{tag}_gamma'] = random_init([features])
        p[f'{tag}_beta'] = random_init([features])

    def attn():
        for name, arr in mha_init(num_heads, features).items():
            p[f'att_{name}'] = arr

    def dense():
        p['d1_w'] = random_init([features, hidden_units])
        p['d1_b'] = random_init([hidden_units])
        p['d2_w'] = random_init([hidden_units, features])
        p['d2_b'] = random_init([features])

    if norm_first:
        norm('n1'); attn(); norm('n2'); dense()
    else:
        attn(); norm('n1'); dense(); norm('n2')
    return p


def _att_params(p):
    return {name: p[f'att_{name}'] for name in MHA_PARAM_NAMES}


def encoder_fwd(p: Dict[str, Array], qkv: Array, norm_first: bool, eps: float = 1e-3,
                eps2: Optional[float] = None, verbatim: bool = False, drop=None):
    """(transformer.py:29-59).  Returns (out, cache).  ``drop = (mask1, mask2, keep_prob)``: the masks DropOut drew
    (transformer.py:35,40,49,55 -- always directly in front of a norm; normalizations.py:21-23), shapes [B,S,F] and [B S,F]."""
    eps2 = eps if eps2 is None else eps2
    b, s, f = qkv.shape
    c: Dict[str, object] = {'drop': drop}
    d1 = (lambda t: dropout_apply(t, drop[0].reshape(t.shape), drop[2])) if drop is not None else (lambda t: t)
    d2 = (lambda t: dropout_apply(t, drop[1].reshape(t.shape), drop[2])) if drop is not None else (lambda t: t)
    skip = qkv
    h0 = qkv
    if norm_first:
        h0 = d1(h0)
        c['n1_x'] = h0
        h0, c['n1'] = layernorm_fwd(h0, p['n1_gamma'], p['n1_beta'], eps)
    out, c['att'] = mha_fwd(_att_params(p), h0, verbatim=verbatim)
    out = out + skip
    if not norm_first:
        out = d1(out)
        c['n1_x'] = out
        out, c['n1'] = layernorm_fwd(out, p['n1_gamma'], p['n1_beta'], eps)
    out = out.reshape(-1, f)
    skip = out
    if norm_first:
        out = d2(out)
        c['n2_x'] = out
        out, c['n2'] = layernorm_fwd(out, p['n2_gamma'], p['n2_beta'], eps2)
    c['d1_x'] = out
    out, c['d1_pre'] = dense_fwd(out, p['d1_w'], p['d1_b'])
    c['d2_x'] = out
    out = linear_fwd(out, p['d2_w'], p['d2_b'])
    out = out + skip
    if not norm_first:
        out = d2(out)
        c['n2_x'] = out
        out, c['n2'] = layernorm_fwd(out, p['n2_gamma'], p['n2_beta'], eps2)
    return out.reshape(b, s, f), c


def encoder_bwd(p, c, dy: Array, norm_first: bool, eps: float = 1e-3,
                eps2: Optional[float] = None, verbatim: bool = False):
    """(transformer.py:61-92).  Returns (dx, grads) -- all grads from pre-update params."""
    eps2 = eps if eps2 is None else eps2
    b, s, f = dy.shape
    g: Dict[str, Array] = {}
    drop = c.get('drop')
    d1 = (lambda t: dropout_apply(t, drop[0].reshape(t.shape), drop[2])) if drop is not None else (lambda t: t)    # normalizations.py:27-30
    d2 = (lambda t: dropout_apply(t, drop[1].reshape(t.shape), drop[2])) if drop is not None else (lambda t: t)
    dy = dy.reshape(-1, f)
    if not norm_first:
        dy, g['n2_gamma'], g['n2_beta'] = layernorm_bwd(c['n2_x'], p['n2_gamma'], eps2, c['n2'], dy, verbatim)
        dy = d2(dy)
    dskip = dy
    dy, g['d2_w'], g['d2_b'] = linear_bwd(c['d2_x'], p['d2_w'], dy)
    dy, g['d1_w'], g['d1_b'] = dense_bwd(c['d1_x'], p['d1_w'], c['d1_pre'], dy)
    if norm_first:
        dy, g['n2_gamma'], g['n2_beta'] = layernorm_bwd(c['n2_x'], p['n2_gamma'], eps2, c['n2'], dy, verbatim)
        dy = d2(dy)
    dy = (dy + dskip).reshape(b, s, f)
    if not norm_first:
        dy, g['n1_gamma'], g['n1_beta'] = layernorm_bwd(c['n1_x'], p['n1_gamma'], eps, c['n1'], dy, verbatim)
        dy = d1(dy)
    dskip = dy
    (dq, dk, dv), ga = mha_bwd(_att_params(p), c['att'], dy, verbatim=verbatim)
    for name, arr in ga.items():
        g[f'att_{name}'] = arr
    dy = dq + dk + dv                                    # np.sum(tuple, axis=0), transformer.py:85
    if norm_first:
        dy, g['n1_gamma'], g['n1_beta'] = layernorm_bwd(c['n1_x'], p['n1_gamma'], eps, c['n1'], dy, verbatim)
        dy = d1(dy)
    return dy + dskip, g


# --------------------------------------------------------------------------- #
# TransformerDecoder (reference layers/transformer.py:117-203), drop_rate == 0
# --------------------------------------------------------------------------- #
def _att(p, tag):
    return {name: p[f'{tag}_{name}'] for name in MHA_PARAM_NAMES}


def decoder_fwd(p: Dict[str, Array], q: Array, kv: Array, norm_first: bool, eps: float = 1e-3,
                verbatim: bool = False, drop=None):
    """Self-attention, cross-attention over kv, feed-forward (transformer.py:117-158).
    Params: sa_*, ca_* (attention), n1/n2/n3_{gamma,beta}, d1_w/b, d2_w/b.  ``drop = (mask1, mask2, mask3, keep_prob)``: the
    masks the three DropOuts drew (each directly in front of its norm, transformer.py:125,131,137,142,149,154)."""
    b, s, f = q.shape
    c: Dict[str, object] = {'drop': drop}
    dr = [(lambda t, m=m: dropout_apply(t, m.reshape(t.shape), drop[3])) for m in drop[:3]] if drop is not None else [lambda t: t] * 3
    skip = q
    h = q
    if norm_first:
        h = dr[0](h)
        c['n1_x'] = h
        h, c['n1'] = layernorm_fwd(h, p['n1_gamma'], p['n1_beta'], eps)
    out, c['sa'] = mha_fwd(_att(p, 'sa'), h, verbatim=verbatim)
    out = out + skip
    if not norm_first:
        out = dr[0](out)
        c['n1_x'] = out
        out, c['n1'] = layernorm_fwd(out, p['n1_gamma'], p['n1_beta'], eps)
    skip = out
    if norm_first:
        out = dr[1](out)
        c['n2_x'] = out
        out, c['n2'] = layernorm_fwd(out, p['n2_gamma'], p['n2_beta'], eps)
    out, c['ca'] = mha_fwd(_att(p, 'ca'), out, kv, verbatim=verbatim)
    out = out + skip
    if not norm_first:
        out = dr[1](out)
        c['n2_x'] = out
        out, c['n2'] = layernorm_fwd(out, p['n2_gamma'], p['n2_beta'], eps)
    out = out.reshape(-1, f)
    skip = out
    if norm_first:
        out = dr[2](out)
        c['n3_x'] = out
        out, c['n3'] = layernorm_fwd(out, p['n3_gamma'], p['n3_beta'], eps)
    c['d1_x'] = out
    out, c['d1_pre'] = dense_fwd(out, p['d1_w'], p['d1_b'])
    c['d2_x'] = out
    out = linear_fwd(out, p['d2_w'], p['d2_b']) + skip
    if not norm_first:
        out = dr[2](out)
        c['n3_x'] = out
        out, c['n3'] = layernorm_fwd(out, p['n3_gamma'], p['n3_beta'], eps)
    return out.reshape(b, s, f), c


def decoder_bwd(p, c, dy: Array, norm_first: bool, eps: float = 1e-3, verbatim: bool = False):
    """Returns ((dq, dkv), grads); dkv = dkey + dvalue of the cross-attention (transformer.py:160-203)."""
    b, s, f = dy.shape
    g: Dict[str, Array] = {}
    drop = c.get('drop')
    dr = [(lambda t, m=m: dropout_apply(t, m.reshape(t.shape), drop[3])) for m in drop[:3]] if drop is not None else [lambda t: t] * 3
    dy = dy.reshape(-1, f)
    if not norm_first:
        dy, g['n3_gamma'], g['n3_beta'] = layernorm_bwd(c['n3_x'], p['n3_gamma'], eps, c['n3'], dy, verbatim)
        dy = dr[2](dy)
    dskip = dy
    dy, g['d2_w'], g['d2_b'] = linear_bwd(c['d2_x'], p['d2_w'], dy)
    dy, g['d1_w'], g['d1_b'] = dense_bwd(c['d1_x'], p['d1_w'], c['d1_pre'], dy)
    if norm_first:
        dy, g['n3_gamma'], g['n3_beta'] = layernorm_bwd(c['n3_x'], p['n3_gamma'], eps, c['n3'], dy, verbatim)
        dy = dr[2](dy)
    dy = (dy + dskip).reshape(b, s, f)
    if not norm_first:
        dy, g['n2_gamma'], g['n2_beta'] = layernorm_bwd(c['n2_x'], p['n2_gamma'], eps, c['n2'], dy, verbatim)
        dy = dr[1](dy)
    dskip = dy
    (dq, dk, dv), ga = mha_bwd(_att(p, 'ca'), c['ca'], dy, verbatim=verbatim)
    for name, arr in ga.items():
        g[f'ca_{name}'] = arr
    dkv = dk + dv
    dy = dq
    if norm_first:
        dy, g['n2_gamma'], g['n2_beta'] = layernorm_bwd(c['n2_x'], p['n2_gamma'], eps, c['n2'], dy, verbatim)
        dy = dr[1](dy)
    dy = dy + dskip
    if not norm_first:
        dy, g['n1_gamma'], g['n1_beta'] = layernorm_bwd(c['n1_x'], p['n1_gamma'], eps, c['n1'], dy, verbatim)
        dy = dr[0](dy)
    dskip = dy
    (dq, dk, dv), ga = mha_bwd(_att(p, 'sa'), c['sa'], dy, verbatim=verbatim)
    for name, arr in ga.items():
        g[f'sa_{name}'] = arr
    dy = dq + dk + dv
    if norm_first:
        dy, g['n1_gamma'], g['n1_beta'] = layernorm_bwd(c['n1_x'], p['n1_gamma'], eps, c['n1'], dy, verbatim)
        dy = dr[0](dy)
    return (dy + dskip, dkv), g


# --------------------------------------------------------------------------- #
# optimizers and losses (reference optimizer.py:26-69, loss.py:20-39)
# --------------------------------------------------------------------------- #
# --------------------------------------------------------------------------- #
# DropOut (reference layers/normalizations.py:14-30) and the device-side mask generator
# --------------------------------------------------------------------------- #
def dropout_apply(x: Array, mask: Array, keep_prob: float) -> Array:
    """np.where(mask, x / keep_prob, 0) (normalizations.py:21-23, and :27-30 for the gradient)."""
    return np.where(mask, x / keep_prob, 0.0)


def philox4x32_10(counter: Array, key) -> Array:
    """Philox4x32-10 of Salmon, Moraes, Dror, Shaw, "Parallel random numbers: as easy as 1, 2, 3" (SC'11), the
    generator csrc/npm_optim.hip draws dropout masks with.  ``counter`` [..., 4] uint32, ``key`` (k0, k1); returns
    [..., 4] uint32.  Pinned by the published known-answer vectors in tests/test_oracle_golden.py."""
    c = np.array(counter, dtype=np.uint64) & 0xFFFFFFFF
    k0, k1 = np.uint64(int(key[0]) & 0xFFFFFFFF), np.uint64(int(key[1]) & 0xFFFFFFFF)
    mask32 = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c[..., 0]
        p1 = np.uint64(0xCD9E8D57) * c[..., 2]
        n0 = (p1 >> np.uint64(32)) ^ c[..., 1] ^ k0
        n2 = (p0 >> np.uint64(32)) ^ c[..., 3] ^ k1
        c = np.stack([n0 & mask32, p1 & mask32, n2 & mask32, p0 & mask32], axis=-1)
        k0 = (k0 + np.uint64(0x9E3779B9)) & mask32
        k1 = (k1 + np.uint64(0xBB67AE85)) & mask32
    return c.astype(np.uint32)


def dropout_philox_mask(n: int, keep_prob: float, seed: int, offset: int) -> Array:
    """The byte mask npm_dropout_philox draws for n elements: element i keeps its value when word i & 3 of
    Philox4x32-10(counter = (i // 4 lo, i // 4 hi, offset lo, offset hi), key = (seed lo, seed hi)) is below
    keep_prob * 2^32 (keep_prob as fp32, the threshold formed in fp64 and truncated)."""
    groups = (n + 3) // 4
    g = np.arange(groups, dtype=np.uint64)
    counter = np.stack([g & np.uint64(0xFFFFFFFF), g >> np.uint64(32),
                        np.full(groups, offset & 0xFFFFFFFF, dtype=np.uint64),
                        np.full(groups, (offset >> 32) & 0xFFFFFFFF, dtype=np.uint64)], axis=-1)
    words = philox4x32_10(counter, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)).reshape(-1)[:n]
    scaled = float(np.float32(keep_prob)) * 4294967296.0
    threshold = 4294967296 if scaled >= 4294967296.0 else int(scaled)
    return words.astype(np.uint64) < np.uint64(threshold)


def f16x2_scales(x: Array, axis: int):
    """Power-of-two scales of the product's scaled fp16 split (np_modeling_amd/csrc/npm_gemm_f16x2.hip scales_kernel):
    2^(14 - e) where the largest magnitude along `axis` is f 2^e with f in [0.5, 1); 1 for an all-zero slice.  Not a
    restatement of the reference (which has no such mode): the arithmetic of the product's OWN opt-in math mode,
    spelled out so that tests can hold the kernel to it."""
    m = np.abs(np.asarray(x, dtype=np.float32)).max(axis=axis, keepdims=True)
    _, e = np.frexp(m)
    return np.where(m > 0, np.ldexp(np.float32(1), np.minimum(14 - e, 126)), np.float32(1)).astype(np.float32)


def gemm_f16x2(a: Array, b: Array) -> Array:
    """C = A B as NPM_MATH_F16X2 forms it: rows of A and columns of B scaled by powers of two, x s = hi + lo with hi =
    fp16(x s) rounded to nearest and lo = fp16(x s - hi), C = (hi hi + hi lo + lo hi) / (s_a s_b).  The three products
    are accumulated in fp64 here (the kernel accumulates them in fp32 on the matrix pipe)."""
    a, b = np.asarray(a, dtype=np.float32), np.asarray(b, dtype=np.float32)
    sa, sb = f16x2_scales(a, 1), f16x2_scales(b, 0)
    xa, xb = a * sa, b * sb                                   # exact: powers of two
    ha, hb = xa.astype(np.float16), xb.astype(np.float16)
    la, lb = (xa - ha.astype(np.float32)).astype(np.float16), (xb - hb.astype(np.float32)).astype(np.float16)
    ha, hb, la, lb = (t.astype(np.float64) for t in (ha, hb, la, lb))
    return (ha @ hb + (ha @ lb + la @ hb)) / (sa.astype(np.float64) * sb.astype(np.float64))


def sgd_step(param: Array, grad: Array, lr: float) -> Array:
    """v -= lr * g, in the parameter's own dtype (optimizer.py:32)."""
    out = param.copy()
    out -= (lr * grad).astype(param.dtype, copy=False)
    return out


def adam_step(param, grad, state: dict, lr, beta1=0.9, beta2=0.999, eps=1e-7):
    """eps INSIDE the sqrt, bias-corrected, fp64 moments (optimizer.py:53-67)."""
    t = state.get('t', 1)
    m = state.get('m', np.zeros(grad.shape))
    v = state.get('v', np.zeros(grad.shape))
    m = beta1 * m + (1 - beta1) * grad
    v = beta2 * v + (1 - beta2) * grad ** 2
    m_hat = m / (1 - beta1 ** t)
    v_hat = v / (1 - beta2 ** t)
    out = param.copy()
    out -= (lr * (m_hat / np.sqrt(v_hat + eps))).astype(param.dtype)
    state.update(t=t + 1, m=m, v=v)
    return out


def xent_fwd(y, targets):
    """-sum(t log y) (loss.py:33-36)."""
    return -np.sum(targets * np.log(y))


def xent_bwd(y, targets):
    """-t / y (loss.py:38-39)."""
    return -targets / y


def mse_fwd(y, targets):
    """sum(diff^2) / size (loss.py:21-25)."""
    diff = y - targets
    return np.sum(diff ** 2) / y.size


def mse_bwd(y, targets):
    """2 * diff / size (loss.py:27-29)."""
    return 2 * (y - targets) / y.size
