"""Recipes of the fixtures at the reference's OWN test shapes (tests/golden/ref_*.npz).

Shared by ``oracle/make_golden.py`` (which runs the real reference over them, build container only) and by the
tests (which rebuild the same inputs and parameters and compare with the stored reference outputs).  Nothing here
imports the reference or the product.

Why these cases (round-3 verdict): the reference's tests run MultiHeadAttention / TransformerEncoder / TransformerDecoder
at B16, S32, F128, H8 -- head size 16 (layers/attentions_test.py:13-85, layers/transformer_test.py:98-156,159-219, kv
[16, 128, 128]) and Conv2D at x [64, 32, 16, 32] -> 16, k 3 (layers/conv_test.py:37-107).  Head sizes 16 / 32 / 64 / 128
are the ones the fused attention kernels take (include/npm_hip.h), so these fixtures compare ``mha_fwd_kernel`` /
``mha_bwd16_kernel`` / ``mha_bwd_kernel`` with reference arrays directly.

Flow of every case, as the reference's tests do it: ``np.random.seed(0)``; inputs then targets with ``utils.rand``
(layers/utils.py:17-18); the layer is initialised by a first call (global-RNG draws, discarded); its parameters are then
REBOUND the way the Flax binders do (layers/utils.py:27-59,62-68,77-88: arrays assigned to the private attributes, layer-norm
epsilon 1e-6) to draws scaled by 1/sqrt(fan_in) -- Flax's default ``lecun_normal`` scale -- so that the softmaxes are not
one-hot; forward; ``dy`` = gradient of ``utils.mse_loss`` (layers/utils.py:21-24) at (float32 output, targets); backward
with SGD.  Conv2D keeps its own initialisation (conv_test.py does not rebind).

Fixtures store the reference's OUTPUTS (float32) and CRC-32s of the regenerated inputs: NumPy's legacy seeded stream is
frozen, so inputs and parameters are rebuilt instead of stored.
"""

import zlib

import numpy as np

MHA_NAMES = ('wq', 'wk', 'wv', 'wo', 'bq', 'bk', 'bv', 'bo')

CASES = {
    # layers/attentions_test.py:13-85 (self-attention; the test's seq_len_kv is unused there)
    'ref_mha_self_d16': dict(kind='mha', batch=16, sq=32, skv=None, feat=128, heads=8, lr=0.01, updated=True),
    'ref_mha_self_d32': dict(kind='mha', batch=2, sq=48, skv=None, feat=128, heads=4, lr=0.01),
    'ref_mha_cross_d64': dict(kind='mha', batch=3, sq=24, skv=72, feat=128, heads=2, lr=0.01),
    # head size 128 = the BASELINE configs' (C4 / C5: d_model 1024, 8 heads); 136 positions: ragged against the
    # kernels' 32-query tiles and one step past their 128-key blocks
    'ref_mha_self_d128': dict(kind='mha', batch=1, sq=136, skv=None, feat=256, heads=2, lr=0.01),
    # layers/transformer_test.py:98-156
    'ref_encoder_prenorm': dict(kind='encoder', batch=16, sq=32, feat=128, heads=8, hidden=256, norm_first=True, lr=1e-3),
    'ref_encoder_postnorm': dict(kind='encoder', batch=16, sq=32, feat=128, heads=8, hidden=256, norm_first=False, lr=1e-3),
    # layers/transformer_test.py:159-219
    'ref_decoder_prenorm': dict(kind='decoder', batch=16, sq=32, skv=128, feat=128, heads=8, hidden=256, norm_first=True, lr=1e-3),
    'ref_decoder_postnorm': dict(kind='decoder', batch=16, sq=32, skv=128, feat=128, heads=8, hidden=256, norm_first=False, lr=1e-3),
    # layers/conv_test.py:37-107
    'ref_conv_k3': dict(kind='conv', shape=[64, 32, 16, 32], channels=16, k=3, lr=0.01),
    # layers/mlp_test.py:35-94 (its own initialisation), activations_test.py:11-32, normalizations_test.py:37-90 (gamma 1,
    # beta 0, epsilon 1e-6 bound in from flax.linen.LayerNorm): the reference tests' own flow (dy = gradient of the MSE)
    'ref_dense': dict(kind='dense', shape=[64, 32], units=16, lr=0.01),
    'ref_softmax': dict(kind='softmax', shape=[128, 128]),
    'ref_layernorm': dict(kind='layernorm', shape=[32, 128], lr=1e-3),
}

LN_EPS = 1e-6                    # flax.linen.LayerNorm's default, bound in by layers/utils.py:62-68
CONV_DX_SAMPLES = (0, 9, 18, 27, 36, 45, 54, 63)


def rand(shape):
    """layers/utils.py:17-18."""
    return np.random.normal(size=shape).astype(np.float32)


def crc(a) -> int:
    return zlib.crc32(np.ascontiguousarray(a).tobytes())


def draw_inputs(case):
    """After ``np.random.seed(0)``: the inputs, then the targets.  Returns a dict of float32 arrays."""
    np.random.seed(0)
    c = case
    if c['kind'] == 'conv':
        x = rand(c['shape'])
        return dict(x=x, targets=rand(c['shape'][:3] + [c['channels']]))
    if c['kind'] == 'dense':
        x = rand(c['shape'])
        return dict(x=x, targets=rand([c['shape'][0], c['units']]))
    if c['kind'] in ('softmax', 'layernorm'):
        x = rand(c['shape'])
        return dict(x=x, targets=rand(c['shape']))
    out = dict(query=rand([c['batch'], c['sq'], c['feat']]))
    if c.get('skv'):
        out['kv'] = rand([c['batch'], c['skv'], c['feat']])
    out['targets'] = rand([c['batch'], c['sq'], c['feat']])
    return out


def _mha_params(rng, feat, heads):
    d = feat // heads
    s = 1.0 / np.sqrt(feat)          # fan_in of every projection is H * D = F

    def w(shape):
        return (rng.standard_normal(shape) * s).astype(np.float32)

    def b(shape):
        return (rng.standard_normal(shape) * 0.1).astype(np.float32)

    return dict(wq=w([heads, d, feat]), wk=w([heads, d, feat]), wv=w([heads, d, feat]), wo=w([feat, heads, d]),
                bq=b([heads, d]), bk=b([heads, d]), bv=b([heads, d]), bo=b([feat]))


def bound_params(case):
    """The parameters the layer is rebound to (reference attribute layouts), from a private seeded stream.
    Keys: MHA -> wq..bo; encoder -> att_*, n1/n2_{gamma,beta}, d1_w/b, d2_w/b; decoder -> sa_*, ca_*, n1..n3, d1, d2."""
    c = case
    rng = np.random.RandomState(20260104)
    if c['kind'] in ('conv', 'dense', 'softmax'):
        return None
    if c['kind'] == 'layernorm':
        return dict(gamma=np.ones(c['shape'][-1:], dtype=np.float32), beta=np.zeros(c['shape'][-1:], dtype=np.float32))
    feat, heads = c['feat'], c['heads']
    if c['kind'] == 'mha':
        return _mha_params(rng, feat, heads)
    p = {}
    tags = ('att',) if c['kind'] == 'encoder' else ('sa', 'ca')
    for tag in tags:
        for k, v in _mha_params(rng, feat, heads).items():
            p[f'{tag}_{k}'] = v
    for i in range(1, 3 if c['kind'] == 'encoder' else 4):
        p[f'n{i}_gamma'] = (1.0 + 0.1 * rng.standard_normal([feat])).astype(np.float32)
        p[f'n{i}_beta'] = (0.1 * rng.standard_normal([feat])).astype(np.float32)
    u = c['hidden']
    p['d1_w'] = (rng.standard_normal([feat, u]) / np.sqrt(feat)).astype(np.float32)
    p['d1_b'] = (0.1 * rng.standard_normal([u])).astype(np.float32)
    p['d2_w'] = (rng.standard_normal([u, feat]) / np.sqrt(u)).astype(np.float32)
    p['d2_b'] = (0.1 * rng.standard_normal([feat])).astype(np.float32)
    return p


def mse_grad(out, targets):
    """``jax.grad(utils.mse_loss)(output, targets)`` (layers/utils.py:21-24) in float32, at the float32 output."""
    out = np.asarray(out, dtype=np.float32)
    return ((out - targets) * np.float32(2.0 / out.size)).astype(np.float32)


# where each named parameter lives in a composite layer: name -> (path of sub-layer attributes, attribute)
def param_sites(case):
    if case['kind'] == 'mha':
        return {n: ('', '_' + n) for n in MHA_NAMES}
    sites = {}
    att = {'att': '_self_attention'} if case['kind'] == 'encoder' else {'sa': '_self_attention', 'ca': '_cross_attention'}
    for tag, attr in att.items():
        for n in MHA_NAMES:
            sites[f'{tag}_{n}'] = (attr, '_' + n)
    for i in range(1, 3 if case['kind'] == 'encoder' else 4):
        sites[f'n{i}_gamma'] = (f'_norm{i}', '_gamma')
        sites[f'n{i}_beta'] = (f'_norm{i}', '_beta')
    sites.update(d1_w=('_dense1._linear', '_w'), d1_b=('_dense1._linear', '_b'), d2_w=('_dense2', '_w'), d2_b=('_dense2', '_b'))
    return sites


def sub(obj, path):
    for part in filter(None, path.split('.')):
        obj = getattr(obj, part)
    return obj


def bind(layer, case, params):
    """Assign ``params`` into the layer's private attributes like layers/utils.py:41-101 and set the norms' epsilon."""
    for name, (path, attr) in param_sites(case).items():
        target = sub(layer, path)
        assert getattr(target, attr).shape == params[name].shape, name
        setattr(target, attr, params[name].copy())
    if case['kind'] != 'mha':
        for i in range(1, 3 if case['kind'] == 'encoder' else 4):
            getattr(layer, f'_norm{i}')._epsilon = LN_EPS


class GradRecorder:
    """An ``optimizer_`` that records the gradients a backward hands to ``update(obj, attribute, gradient)``
    (optimizer.py:13-18's signature) instead of applying them."""

    def __init__(self):
        self.grads = {}

    def update(self, obj, attribute, gradient):
        self.grads[(id(obj), attribute)] = np.array(gradient, dtype=np.float64, copy=True)

    def named(self, layer, case):
        return {name: self.grads[(id(sub(layer, path)), attr)] for name, (path, attr) in param_sites(case).items()}
