"""GPU: HBM-bound kernels (elementwise, column sum, softmax, LayerNorm) against the oracle."""

import ctypes as C

import numpy as np
import pytest

from conftest import assert_close, load_golden
from oracle import np_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def env():
    from np_modeling_amd import _C, device
    return _C, device


@pytest.mark.parametrize('n', [0, 1, 3, 4, 1023, 4096, 1 << 20, (1 << 20) + 5])
def test_elementwise(env, n):
    _C, D = env
    lib = _C.lib()
    rng = np.random.default_rng(n)
    a = rng.standard_normal(n).astype(np.float32)
    b = rng.standard_normal(n).astype(np.float32)
    c = rng.standard_normal(n).astype(np.float32)
    if n > 2:
        a[:2] = 0.0
    da, db, dc, out = D.from_host(a), D.from_host(b), D.from_host(c), D.empty([n])
    _C.check(lib.npm_relu_fwd(da.ptr, out.ptr, n))
    np.testing.assert_array_equal(out.numpy(), O.relu_fwd(a))
    _C.check(lib.npm_relu_bwd(da.ptr, db.ptr, out.ptr, n))
    np.testing.assert_array_equal(out.numpy(), O.relu_bwd(a, b).astype(np.float32))
    _C.check(lib.npm_add(da.ptr, db.ptr, out.ptr, n))
    np.testing.assert_array_equal(out.numpy(), a + b)
    _C.check(lib.npm_add3(da.ptr, db.ptr, dc.ptr, out.ptr, n))
    np.testing.assert_array_equal(out.numpy(), (a + b) + c)
    y = D.from_host(a)
    _C.check(lib.npm_axpy(y.ptr, db.ptr, -0.25, n))
    np.testing.assert_allclose(y.numpy(), a - np.float32(0.25) * b, rtol=1e-6, atol=1e-7)
    _C.check(lib.npm_scale(da.ptr, out.ptr, 3.0, n))
    np.testing.assert_array_equal(out.numpy(), a * np.float32(3.0))


def test_elementwise_unaligned_views(env):
    _C, D = env
    base = D.from_host(np.arange(1000, dtype=np.float32))
    view = base.flat_view(1, [997])                # 4-byte offset: scalar path
    out = D.empty([1001]).flat_view(3, [997])
    _C.check(_C.lib().npm_scale(view.ptr, out.ptr, 2.0, 997))
    np.testing.assert_array_equal(out.numpy(), 2 * np.arange(1, 998, dtype=np.float32))


@pytest.mark.parametrize('rows,cols', [(0, 6), (1, 1), (7, 5), (1000, 64), (4099, 130), (20000, 1024), (3, 4096),
                                       (40008, 128), (40009, 128), (1 << 20, 4), (70000, 64)])   # >= 4 M elements, cols | 1024: whole-line kernel
def test_colsum(env, rows, cols):
    _C, D = env
    x = np.random.default_rng(rows + cols).standard_normal((rows, cols)).astype(np.float32)
    out = D.colsum(D.from_host(x), rows, cols)
    assert_close(out, x.astype(np.float64).sum(axis=0), tol=2e-6)


@pytest.mark.parametrize('rows,cols', [(0, 8), (1, 1), (7, 5), (1000, 64), (4099, 130), (50000, 128), (3, 4096),
                                       (40001, 128), (70000, 64), (66000, 256)])               # incl. odd line counts and the strip fallback
def test_relu_bwd_colsum(env, rows, cols):
    """ReLU backward + bias gradient in one pass (activations.py:19 then conv.py:55 / mlp.py:34)."""
    _C, D = env
    rng = np.random.default_rng(rows * 7 + cols)
    pre = rng.standard_normal((rows, cols)).astype(np.float32)
    dy = rng.standard_normal((rows, cols)).astype(np.float32)
    if rows:
        pre[0, : min(cols, 3)] = 0.0                 # x == 0 passes the gradient (>=)
        if cols > 1:
            pre[-1, 1] = -0.0
    out = D.from_host(np.full(cols, 9.0, dtype=np.float32))
    g = D.relu_bwd_colsum(D.from_host(pre), D.from_host(dy), cols, out)
    want = O.relu_bwd(pre, dy).astype(np.float32)
    np.testing.assert_array_equal(g.numpy(), want)
    assert_close(out, want.astype(np.float64).sum(axis=0), tol=2e-6)


def test_relu_bwd_colsum_unaligned(env):
    _C, D = env
    rows, cols = 37, 12
    rng = np.random.default_rng(5)
    pre = rng.standard_normal((rows, cols)).astype(np.float32)
    dy = rng.standard_normal((rows, cols)).astype(np.float32)
    base_pre, base_dy = D.empty([rows * cols + 1]), D.empty([rows * cols + 1])
    vp, vd = base_pre.flat_view(1, [rows, cols]), base_dy.flat_view(1, [rows, cols])     # 4-byte offset
    vp.set(pre)
    vd.set(dy)
    guard = D.from_host(np.full(rows * cols + 2, 7.0, dtype=np.float32))
    g = guard.flat_view(1, [rows, cols])
    out = D.empty([cols])
    _C.check(_C.lib().npm_relu_bwd_colsum(vp.ptr, vd.ptr, g.ptr, out.ptr, rows, cols))
    want = np.where(pre >= 0, dy, 0).astype(np.float32)
    host = guard.numpy()
    np.testing.assert_array_equal(host[1:-1].reshape(rows, cols), want)
    assert host[0] == 7.0 and host[-1] == 7.0
    assert_close(out, want.astype(np.float64).sum(axis=0), tol=2e-6)


@pytest.mark.parametrize('rows,n', [(1, 1), (5, 3), (128, 128), (33, 40), (64, 512), (10, 1000), (7, 4096),
                                    (3, 5000), (9, 30)])
@pytest.mark.parametrize('scale', [1.0, 0.125])
def test_softmax(env, rows, n, scale):
    _C, D = env
    rng = np.random.default_rng(rows * 31 + n)
    x = (rng.standard_normal((rows, n)) * 4).astype(np.float32)
    dy = rng.standard_normal((rows, n)).astype(np.float32)
    y, dxin = D.empty([rows, n]), D.from_host(x)
    _C.check(_C.lib().npm_softmax_fwd(dxin.ptr, y.ptr, rows, n, scale))
    want = O.softmax_fwd(np.float64(scale) * x.astype(np.float64))
    assert_close(y, want, tol=2e-6)
    np.testing.assert_allclose(y.numpy().sum(axis=-1), 1.0, rtol=1e-5)
    dx, ddy = D.empty([rows, n]), D.from_host(dy)
    _C.check(_C.lib().npm_softmax_bwd(y.ptr, ddy.ptr, dx.ptr, rows, n, scale))
    want_dx = scale * O.softmax_bwd(y.numpy(), dy)
    assert_close(dx, want_dx, tol=5e-6)


def test_softmax_golden_and_in_place(env):
    _C, D = env
    g = load_golden('softmax')
    x = D.from_host(g['x'])
    rows, n = g['x'].size // g['x'].shape[-1], g['x'].shape[-1]
    _C.check(_C.lib().npm_softmax_fwd(x.ptr, x.ptr, rows, n, 1.0))       # in place
    assert_close(x, g['y'], tol=2e-6)
    dy = D.from_host(g['dy'])
    _C.check(_C.lib().npm_softmax_bwd(x.ptr, dy.ptr, dy.ptr, rows, n, 1.0))
    assert_close(dy, g['dx'], tol=5e-6)


def test_softmax_extreme_logits(env):
    _C, D = env
    x = np.array([[1e4, 1e4 - 1, -1e4, 0.0] * 4, [-1e30] * 16], dtype=np.float32)
    y, dxin = D.empty(x.shape), D.from_host(x)
    _C.check(_C.lib().npm_softmax_fwd(dxin.ptr, y.ptr, 2, 16, 1.0))
    assert np.all(np.isfinite(y.numpy()))
    assert_close(y, O.softmax_fwd(x.astype(np.float64)), tol=2e-6)


@pytest.mark.parametrize('rows,d', [(1, 4), (32, 128), (30, 72), (257, 1024), (5, 4096), (17, 100), (6, 33),
                                    (3, 5000), (5000, 256)])
@pytest.mark.parametrize('with_residual', [False, True])
def test_layernorm(env, rows, d, with_residual):
    _C, D = env
    lib = _C.lib()
    rng = np.random.default_rng(rows * 7 + d)
    x = (rng.standard_normal((rows, d)) * 2 + 0.5).astype(np.float32)
    gamma = rng.standard_normal(d).astype(np.float32)
    beta = rng.standard_normal(d).astype(np.float32)
    dz = rng.standard_normal((rows, d)).astype(np.float32)
    res = rng.standard_normal((rows, d)).astype(np.float32)
    eps = 1e-3
    dx_, dg_, dbt_ = D.from_host(x), D.from_host(gamma), D.from_host(beta)
    z, mean, rstd = D.empty([rows, d]), D.empty([rows]), D.empty([rows])
    _C.check(lib.npm_layernorm_fwd(dx_.ptr, dg_.ptr, dbt_.ptr, eps, rows, d, z.ptr, mean.ptr, rstd.ptr))
    x64 = x.astype(np.float64)
    want_z, cache = O.layernorm_fwd(x64, gamma.astype(np.float64), beta.astype(np.float64), eps)
    assert_close(z, want_z, tol=3e-6)
    assert_close(mean, cache[0][:, 0], tol=2e-6)
    assert_close(rstd, 1 / np.sqrt(cache[1][:, 0] + eps), tol=2e-6)
    dx, dgamma, dbeta = D.empty([rows, d]), D.empty([d]), D.empty([d])
    dres, ddz = D.from_host(res), D.from_host(dz)      # keep operands alive: raw pointers do not
    _C.check(lib.npm_layernorm_bwd(ddz.ptr, dx_.ptr, mean.ptr, rstd.ptr, dg_.ptr,
                                   dres.ptr if with_residual else None, rows, d, dx.ptr, dgamma.ptr, dbeta.ptr))
    want_dx, want_dg, want_db = O.layernorm_bwd(x64, gamma.astype(np.float64), eps, cache, dz.astype(np.float64))
    if with_residual:
        want_dx = want_dx + res
    assert_close(dx, want_dx, tol=5e-6)
    assert_close(dgamma, want_dg, tol=5e-6)
    assert_close(dbeta, want_db, tol=5e-6)
