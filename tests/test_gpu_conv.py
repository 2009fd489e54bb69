"""GPU: Conv2D (implicit-im2col MFMA GEMM) against the reference's outputs and the oracle."""

import ctypes as C

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


@pytest.mark.parametrize('name,shape,channels,k', [('conv_k3', [3, 9, 7, 8], 12, 3), ('conv_k5', [2, 8, 8, 4], 6, 5),
                                                   ('conv_k1', [2, 6, 5, 8], 4, 1)])
def test_conv_golden(npm, name, shape, channels, k, math_mode):
    """Flow of reference layers/conv_test.py:37-107 at fixture size, seeded like make_golden.py."""
    g = load_golden(name)
    np.random.seed(0)
    layer = npm.layers.Conv2D(channels=channels, kernel_size=k)
    x = rand(shape)
    y = layer(x)
    w, b = layer.w, layer.b
    np.testing.assert_array_equal(np.asarray(w), g['w0'])
    np.testing.assert_array_equal(np.asarray(b), g['b0'])
    assert_close(y, g['y'], tol=3e-6)
    dx = layer(g['dy'], backprop=True, learning_rate=float(g['lr']))
    assert_close(dx, g['dx'], tol=3e-6)
    assert_close(w, g['w1'], tol=3e-6)          # aliases observe the in-place update (conv_test.py:57-58)
    assert_close(b, g['b1'], tol=3e-6)


@pytest.mark.parametrize('n,h,w,c0,c1,k', [(1, 1, 1, 1, 1, 1), (2, 5, 4, 3, 5, 3), (1, 7, 9, 6, 10, 5),
                                           (64, 32, 16, 32, 16, 3),       # reference conv_test.py shape
                                           (2, 16, 16, 64, 128, 3),       # C3 channel counts, small image
                                           (3, 6, 6, 20, 132, 1), (1, 3, 40, 8, 8, 7),
                                           (2, 20, 12, 64, 128, 3),       # grad_x takes the tall 256x64 tile (N = 64)
                                           (1, 9, 9, 16, 48, 5), (3, 7, 5, 32, 16, 3)])
def test_conv_kernels_vs_oracle(npm, n, h, w, c0, c1, k, math_mode):
    from np_modeling_amd import _C, device as D
    lib = _C.lib()
    rng = np.random.default_rng(n * 131 + h * 17 + c0 + k)
    x = rng.standard_normal((n, h, w, c0)).astype(np.float32)
    filt = rng.standard_normal((k, k, c0, c1)).astype(np.float32)
    bias = rng.standard_normal(c1).astype(np.float32)
    dy = rng.standard_normal((n, h, w, c1)).astype(np.float32)
    dx_, df, db, ddy = D.from_host(x), D.from_host(filt), D.from_host(bias), D.from_host(dy)
    y, pre = D.empty([n, h, w, c1]), D.empty([n, h, w, c1])
    desc = _C.npm_conv2d(n=n, h=h, w=w, c_in=c0, c_out=c1, ksize=k, x=dx_.ptr, filt=df.ptr, bias=db.ptr,
                         y=y.ptr, pre=pre.ptr, relu=1)
    _C.check(lib.npm_conv2d_fwd(C.byref(desc)))
    want_y, want_pre = O.conv_layer_fwd(x.astype(np.float64), filt.astype(np.float64), bias.astype(np.float64))
    assert_close(pre, want_pre, tol=3e-6)
    np.testing.assert_array_equal(y.numpy(), np.maximum(pre.numpy(), 0))
    # relu without the saved pre-activation, no bias
    y2 = D.empty([n, h, w, c1])
    desc2 = _C.npm_conv2d(n=n, h=h, w=w, c_in=c0, c_out=c1, ksize=k, x=dx_.ptr, filt=df.ptr, bias=None,
                          y=y2.ptr, pre=None, relu=1)
    _C.check(lib.npm_conv2d_fwd(C.byref(desc2)))
    assert_close(y2, np.maximum(O.conv2d_fwd(x.astype(np.float64), filt.astype(np.float64)), 0), tol=3e-6)
    gx, gw = D.empty([n, h, w, c0]), D.empty([k, k, c0, c1])
    _C.check(lib.npm_conv2d_bwd_x(ddy.ptr, df.ptr, gx.ptr, n, h, w, c0, c1, k))
    _C.check(lib.npm_conv2d_bwd_w(ddy.ptr, dx_.ptr, gw.ptr, n, h, w, c0, c1, k))
    assert_close(gx, O.conv2d_grad_x(dy.astype(np.float64), filt.astype(np.float64)), tol=3e-6)
    assert_close(gw, O.conv2d_grad_w(dy.astype(np.float64), x.astype(np.float64), k), tol=5e-6)


def test_conv_random_sweep(npm, math_mode):
    """Randomised Conv2D geometries (odd kernels 1..7, channel counts on and off the DMA path's multiples of 16 / 4,
    images smaller than the kernel's reach) through forward, input gradient and filter gradient."""
    from np_modeling_amd import _C, device as D
    lib = _C.lib()
    rng = np.random.default_rng(77)
    for case in range(24):
        n, h, w = int(rng.integers(1, 5)), int(rng.integers(1, 20)), int(rng.integers(1, 20))
        c0 = int(rng.choice([1, 3, 4, 8, 16, 20, 32, 48, 64]))
        c1 = int(rng.choice([1, 2, 4, 12, 16, 32, 64, 80, 128, 144]))
        k = int(rng.choice([1, 3, 3, 5, 7]))
        x = rng.standard_normal((n, h, w, c0)).astype(np.float32)
        filt = rng.standard_normal((k, k, c0, c1)).astype(np.float32)
        bias = rng.standard_normal(c1).astype(np.float32)
        dy = rng.standard_normal((n, h, w, c1)).astype(np.float32)
        dx_, df, db, ddy = D.from_host(x), D.from_host(filt), D.from_host(bias), D.from_host(dy)
        y, pre = D.empty([n, h, w, c1]), D.empty([n, h, w, c1])
        desc = _C.npm_conv2d(n=n, h=h, w=w, c_in=c0, c_out=c1, ksize=k, x=dx_.ptr, filt=df.ptr, bias=db.ptr,
                             y=y.ptr, pre=pre.ptr, relu=1)
        _C.check(lib.npm_conv2d_fwd(C.byref(desc)))
        what = f'case {case}: n={n} h={h} w={w} c0={c0} c1={c1} k={k} math={math_mode}'
        x64, f64, dy64 = x.astype(np.float64), filt.astype(np.float64), dy.astype(np.float64)
        _, want_pre = O.conv_layer_fwd(x64, f64, bias.astype(np.float64))
        assert_close(pre, want_pre, tol=3e-6, what=what + ' fwd')
        np.testing.assert_array_equal(y.numpy(), np.maximum(pre.numpy(), 0))
        gx, gw = D.empty([n, h, w, c0]), D.empty([k, k, c0, c1])
        _C.check(lib.npm_conv2d_bwd_x(ddy.ptr, df.ptr, gx.ptr, n, h, w, c0, c1, k))
        _C.check(lib.npm_conv2d_bwd_w(ddy.ptr, dx_.ptr, gw.ptr, n, h, w, c0, c1, k))
        assert_close(gx, O.conv2d_grad_x(dy64, f64), tol=3e-6, what=what + ' grad_x')
        assert_close(gw, O.conv2d_grad_w(dy64, x64, k), tol=5e-6, what=what + ' grad_w')


@pytest.mark.parametrize('fused', [1, 2, 3, 0])
@pytest.mark.parametrize('n,h,w,c0,c1,k', [
    (2, 16, 16, 64, 128, 3),       # C3 channel counts: M = 576 = 3 x 192 rows
    (1, 32, 48, 4, 20, 3),         # M = 36 (one ragged tile), N = 20 (ragged columns)
    (3, 16, 32, 16, 132, 1),       # k = 1: no halo; N = 132: two column tiles, the second ragged
    (2, 24, 16, 8, 16, 5),         # k = 5: two-pixel halo, image exactly one K tile wide
    (4, 40, 40, 32, 64, 3),        # rows that wrap inside a K tile (40 is not a multiple of 16): per-lane offsets
    (64, 32, 16, 32, 16, 3),       # reference conv_test.py shape
    (1, 5, 7, 3, 5, 3), (2, 6, 10, 4, 8, 3),      # off the fused path: channels / pixels / width
])
def test_conv_bwd_w_relu(npm, n, h, w, c0, c1, k, fused):
    """npm_conv2d_bwd_w_relu = the first three lines of Conv2D.backward (conv.py:54-56 with activations.py:19): g,
    db and dw from dy, the saved pre-activation and x, with the ReLU mask applied inside the filter-gradient kernel
    (NPM_TUNE_CONV_WGRAD_FUSED 1: tile height picked, 2 / 3: 128- / 192-row tiles, 0: the two-pass form).  The mask
    is exact at pre = +0 and -0 (x >= 0 keeps dy); g and db are bit-equal to the elementwise reference."""
    from np_modeling_amd import _C, device as D
    lib = _C.lib()
    rng = np.random.default_rng(n * 7 + h + c1 + k)
    x = rng.standard_normal((n, h, w, c0)).astype(np.float32)
    dy = rng.standard_normal((n, h, w, c1)).astype(np.float32)
    pre = rng.standard_normal((n, h, w, c1)).astype(np.float32)
    pre.reshape(-1)[::7] = 0.0
    pre.reshape(-1)[3::11] = -0.0
    guard = 96
    gbuf, dwbuf, dbbuf = D.full([dy.size + guard], 777.0), D.full([k * k * c0 * c1 + guard], 777.0), D.full([c1 + guard], 777.0)
    ddy, dpre, dx_ = D.from_host(dy), D.from_host(pre), D.from_host(x)          # (kept alive: the pool reuses freed blocks)
    _C.check(lib.npm_set_tuning(13, fused), 'npm_set_tuning')
    try:
        _C.check(lib.npm_conv2d_bwd_w_relu(ddy.ptr, dpre.ptr, dx_.ptr, gbuf.ptr, dwbuf.ptr, dbbuf.ptr, n, h, w, c0, c1, k),
                 'npm_conv2d_bwd_w_relu')
    finally:
        _C.check(lib.npm_set_tuning(13, 1), 'npm_set_tuning')
    want_g = np.where(pre >= 0, dy, 0).astype(np.float32)
    for buf, size in ((gbuf, dy.size), (dwbuf, k * k * c0 * c1), (dbbuf, c1)):
        np.testing.assert_array_equal(buf.numpy()[size:], 777.0)          # nothing written past the end
    np.testing.assert_array_equal(gbuf.numpy()[:dy.size].reshape(dy.shape), want_g)
    assert_close(dbbuf.numpy()[:c1], want_g.astype(np.float64).reshape(-1, c1).sum(axis=0), tol=3e-6)
    assert_close(dwbuf.numpy()[:k * k * c0 * c1].reshape(k, k, c0, c1),
                 O.conv2d_grad_w(want_g.astype(np.float64), x.astype(np.float64), k), tol=5e-6)


def test_conv_bwd_w_relu_is_deterministic(npm):
    from np_modeling_amd import _C, device as D
    rng = np.random.default_rng(5)
    n, h, w, c0, c1, k = 8, 32, 32, 64, 128, 3
    x, dy, pre = (D.from_host(rng.standard_normal(s).astype(np.float32)) for s in ((n, h, w, c0), (n, h, w, c1), (n, h, w, c1)))
    runs = []
    for _ in range(2):
        g, dw, db = D.empty([n, h, w, c1]), D.empty([k, k, c0, c1]), D.empty([c1])
        _C.check(_C.lib().npm_conv2d_bwd_w_relu(dy.ptr, pre.ptr, x.ptr, g.ptr, dw.ptr, db.ptr, n, h, w, c0, c1, k))
        runs.append((dw.numpy(), db.numpy()))
    np.testing.assert_array_equal(runs[0][0], runs[1][0])
    np.testing.assert_array_equal(runs[0][1], runs[1][1])


def test_conv_bwd_w_relu_rendezvous_changes_nothing(npm):
    """The K rendezvous of the co-resident filter-gradient blocks (NPM_TUNE_KSYNC) engages when every split is at least
    256 K tiles long (C3-like sizes: here 2^20 pixels): dw, db and g are bit-equal with it off, on, and at a short interval."""
    from np_modeling_amd import _C, device as D
    lib = _C.lib()
    n, h, w, c0, c1, k = 16, 256, 256, 64, 128, 3
    x, dy, pre = D.empty([n, h, w, c0]), D.empty([n, h, w, c1]), D.empty([n, h, w, c1])
    rng = np.random.default_rng(6)
    row = rng.standard_normal(h * w * c1).astype(np.float32)
    for i in range(n):                                       # cheap, non-repeating enough: rolled copies of one image
        for dst, src in ((dy, np.roll(row, 7 * i)), (pre, np.roll(row, 11 * i + 3)), (x, np.roll(row[:h * w * c0], 5 * i + 1))):
            src = np.ascontiguousarray(src)                  # (kept alive across the copy)
            _C.check(lib.npm_h2d(dst.ptr + 4 * i * src.size, src.ctypes.data, 4 * src.size))
    runs = {}
    _C.check(lib.npm_set_tuning(13, 3))                      # the four-wave kernel, three blocks per split (the default at this shape
    try:                                                     # is one block of twelve waves, which needs no rendezvous: next test)
        for every in (0, 128, 32):
            _C.check(lib.npm_set_tuning(15, every))
            try:
                g, dw, db = D.empty([n, h, w, c1]), D.empty([k, k, c0, c1]), D.empty([c1])
                _C.check(lib.npm_conv2d_bwd_w_relu(dy.ptr, pre.ptr, x.ptr, g.ptr, dw.ptr, db.ptr, n, h, w, c0, c1, k))
                runs[every] = (dw.numpy().copy(), db.numpy().copy(), g.numpy()[::5, ::37, ::41].copy())
            finally:
                _C.check(lib.npm_set_tuning(15, 128))
    finally:
        _C.check(lib.npm_set_tuning(13, 1))
    for every in (128, 32):
        for got, want in zip(runs[every], runs[0]):
            np.testing.assert_array_equal(got, want)
    assert np.isfinite(runs[0][0]).all() and np.abs(runs[0][0]).max() > 0


@pytest.mark.gpu
@pytest.mark.parametrize('n,h,w', [(4, 64, 64), (3, 40, 48), (16, 128, 128)])
def test_conv_bwd_w_relu_twelve_wave_block_equals_three_blocks(npm, n, h, w):
    """k*k*C0 = 576 (C3's filter: three tile rows of 192): by default ONE block of twelve waves carries the three tile rows of a
    split and they share one masked dy tile (conv_wgrad_relu_kernel<3, true>); NPM_TUNE_CONV_WGRAD_FUSED = 3 runs the three
    four-wave blocks of rounds 3-4.  Same products in the same order: dw and g are bit-equal; db is summed in another grouping
    (16 k-row groups per block against 8 per tile row with turns) and agrees to rounding; all three match the elementwise /
    NumPy reference.  Includes a pixel count whose splits are uneven and borders on every tile."""
    from np_modeling_amd import _C, device as D
    lib = _C.lib()
    c0, c1, k = 64, 128, 3
    rng = np.random.default_rng(n * h + w)
    xh = rng.standard_normal([n, h, w, c0]).astype(np.float32)
    dyh = rng.standard_normal([n, h, w, c1]).astype(np.float32)
    preh = rng.standard_normal([n, h, w, c1]).astype(np.float32)
    preh[0, 0, :8, :4] = np.array([0.0, -0.0, 1e-30, -1e-30], dtype=np.float32)      # the mask is x >= 0, +-0 included
    x, dy, pre = D.from_host(xh), D.from_host(dyh), D.from_host(preh)
    runs = {}
    for mode in (1, 3):
        _C.check(lib.npm_set_tuning(13, mode))
        try:
            g, dw, db = D.full([n, h, w, c1], 7.0), D.empty([k, k, c0, c1]), D.empty([c1])
            _C.check(lib.npm_conv2d_bwd_w_relu(dy.ptr, pre.ptr, x.ptr, g.ptr, dw.ptr, db.ptr, n, h, w, c0, c1, k))
            runs[mode] = (dw.numpy().copy(), g.numpy().copy(), db.numpy().copy())
        finally:
            _C.check(lib.npm_set_tuning(13, 1))
    np.testing.assert_array_equal(runs[1][0], runs[3][0])
    np.testing.assert_array_equal(runs[1][1], runs[3][1])
    gref = np.where(preh >= 0, dyh, np.float32(0))
    np.testing.assert_array_equal(runs[1][1], gref)
    dbref = gref.astype(np.float64).sum(axis=(0, 1, 2))
    for mode in (1, 3):
        assert np.abs(runs[mode][2] - dbref).max() <= 1e-5 * np.abs(dbref).max() + 1e-4
    # dw against fp64 on a few filter taps (the whole product is test_conv_bwd_w_relu's business)
    xp = np.pad(xh.astype(np.float64), ((0, 0), (1, 1), (1, 1), (0, 0)))
    for (ti, tj) in ((0, 0), (1, 1), (2, 1)):
        want = np.einsum('nhwc,nhwd->cd', xp[:, ti:ti + h, tj:tj + w, :], gref.astype(np.float64))
        got = runs[1][0][ti, tj]
        assert np.abs(got - want).max() <= 2e-5 * np.abs(want).max()


def test_conv_rejects_even_kernel(npm):
    layer = npm.layers.Conv2D(channels=4, kernel_size=2)
    with pytest.raises(AssertionError):
        layer(rand([1, 4, 4, 3]))
    with pytest.raises(AssertionError):
        npm.layers.Conv2D(channels=4, kernel_size=3, padding='VALID')
    with pytest.raises(AssertionError):
        npm.layers.Conv2D(channels=4, kernel_size=3, strides=(2, 2))


def test_trainer_conv_stack(npm, capsys):
    """reference train_test.py:51-81 (k = 1,3,5,3,1; c = 16,32,64,32,16 on [16,32,32,16], SGD 1e-6),
    3 steps, losses and final filters against the oracle run on the same seeded parameters."""
    import re
    np.random.seed(0)
    ks, cs = [1, 3, 5, 3, 1], [16, 32, 64, 32, 16]
    stack = [npm.layers.Conv2D(channels=c, kernel_size=k, name=f'layer_{i}') for i, (c, k) in enumerate(zip(cs, ks))]
    x = np.random.uniform(-1.0, 1.0, size=[16, 32, 32, 16]).astype(np.float32)
    t = np.random.uniform(0.0, 1.0, size=[16, 32, 32, 16]).astype(np.float32)
    state = np.random.get_state()
    trainer = npm.train.Trainer(stack)
    trainer.train(inputs=x, targets=t, steps=3, optimizer_=npm.optimizer.SGDOptimizer(1e-6))
    losses = [float(v) for v in re.findall(r'Loss:\s+([0-9.eE+-]+)', capsys.readouterr().out)]
    # oracle: same draws (w then b per layer at first forward), same loop
    np.random.set_state(state)
    params = []
    cin = 16
    for c, k in zip(cs, ks):
        params.append([O.random_init([k, k, cin, c]).astype(np.float64), O.random_init([c]).astype(np.float64)])
        cin = c
    want = []
    for _ in range(3):
        acts, pres = [x.astype(np.float64)], []
        for wgt, b in params:
            y, pre = O.conv_layer_fwd(acts[-1], wgt, b)
            acts.append(y)
            pres.append(pre)
        want.append(O.mse_fwd(acts[-1], t))
        dy = O.mse_bwd(acts[-1], t)
        for i in reversed(range(5)):
            dy, dw, db = O.conv_layer_bwd(acts[i], params[i][0], pres[i], dy)
            params[i][0] = params[i][0] - 1e-6 * dw
            params[i][1] = params[i][1] - 1e-6 * db
    np.testing.assert_allclose(losses, want, rtol=2e-5)
    for i, layer in enumerate(stack):
        assert_close(layer.w, params[i][0], tol=1e-5, what=f'w{i}')
