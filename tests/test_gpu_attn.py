"""GPU: the fused attention core (csrc/npm_attn.hip: npm_mha_core_fwd / npm_mha_core_bwd) through the C ABI against
the oracle's restatement of reference layers/attentions.py:103-112,146-162, and through the layer (masks, the
GEMM composition as a second opinion)."""

import ctypes as C

import numpy as np
import pytest

from conftest import assert_close as _assert_close
from oracle import np_oracle as O


def assert_close(got, ref, tol=1e-5, what=''):
    """conftest's metric with a floor on the scale: operands here are O(1), and a single-key problem has dq = dk = 0
    exactly (P = 1, dS = P (dP - delta) = 0), which fp32 returns as rounding noise of that magnitude times 1e-7."""
    ref = np.asarray(ref, dtype=np.float64)
    if ref.size and np.abs(ref).max() < 1.0:
        np.testing.assert_allclose(np.asarray(got, dtype=np.float64), ref, rtol=tol, atol=tol, err_msg=what)
    else:
        _assert_close(got, ref, tol=tol, what=what)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def npm():
    import np_modeling_amd
    return np_modeling_amd


@pytest.fixture(params=[3, 2, 1, 0], ids=['bwd8', 'default', 'bwd16', 'bwd4'], autouse=True)
def bwd_kernel(request, npm):
    """Every test of this module under each choice of the attention backward (NPM_TUNE_ATTN_BWD16, include/npm_hip.h):
    3 = mha_bwd8_kernel always (8 waves, one barrier per tile, every head size), 2 = the default (the same, except head size 128
    with saved scores and no tile summary -> mha_bwd16_kernel), 1 = round 3's choice (mha_bwd16_kernel for head size 128 with saved
    scores, else the 4-wave kernel), 0 = the 4-wave kernel."""
    from np_modeling_amd import _C
    _C.check(_C.lib().npm_set_tuning(14, request.param), 'npm_set_tuning')
    yield request.param
    _C.check(_C.lib().npm_set_tuning(14, 2), 'npm_set_tuning')


@pytest.fixture(params=[2, 0], ids=['fwd8', 'fwd4'], autouse=True)
def fwd_kernel(request, npm):
    """... and under each attention forward (NPM_TUNE_ATTN_FWD8): 2 = mha_fwd8_kernel (8 waves on the 16x16x4 MFMA), 0 = the 4-wave
    32x32x2 mha_fwd_kernel."""
    from np_modeling_amd import _C
    _C.check(_C.lib().npm_set_tuning(17, request.param), 'npm_set_tuning')
    yield request.param
    _C.check(_C.lib().npm_set_tuning(17, 2), 'npm_set_tuning')


def _run_core(npm, q, k, v, scale, dctx=None, mask=None, save=False, packed=False, lse_ctx=None, skip=True, neg_delta=None):
    """q [B,Sq,H,D], k/v [B,Skv,H,D] host arrays -> dict of host results from the C ABI.  ``packed``: q, k, v
    live in one [B, S, 3, H, D] buffer (row pitch 3 H D), like the layer's packed projection."""
    from np_modeling_amd import _C, device as D
    lib = _C.lib()
    b, sq, h, d = q.shape
    skv = k.shape[1]
    guard = 64
    if packed:
        assert sq == skv
        buf = np.stack([q, k, v], axis=2)                       # [B, S, 3, H, D]
        dev = D.from_host(buf)
        qd, kd, vd = dev, dev.flat_view(h * d, [dev.size - h * d]), dev.flat_view(2 * h * d, [dev.size - 2 * h * d])
        pitch = 3 * h * d
        pitches = (pitch, pitch, pitch)
    else:
        qd, kd, vd = D.from_host(q), D.from_host(k), D.from_host(v)
        pitches = (h * d, h * d, h * d)
    ctx = D.full([b * sq * h * d + guard], 777.0)               # guard region behind every output
    lse = D.full([b * h * sq + guard], 777.0)
    c = _C.npm_mha_core()
    c.batch, c.heads, c.seq_q, c.seq_kv, c.head_dim, c.scale = b, h, sq, skv, d, scale
    c.q, c.k, c.v = qd.ptr, kd.ptr, vd.ptr
    c.q_pitch, c.k_pitch, c.v_pitch = pitches
    c.ctx, c.ctx_pitch, c.lse = ctx.ptr, h * d, lse.ptr
    mask_dev = None
    if mask is not None:
        mask_dev = D.AttnMask(mask, b, h, sq, skv)
        c.mask = mask_dev.buf.ptr
        c.mask_stride_b, c.mask_stride_h, c.mask_stride_q = mask_dev.strides
        if skip and mask_dev.summary is not None:               # the mask's tile summary: the kernels skip empty tiles
            c.tile_summary = mask_dev.summary.ptr
            c.summary_stride_b, c.summary_stride_h = mask_dev.summary_strides
            c.summary_all_offset = mask_dev.summary_all_offset
    scores = None
    if save:
        scores = D.full([b * h * sq * skv + guard], 777.0)
        c.scores = scores.ptr
    _C.check(lib.npm_mha_core_fwd(C.byref(c)), 'npm_mha_core_fwd')
    out = {'ctx': ctx.numpy(), 'lse': lse.numpy()}
    for key, n in (('ctx', b * sq * h * d), ('lse', b * h * sq)):
        np.testing.assert_array_equal(out[key][n:], 777.0)      # nothing written past the end
        out[key] = out[key][:n]
    out['ctx'] = out['ctx'].reshape(b, sq, h, d)
    out['lse'] = out['lse'].reshape(b, h, sq)
    if save:
        raw = scores.numpy()
        np.testing.assert_array_equal(raw[b * h * sq * skv:], 777.0)
        out['scores'] = raw[:b * h * sq * skv].reshape(b, h, sq, skv)
    if dctx is not None:
        if lse_ctx is not None:                                 # the oracle's forward results: isolates the backward
            ctx.flat_view(0, [b, sq, h, d]).set(lse_ctx[1])
            lse.flat_view(0, [b, h, sq]).set(lse_ctx[0])
        grads = {}
        if packed:
            gbuf = D.full([b * sq * 3 * h * d + guard], 777.0)
            views = [gbuf.flat_view(i * h * d, [gbuf.size - i * h * d]) for i in range(3)]
            gp = (3 * h * d,) * 3
        else:
            views = [D.full([b * s * h * d + guard], 777.0) for s in (sq, skv, skv)]
            gp = (h * d,) * 3
        dctx_d = D.from_host(dctx)
        c.dctx, c.dctx_pitch = dctx_d.ptr, h * d
        c.dq, c.dk, c.dv = (x.ptr for x in views)
        c.dq_pitch, c.dk_pitch, c.dv_pitch = gp
        if neg_delta is not None:                               # [H, B, Sq] row terms taken by the caller (include/npm_hip.h)
            nd = D.from_host(neg_delta)
            c.neg_delta, c.neg_delta_stride_b, c.neg_delta_stride_h = nd.ptr, neg_delta.shape[2], neg_delta.shape[1] * neg_delta.shape[2]
        _C.check(lib.npm_mha_core_bwd(C.byref(c)), 'npm_mha_core_bwd')
        if packed:
            raw = gbuf.numpy()
            np.testing.assert_array_equal(raw[b * sq * 3 * h * d:], 777.0)
            g = raw[:b * sq * 3 * h * d].reshape(b, sq, 3, h, d)
            grads = {'dq': g[:, :, 0], 'dk': g[:, :, 1], 'dv': g[:, :, 2]}
        else:
            for name, x, s in zip(('dq', 'dk', 'dv'), views, (sq, skv, skv)):
                raw = x.numpy()
                np.testing.assert_array_equal(raw[b * s * h * d:], 777.0)
                grads[name] = raw[:b * s * h * d].reshape(b, s, h, d)
        out.update(grads)
    return out


SHAPES = [  # b, h, sq, skv, d
    (1, 1, 1, 1, 16), (2, 3, 32, 32, 16), (2, 2, 33, 47, 32), (1, 2, 100, 257, 64), (2, 2, 128, 128, 128),
    (1, 3, 200, 130, 128), (3, 1, 31, 300, 16), (1, 1, 130, 5, 64), (1, 2, 512, 512, 128), (2, 8, 32, 128, 16),
    (2, 2, 100, 300, 16), (1, 3, 70, 260, 32), (1, 2, 257, 129, 16),      # several query tiles x several key blocks, ragged: the spread dQ sum (16), two blocks per CU (32)
]


@pytest.mark.parametrize('b,h,sq,skv,d', SHAPES)
def test_core_vs_oracle(npm, b, h, sq, skv, d):
    rng = np.random.default_rng(b * 1000 + sq * 7 + skv + d)
    q = rng.standard_normal([b, sq, h, d]).astype(np.float32)
    k = rng.standard_normal([b, skv, h, d]).astype(np.float32)
    v = rng.standard_normal([b, skv, h, d]).astype(np.float32)
    dctx = rng.standard_normal([b, sq, h, d]).astype(np.float32)
    scale = 1.0 / np.sqrt(d)
    q64, k64, v64, d64 = (x.astype(np.float64) for x in (q, k, v, dctx))
    ctx, lse, probs = O.attention_core_fwd(q64, k64, v64, scale)
    dq, dk, dv = O.attention_core_bwd(q64, k64, v64, probs, d64, scale)
    got = _run_core(npm, q, k, v, scale, dctx=dctx, lse_ctx=(lse, ctx))
    assert_close(got['ctx'], ctx, tol=2e-6)
    np.testing.assert_allclose(got['lse'], lse, rtol=0, atol=3e-6)
    assert_close(got['dv'], dv, tol=3e-6)
    assert_close(got['dq'], dq, tol=3e-6)
    assert_close(got['dk'], dk, tol=3e-6)
    # end to end: the kernel's own lse / ctx feed its backward
    own = _run_core(npm, q, k, v, scale, dctx=dctx)
    for name, want in (('dq', dq), ('dk', dk), ('dv', dv)):
        assert_close(own[name], want, tol=3e-6, what=name)


@pytest.mark.parametrize('b,h,sq,skv', [(1, 1, 1, 1), (2, 3, 32, 32), (2, 2, 33, 47), (1, 2, 100, 257), (1, 3, 200, 130),
                                         (3, 1, 31, 300), (1, 1, 130, 5), (1, 2, 512, 512), (2, 1, 17, 128), (1, 1, 64, 129)])
def test_core_saved_scores_head_128(npm, b, h, sq, skv, bwd_kernel):
    """Head size 128 with saved scores -- the default path of C4 / C5 -- through ragged lengths, under each backward kernel
    (the module's ``bwd_kernel`` fixture): mha_bwd8_kernel, mha_bwd16_kernel, the 4-wave 32x32x2 kernel; against the oracle,
    and the saved scores themselves."""
    from np_modeling_amd import _C
    d = 128
    bwd16 = bwd_kernel
    rng = np.random.default_rng(b * 1000 + sq * 7 + skv)
    q = rng.standard_normal([b, sq, h, d]).astype(np.float32)
    k = rng.standard_normal([b, skv, h, d]).astype(np.float32)
    v = rng.standard_normal([b, skv, h, d]).astype(np.float32)
    dctx = rng.standard_normal([b, sq, h, d]).astype(np.float32)
    scale = 1.0 / np.sqrt(d)
    q64, k64, v64, d64 = (x.astype(np.float64) for x in (q, k, v, dctx))
    ctx, lse, probs = O.attention_core_fwd(q64, k64, v64, scale)
    dq, dk, dv = O.attention_core_bwd(q64, k64, v64, probs, d64, scale)
    got = _run_core(npm, q, k, v, scale, dctx=dctx, save=True)
    assert _C.last_attn_kernel().startswith({3: 'mha_bwd8_kernel', 2: 'mha_bwd16_kernel', 1: 'mha_bwd16_kernel', 0: 'mha_bwd_kernel'}[bwd16] + ' D=128')
    assert_close(got['ctx'], ctx, tol=2e-6)
    assert_close(got['scores'], np.einsum('bqhd,bkhd->bhqk', q64, k64), tol=2e-6)
    for name, want in (('dq', dq), ('dk', dk), ('dv', dv)):
        assert_close(got[name], want, tol=3e-6, what=f'{name} bwd16={bwd16}')


@pytest.mark.parametrize('b,h,sq,skv,d', [(2, 2, 64, 64, 128), (1, 3, 200, 130, 128), (2, 2, 36, 70, 64), (1, 2, 128, 96, 32),
                                           (2, 1, 33, 47, 128), (1, 2, 130, 64, 16), (1, 1, 7, 9, 64)])
def test_core_row_terms_from_the_caller(npm, b, h, sq, skv, d, bwd_kernel):
    """``npm_mha_core.neg_delta``: MINUS scale * (dctx_i . ctx_i) handed in by the caller (the layer takes it from the GEMM that
    produces dctx).  Any head size on the eight-wave kernels; a sequence length that is not a multiple of 4 (rows not aligned
    for the kernels' 16-byte row-term loads) and the four-wave kernels recompute the terms instead -- same results either way.
    The terms handed in are deliberately the EXACT ones: a kernel that ignored the pointer where it should not would still
    pass, so a second run with WRONG terms must change dq exactly where the pointer is honoured."""
    rng = np.random.default_rng(b * 77 + sq * 5 + skv + d)
    q = rng.standard_normal([b, sq, h, d]).astype(np.float32)
    k = rng.standard_normal([b, skv, h, d]).astype(np.float32)
    v = rng.standard_normal([b, skv, h, d]).astype(np.float32)
    dctx = rng.standard_normal([b, sq, h, d]).astype(np.float32)
    scale = 1.0 / np.sqrt(d)
    q64, k64, v64, d64 = (x.astype(np.float64) for x in (q, k, v, dctx))
    ctx, lse, probs = O.attention_core_fwd(q64, k64, v64, scale)
    dq, dk, dv = O.attention_core_bwd(q64, k64, v64, probs, d64, scale)
    terms = -scale * np.einsum('bqhd,bqhd->hbq', d64, ctx)
    got = _run_core(npm, q, k, v, scale, dctx=dctx, save=d >= 64, lse_ctx=(lse, ctx), neg_delta=terms.astype(np.float32))
    for name, want in (('dq', dq), ('dk', dk), ('dv', dv)):
        assert_close(got[name], want, tol=3e-6, what=name)
    wrong = _run_core(npm, q, k, v, scale, dctx=dctx, save=d >= 64, lse_ctx=(lse, ctx), neg_delta=(terms + 1.0).astype(np.float32))
    honoured = bwd_kernel != 0 and sq % 4 == 0 and not (bwd_kernel == 1 and d != 128)
    assert (np.abs(wrong['dq'] - got['dq']).max() > 1e-3) == honoured
    assert_close(wrong['dv'], dv, tol=3e-6)                      # dV = P^T dctx does not involve the row terms


@pytest.mark.parametrize('b,h,s,d', [(2, 4, 96, 16), (1, 8, 160, 128), (2, 2, 64, 64)])
def test_core_packed_operands_and_saved_scores(npm, b, h, s, d):
    """q/k/v (and dq/dk/dv) as thirds of one [B, S, 3, H, D] buffer (row pitch 3 H D), and the variant that keeps
    the raw scores for the backward instead of recomputing q.k."""
    rng = np.random.default_rng(s + d)
    q, k, v, dctx = (rng.standard_normal([b, s, h, d]).astype(np.float32) for _ in range(4))
    scale = 1.0 / np.sqrt(d)
    ctx, lse, probs = O.attention_core_fwd(*(x.astype(np.float64) for x in (q, k, v)), scale)
    dq, dk, dv = O.attention_core_bwd(*(x.astype(np.float64) for x in (q, k, v)), probs, dctx.astype(np.float64), scale)
    for save in (False, True):
        got = _run_core(npm, q, k, v, scale, dctx=dctx, packed=True, save=save)
        assert_close(got['ctx'], ctx, tol=2e-6)
        for name, want in (('dq', dq), ('dk', dk), ('dv', dv)):
            assert_close(got[name], want, tol=3e-6, what=f'{name} save={save}')
        if save:
            raw = np.einsum('bqhd,bkhd->bhqk', q.astype(np.float64), k.astype(np.float64))
            assert_close(got['scores'], raw, tol=2e-6)


@pytest.mark.parametrize('b,h,sq,skv,d', [(2, 2, 40, 70, 16), (1, 2, 130, 130, 128), (2, 1, 64, 33, 32)])
def test_core_masked(npm, b, h, sq, skv, d):
    """np.where(mask, scaled, -inf) (attentions.py:105-107): causal, random, and broadcast masks; every row keeps
    at least one position."""
    rng = np.random.default_rng(sq + skv)
    q = rng.standard_normal([b, sq, h, d]).astype(np.float32)
    k, v = (rng.standard_normal([b, skv, h, d]).astype(np.float32) for _ in range(2))
    dctx = rng.standard_normal([b, sq, h, d]).astype(np.float32)
    scale = 1.0 / np.sqrt(d)
    causal = np.tril(np.ones([sq, skv], dtype=bool))[None, None]                    # broadcast over batch and head
    random = rng.random([b, h, sq, skv]) < 0.6
    random[..., 0] = True
    per_head = (rng.random([1, h, 1, skv]) < 0.7)
    per_head[..., 3] = True
    for mask in (causal, random, per_head):
        full = np.broadcast_to(mask, (b, h, sq, skv))
        ctx, lse, probs = O.attention_core_fwd(*(x.astype(np.float64) for x in (q, k, v)), scale, full)
        dq, dk, dv = O.attention_core_bwd(*(x.astype(np.float64) for x in (q, k, v)), probs, dctx.astype(np.float64), scale)
        for save in (False, True):
            got = _run_core(npm, q, k, v, scale, dctx=dctx, mask=mask, save=save)
            assert_close(got['ctx'], ctx, tol=2e-6)
            np.testing.assert_allclose(got['lse'], lse, rtol=0, atol=3e-6)
            for name, want in (('dq', dq), ('dk', dk), ('dv', dv)):
                assert_close(got[name], want, tol=3e-6, what=name)


@pytest.mark.parametrize('d', [16, 64, 128])
def test_core_masks_that_hide_whole_leading_tiles(npm, d):
    """Rows whose FIRST 32-key tile(s) are fully masked while later keys are allowed: sliding window (row i sees
    keys [i - 8, i]), left padding (the first 40 keys excluded), block-diagonal.  The online softmax starts from
    m = -inf there; np.where(mask, s, -inf) + softmax (attentions.py:105-109) gives finite values for these rows, so
    must the kernel -- forward and backward, with saved scores and with recomputed ones."""
    rng = np.random.default_rng(d)
    b, h, sq, skv = 2, 2, 130, 136
    q = rng.standard_normal([b, sq, h, d]).astype(np.float32)
    k, v = (rng.standard_normal([b, skv, h, d]).astype(np.float32) for _ in range(2))
    dctx = rng.standard_normal([b, sq, h, d]).astype(np.float32)
    scale = 1.0 / np.sqrt(d)
    i, j = np.arange(sq)[:, None], np.arange(skv)[None, :]
    band = ((j <= i) & (j >= i - 8))[None, None]
    left_pad = np.broadcast_to(j >= 40, (sq, skv))[None, None]
    blocks = ((i // 48) == (j // 48))[None, None]
    for name, mask in (('band', band), ('left padding', left_pad), ('block diagonal', blocks)):
        full = np.broadcast_to(mask, (b, h, sq, skv))
        assert full.any(axis=-1).all() and not full[..., :32].any(axis=-1).all()      # the case: leading tile hidden
        ctx, lse, probs = O.attention_core_fwd(*(x.astype(np.float64) for x in (q, k, v)), scale, full)
        dq, dk, dv = O.attention_core_bwd(*(x.astype(np.float64) for x in (q, k, v)), probs, dctx.astype(np.float64), scale)
        for save in (False, True):
            got = _run_core(npm, q, k, v, scale, dctx=dctx, mask=mask, save=save)
            assert np.isfinite(got['ctx']).all() and np.isfinite(got['lse']).all(), name
            assert_close(got['ctx'], ctx, tol=2e-6, what=name)
            np.testing.assert_allclose(got['lse'], lse, rtol=0, atol=3e-6)
            for g, want in (('dq', dq), ('dk', dk), ('dv', dv)):
                assert_close(got[g], want, tol=3e-6, what=f'{name} {g} save={save}')


def _summary_numpy(full, every=False):
    """byte (qt, kb): bit w = some (``every``: every in-range) position allowed in queries 32 qt .. x keys 128 kb + 16 w ..
    (include/npm_hip.h)"""
    nb, nh, sq, skv = full.shape
    nqt, nkb = (sq + 31) // 32, (skv + 127) // 128
    out = np.zeros([nb, nh, nqt, nkb], dtype=np.uint8)
    for qt in range(nqt):
        for kb in range(nkb):
            for w in range(8):
                sub = full[:, :, 32 * qt:32 * qt + 32, 128 * kb + 16 * w:128 * kb + 16 * w + 16]
                if sub.size:
                    out[:, :, qt, kb] |= ((sub.all(axis=(2, 3)) if every else sub.any(axis=(2, 3))).astype(np.uint8) << w)
    if not every:
        # a plane with a query row that has NO allowed key is never skipped: all of its "some position allowed" bytes read 0xFF
        # (the NaNs of that row reach every dk / dv row of the plane, as np.where(mask, s, -inf) + softmax has them)
        out[~full.any(axis=3).all(axis=2)] = 0xFF
    return out


def test_mask_summary(npm):
    """npm_mha_mask_summary against NumPy: dense, broadcast (batch / head / query) and ragged masks."""
    from np_modeling_amd import device as D
    rng = np.random.default_rng(0)
    kinds = set()
    for shape, (b, h, sq, skv) in (([2, 3, 70, 300], (2, 3, 70, 300)), ([1, 1, 129, 129], (4, 2, 129, 129)),
                                   ([1, 2, 1, 200], (3, 2, 50, 200)), ([2, 1, 33, 16], (2, 5, 33, 16))):
        mask = rng.random(shape) < 0.02
        mask[..., :40, 130:] = False
        mask[..., 160:] |= rng.random(shape[:2] + [1, 1]) < 0.5          # ... and whole sub-tiles without an excluded position
        mask[..., 0] |= rng.random(shape[:2] + [1]) < 0.6                 # most planes: every row keeps a key; the others have a
        # row without any -- both kinds must occur over the four shapes (asserted below)
        dev = D.AttnMask(mask, b, h, sq, skv)
        assert dev.summary is not None
        nb, nh = shape[0], shape[1]
        full = np.broadcast_to(mask, (nb, nh, sq, skv))
        got = dev.summary.numpy().reshape(2, nb, nh, (sq + 31) // 32, (skv + 127) // 128)
        np.testing.assert_array_equal(got[0], _summary_numpy(full))
        np.testing.assert_array_equal(got[1], _summary_numpy(full, every=True))
        assert dev.summary_all_offset == got[0].size
        assert dev.summary_strides == (0 if nb == 1 else got[0, 0].size, 0 if nh == 1 else got[0, 0, 0].size)
        kinds.update(full.any(axis=3).all(axis=2).reshape(-1).tolist())
    assert kinds == {True, False}, 'the shapes above must cover planes with and without a key-less row'


@pytest.mark.parametrize('d', [16, 64, 128])
@pytest.mark.parametrize('kind', ['causal', 'band', 'blocks', 'sparse', 'cross'])
def test_core_tile_skipping_changes_nothing(npm, d, kind, bwd_kernel):
    """With the mask's tile summary the kernels skip tiles (and, inside a tile, waves) that have no allowed position: results are
    BIT-equal to the unskipped run -- a skipped tile only ever added zeros -- and match the oracle; saved scores inside skipped
    tiles stay unwritten.  Masks with many empty tiles: causal, a narrow band, block-diagonal, 2 % random, and a cross-attention
    shape whose first key block is hidden from most queries."""
    rng = np.random.default_rng(d + len(kind))
    b, h, sq, skv = (2, 2, 288, 300) if kind != 'cross' else (1, 2, 200, 400)
    q = rng.standard_normal([b, sq, h, d]).astype(np.float32)
    k, v = (rng.standard_normal([b, skv, h, d]).astype(np.float32) for _ in range(2))
    dctx = rng.standard_normal([b, sq, h, d]).astype(np.float32)
    scale = 1.0 / np.sqrt(d)
    i, j = np.arange(sq)[:, None], np.arange(skv)[None, :]
    mask = {'causal': (j <= i)[None, None], 'band': ((j <= i) & (j >= i - 20))[None, None],
            'blocks': ((i // 96) == (j // 96))[None, None],
            'sparse': np.where(rng.random([b, h, sq, skv]) < 0.02, True, j == (i * 7) % skv),
            'cross': ((j >= 130) | (i < 20))[None, None]}[kind]
    full = np.broadcast_to(mask, (b, h, sq, skv))
    assert full.any(axis=-1).all()                                  # every query keeps a key: no NaN rows
    assert (_summary_numpy(full) == 0).any() or kind == 'sparse'    # ... and whole tiles are empty
    ctx, lse, probs = O.attention_core_fwd(*(x.astype(np.float64) for x in (q, k, v)), scale, full)
    dq, dk, dv = O.attention_core_bwd(*(x.astype(np.float64) for x in (q, k, v)), probs, dctx.astype(np.float64), scale)
    for save in (False, True):
        got = _run_core(npm, q, k, v, scale, dctx=dctx, mask=mask, save=save)
        plain = _run_core(npm, q, k, v, scale, dctx=dctx, mask=mask, save=save, skip=False)
        # (the default choice runs head size 128 with saved scores on mha_bwd16_kernel when there is NO summary and on
        # mha_bwd8_kernel when there is one: two kernels, equal to rounding only)
        same_kernel = not (bwd_kernel == 2 and d == 128 and save)
        # (bit equality needs whole query tiles: a tile without excluded positions runs WITHOUT the mask, so the idle lanes of a
        # ragged last tile -- queries beyond seq_q -- carry scores of 0 instead of -inf, and the wave-wide decision when to move the
        # softmax's reference point can fall differently: the 'cross' shape ends in 8 of 32 rows and is compared to rounding)
        for name in ('ctx', 'lse', 'dq', 'dk', 'dv'):
            if sq % 32:
                assert_close(got[name], plain[name], tol=5e-7, what=f'{kind} {name} save={save}')
            elif same_kernel or name in ('ctx', 'lse'):
                np.testing.assert_array_equal(got[name], plain[name], err_msg=f'{kind} {name} save={save}')
        assert_close(got['ctx'], ctx, tol=2e-6, what=kind)
        for name, want in (('dq', dq), ('dk', dk), ('dv', dv)):
            assert_close(got[name], want, tol=3e-6, what=f'{kind} {name} save={save}')
        if save and bwd_kernel >= 2:
            raw = np.einsum('bqhd,bkhd->bhqk', q.astype(np.float64), k.astype(np.float64))
            written = got['scores'] != 777.0
            assert written[full].all()                              # every allowed position was stored ...
            assert_close(np.where(full, got['scores'], 0), np.where(full, raw, 0), tol=2e-6)
            if kind != 'sparse':
                assert (~written).any()                             # ... and whole tiles were not


@pytest.mark.parametrize('rows', ['whole tile', 'one row', 'two rows in two tiles'])
def test_core_rows_without_any_key_are_never_skipped(npm, rows, bwd_kernel):
    """Query rows with NO allowed key: np.where(mask, s, -inf) + softmax makes such a row NaN -- its ctx and dq rows, and through
    P = NaN every dk / dv row of its (batch, head).  ONE rule in every kernel and mode (round-4 advisor): with the tile summary
    the results are those of the unskipped run, bit for bit, NaNs included -- npm_mha_mask_summary marks every tile of a plane
    that has such a row as "visit" -- whether the rows fill a whole 32-query tile (round 4 zero-filled their dq and kept dk / dv
    clean) or not.  The other head of the same batch is not touched by any of it."""
    rng = np.random.default_rng(5)
    b, h, sq, skv, d = 1, 2, 96, 160, 128
    q = rng.standard_normal([b, sq, h, d]).astype(np.float32)
    k, v = (rng.standard_normal([b, skv, h, d]).astype(np.float32) for _ in range(2))
    dctx = rng.standard_normal([b, sq, h, d]).astype(np.float32)
    mask = np.ones([b, h, sq, skv], dtype=bool)
    mask[0, 1] = (np.arange(skv)[None, :] <= np.arange(sq)[:, None] + 40)         # head 1: tiles to skip, no key-less row
    dead = {'whole tile': slice(32, 64), 'one row': slice(37, 38), 'two rows in two tiles': [5, 70]}[rows]
    mask[0, 0, dead, :] = False
    scale = 1.0 / np.sqrt(d)
    sub = [x.astype(np.float64) for x in (q[:, :, 1:], k[:, :, 1:], v[:, :, 1:])]
    ctx1, _, probs1 = O.attention_core_fwd(*sub, scale, mask[:, 1:])
    dq1, dk1, dv1 = O.attention_core_bwd(*sub, probs1, dctx[:, :, 1:].astype(np.float64), scale)
    alive = np.ones(sq, dtype=bool)
    alive[dead] = False
    for save in (False, True):
        got = _run_core(npm, q, k, v, scale, dctx=dctx, mask=mask, save=save)
        plain = _run_core(npm, q, k, v, scale, dctx=dctx, mask=mask, save=save, skip=False)
        same_kernel = not (bwd_kernel == 2 and d == 128 and save)      # (see test_core_tile_skipping_changes_nothing)
        for name in ('ctx', 'lse', 'dq', 'dk', 'dv'):
            assert (np.isnan(got[name]) == np.isnan(plain[name])).all(), f'{rows} {name} save={save}: NaNs differ from the unskipped run'
            if same_kernel or name in ('ctx', 'lse'):
                np.testing.assert_array_equal(got[name][:, :, :1], plain[name][:, :, :1], err_msg=f'{rows} {name} save={save}')   # head 0: unskipped
        # what NumPy gives: the dead rows NaN, and with them all of head 0's key gradients; every other row of ctx exact
        assert np.isnan(got['ctx'][0, ~alive, 0]).all() and np.isfinite(got['ctx'][0, alive, 0]).all()
        assert np.isnan(got['dq'][0, ~alive, 0]).all()
        assert np.isnan(got['dk'][0, :, 0]).all() and np.isnan(got['dv'][0, :, 0]).all()
        # head 1 (its own plane of the summary, with tiles that ARE skipped): finite and right
        assert_close(got['ctx'][:, :, 1:], ctx1, tol=2e-6)
        for name, want in (('dq', dq1), ('dk', dk1), ('dv', dv1)):
            assert np.isfinite(got[name][:, :, 1:]).all()
            assert_close(got[name][:, :, 1:], want, tol=3e-6, what=f'{rows} head 1 {name} save={save}')


def test_core_row_without_any_key_is_nan_and_only_that_row(npm):
    """A query row with NO key left is NaN, as the softmax of a row of -inf is in NumPy; every other row of the same
    wave, tile and head is untouched (forward)."""
    rng = np.random.default_rng(3)
    b, h, sq, skv, d = 1, 2, 70, 96, 64
    q = rng.standard_normal([b, sq, h, d]).astype(np.float32)
    k, v = (rng.standard_normal([b, skv, h, d]).astype(np.float32) for _ in range(2))
    mask = np.ones([b, h, sq, skv], dtype=bool)
    mask[0, 1, 37, :] = False
    mask[0, 0, 5, :64] = False
    scale = 1.0 / np.sqrt(d)
    with np.errstate(invalid='ignore'):
        ctx, lse, _ = O.attention_core_fwd(*(x.astype(np.float64) for x in (q, k, v)), scale, mask)
    for save in (False, True):
        got = _run_core(npm, q, k, v, scale, mask=mask, save=save)
        assert np.isnan(got['ctx'][0, 37, 1]).all()
        keep = np.ones([b, sq, h], dtype=bool)
        keep[0, 37, 1] = False
        assert np.isfinite(got['ctx'][keep]).all()
        assert_close(got['ctx'][keep], ctx[keep], tol=2e-6)


def test_core_rejects_unsupported_head_dim(npm):
    from np_modeling_amd import _C, device as D
    assert not D.mha_core_supported(24) and D.mha_core_supported(64) and not D.mha_core_supported(64, 32)
    x = D.zeros([1, 4, 1, 24])
    lse = D.zeros([4])
    c = _C.npm_mha_core()
    c.batch, c.heads, c.seq_q, c.seq_kv, c.head_dim, c.scale = 1, 1, 4, 4, 24, 0.2
    c.q = c.k = c.v = c.ctx = x.ptr
    c.q_pitch = c.k_pitch = c.v_pitch = c.ctx_pitch = 24
    c.lse = lse.ptr
    assert _C.lib().npm_mha_core_fwd(C.byref(c)) == 10003          # NPM_E_UNSUPPORTED
    assert b'head_dim' in _C.lib().npm_last_error()


def test_core_is_deterministic(npm):
    """dQ is summed over key blocks by one wave in program order: two runs give the same bits."""
    rng = np.random.default_rng(9)
    q, k, v, dctx = (rng.standard_normal([2, 384, 2, 128]).astype(np.float32) for _ in range(4))
    a = _run_core(npm, q, k, v, 0.09, dctx=dctx)
    b = _run_core(npm, q, k, v, 0.09, dctx=dctx)
    for name in ('ctx', 'lse', 'dq', 'dk', 'dv'):
        np.testing.assert_array_equal(a[name], b[name])


# ---- through the layer ------------------------------------------------------------------------------------
_MHA = ['wq', 'wk', 'wv', 'wo', 'bq', 'bk', 'bv', 'bo']


def _layer(npm, heads, feat, seed, kv_len=None):
    np.random.seed(seed)
    layer = npm.layers.MultiHeadAttention(num_heads=heads)
    probe = np.zeros([1, 4, feat], dtype=np.float32)
    layer(probe, np.zeros([1, kv_len and 4, feat], dtype=np.float32)) if kv_len else layer(probe)
    p = {}
    for n in _MHA:
        arr = np.asarray(getattr(layer, '_' + n)) / (np.sqrt(feat) if n[0] == 'w' else 1.0)
        p[n] = arr.astype(np.float32)
    return layer, p


@pytest.mark.parametrize('heads,feat,sq,skv', [(8, 128, 32, 32), (2, 256, 70, 70), (4, 128, 40, 100)])
def test_layer_fused_core_equals_gemm_composition_and_oracle(npm, heads, feat, sq, skv):
    from np_modeling_amd import device as D
    rng = np.random.default_rng(sq)
    cross = sq != skv
    query = rng.standard_normal([3, sq, feat]).astype(np.float32)
    kv = rng.standard_normal([3, skv, feat]).astype(np.float32) if cross else None
    dy = rng.standard_normal([3, sq, feat]).astype(np.float32)
    results = []
    saved_default = D.ATTN_SAVE_SCORES
    for core, save in ((True, True), (True, False), (False, False)):       # fused (scores kept), fused (log-sum-exp only), GEMMs
        D.ATTN_CORE, D.ATTN_SAVE_SCORES = core, save
        try:
            layer, p = _layer(npm, heads, feat, 3)
            for n in _MHA:
                setattr(layer, '_' + n, p[n].copy())
            out = np.asarray(layer(query, kv) if cross else layer(query))
            assert layer._core is core and (layer._raw_scores is not None) is (core and save) if core else True
            grads = [np.asarray(g) for g in layer(dy, backprop=True, learning_rate=0.05)]
            results.append((out, grads, {n: np.asarray(getattr(layer, '_' + n)) for n in _MHA}))
        finally:
            D.ATTN_CORE, D.ATTN_SAVE_SCORES = True, saved_default
    p64 = {n: p[n].astype(np.float64) for n in _MHA}
    want, cache = O.mha_fwd(p64, query.astype(np.float64), None if kv is None else kv.astype(np.float64))
    wg, pg = O.mha_bwd(p64, cache, dy.astype(np.float64))
    for out, grads, params in results:
        assert_close(out, want, tol=1e-5)
        if cross:
            assert_close(grads[0], wg[0], tol=1e-5)
            assert_close(grads[1] + grads[2], wg[1] + wg[2], tol=1e-5)
        else:
            assert_close(sum(grads), sum(wg), tol=1e-5)
        for n in _MHA:
            assert_close(params[n], p64[n] - 0.05 * pg[n], tol=1e-5, what=n)


def test_layer_masked_attention(npm):
    """The layer's mask semantics (np_modeling_amd/layers/attentions.py docstring): a boolean array broadcastable
    to [B, H, Sq, Skv]; forward and backward against the oracle's masked restatement."""
    rng = np.random.default_rng(12)
    heads, feat, b, s = 4, 64, 2, 48
    layer, p = _layer(npm, heads, feat, 5)
    for n in _MHA:
        setattr(layer, '_' + n, p[n].copy())
    query = rng.standard_normal([b, s, feat]).astype(np.float32)
    dy = rng.standard_normal([b, s, feat]).astype(np.float32)
    mask = np.tril(np.ones([s, s], dtype=bool))[None, None]
    out = layer(query, mask=mask)
    p64 = {n: p[n].astype(np.float64) for n in _MHA}
    want, cache = O.mha_fwd(p64, query.astype(np.float64), mask=np.broadcast_to(mask, (b, heads, s, s)))
    assert_close(out, want, tol=1e-5)
    grads = [np.asarray(g) for g in layer(dy, backprop=True, learning_rate=0.05)]
    wg, pg = O.mha_bwd(p64, cache, dy.astype(np.float64))
    assert_close(sum(grads), sum(wg), tol=1e-5)
    for n in _MHA:
        assert_close(getattr(layer, '_' + n), p64[n] - 0.05 * pg[n], tol=1e-5, what=n)
    # `if mask:` false in the reference: no mask
    np.testing.assert_array_equal(np.asarray(layer(query, mask=False)), np.asarray(layer(query)))
    with pytest.raises(AssertionError):
        layer(query, mask=np.ones([b, heads, s + 1, s], dtype=bool))


def test_core_bwd_without_queries_zeroes_key_value_grads(npm):
    """seq_q = 0: no query attends to the keys, so dk = dv = 0 (and nothing else is touched)."""
    from np_modeling_amd import _C, device as D
    b, h, skv, d = 2, 3, 40, 16
    one = D.full([16], 1.0)                                     # valid, never dereferenced: the query side is empty
    k = D.from_host(np.random.default_rng(0).normal(size=(b, skv, h, d)).astype(np.float32))
    dk, dv = D.full([b * skv * h * d + 64], 777.0), D.full([b * skv * h * d + 64], 777.0)
    c = _C.npm_mha_core()
    c.batch, c.heads, c.seq_q, c.seq_kv, c.head_dim, c.scale = b, h, 0, skv, d, 0.25
    c.q = c.ctx = c.lse = c.dctx = c.dq = one.ptr
    c.k = c.v = k.ptr
    c.q_pitch = c.k_pitch = c.v_pitch = c.ctx_pitch = c.dctx_pitch = c.dq_pitch = c.dk_pitch = c.dv_pitch = h * d
    c.dk, c.dv = dk.ptr, dv.ptr
    _C.check(_C.lib().npm_mha_core_fwd(C.byref(c)), 'npm_mha_core_fwd')
    _C.check(_C.lib().npm_mha_core_bwd(C.byref(c)), 'npm_mha_core_bwd')
    for g in (dk, dv):
        raw = g.numpy()
        np.testing.assert_array_equal(raw[:b * skv * h * d], 0.0)
        np.testing.assert_array_equal(raw[b * skv * h * d:], 777.0)
    np.testing.assert_array_equal(one.numpy(), 1.0)
