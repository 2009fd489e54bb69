"""GPU: npm_sgemm (fp32 MFMA GEMM) through the C ABI against NumPy fp64.

Covers every operand layout the hot path uses (NN / NT / TN), ragged and tiny shapes,
unaligned leading dimensions (scalar staging path), head-strided batched operands,
all epilogues and split-K."""

import numpy as np
import pytest

from conftest import assert_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def D():
    from np_modeling_amd import device
    return device


@pytest.fixture(autouse=True)
def _every_math_mode(math_mode):
    """Every test of this module runs under the exact-f32 MFMA and under both split-bf16 modes, at the same
    tolerances (shapes off the LDS-DMA path fall back to the f32 kernel in every mode)."""
    return math_mode


def _ref(a, b, ta, tb):
    a = a.astype(np.float64)
    b = b.astype(np.float64)
    return (a.T if ta else a) @ (b.T if tb else b)


@pytest.mark.parametrize('ta,tb', [(False, False), (False, True), (True, False)])
@pytest.mark.parametrize('m,n,k', [(128, 128, 32), (256, 384, 96), (64, 16, 32), (1, 1, 1), (5, 7, 3),
                                   (130, 129, 33), (300, 200, 500), (128, 128, 0), (37, 515, 260)])
def test_layouts_and_shapes(D, ta, tb, m, n, k):
    rng = np.random.default_rng(m * 1000003 + n * 1009 + k)
    a = rng.standard_normal((k, m) if ta else (m, k)).astype(np.float32)
    b = rng.standard_normal((n, k) if tb else (k, n)).astype(np.float32)
    da, db, dc = D.from_host(a), D.from_host(b), D.full([m, n], np.nan)
    D.gemm(m, n, k, D.Mat(da, a.shape[1] if a.ndim == 2 else 1), D.Mat(db, b.shape[1]), D.Mat(dc, n),
           trans_a=ta, trans_b=tb)
    assert_close(dc, _ref(a, b, ta, tb), tol=2e-6, what=f'{m}x{n}x{k} ta={ta} tb={tb}')


def test_identity_times_asymmetric_is_exact(D):
    """A = I with an asymmetric B catches a transposed C/D register map; fp32 MFMA is exact here."""
    n = 160
    b = (np.arange(n * n, dtype=np.float32).reshape(n, n) % 251) - 100.0
    dc = D.empty([n, n])
    D.gemm(n, n, n, D.Mat(D.from_host(np.eye(n, dtype=np.float32)), n), D.Mat(D.from_host(b), n), D.Mat(dc, n))
    np.testing.assert_array_equal(dc.numpy(), b)


def test_matches_fmaf_chain_bitwise_small_k(D):
    """The f32 MFMA is a k-ordered fmaf chain: with K <= 4 per lane-half group the result is
    bit-exact against float32 arithmetic done in the kernel's k order (k = 8g + 4h + s)."""
    rng = np.random.default_rng(0)
    a = rng.standard_normal((64, 8)).astype(np.float32)
    b = rng.standard_normal((8, 64)).astype(np.float32)
    dc = D.empty([64, 64])
    D.gemm(64, 64, 8, D.Mat(D.from_host(a), 8), D.Mat(D.from_host(b), 64), D.Mat(dc, 64))
    # MFMA step s consumes k = s (half 0) then k = 4 + s (half 1), accumulating in fp32 fma order
    acc = np.zeros((64, 64), dtype=np.float64)
    exact = np.zeros((64, 64), dtype=np.float32)
    for s in range(4):
        for kk in (s, 4 + s):
            prod = a[:, kk:kk + 1].astype(np.float64) * b[kk:kk + 1, :].astype(np.float64)
            exact = (exact.astype(np.float64) + prod).astype(np.float32)     # fma: one rounding
    np.testing.assert_array_equal(dc.numpy(), exact)


def test_epilogues(D):
    rng = np.random.default_rng(3)
    m, n, k = 200, 136, 72
    a = rng.standard_normal((m, k)).astype(np.float32)
    b = rng.standard_normal((k, n)).astype(np.float32)
    bias = rng.standard_normal(n).astype(np.float32)
    res = rng.standard_normal((m, n)).astype(np.float32)
    base = 0.5 * _ref(a, b, False, False) + bias + res
    da, db = D.from_host(a), D.from_host(b)
    # alpha + bias + residual
    c = D.empty([m, n])
    D.gemm(m, n, k, D.Mat(da, k), D.Mat(db, n), D.Mat(c, n), alpha=0.5, bias=D.from_host(bias),
           residual=D.Mat(D.from_host(res), n))
    assert_close(c, base, tol=2e-6)
    # accumulate in place (residual aliases C)
    c2 = D.from_host(res)
    D.gemm(m, n, k, D.Mat(da, k), D.Mat(db, n), D.Mat(c2, n), alpha=0.5, bias=D.from_host(bias),
           residual=D.Mat(c2, n))
    assert_close(c2, base, tol=2e-6)
    # relu with saved pre-activation
    pre = D.empty([m, n])
    y = D.empty([m, n])
    D.gemm(m, n, k, D.Mat(da, k), D.Mat(db, n), D.Mat(y, n), bias=D.from_host(bias), relu_save=D.Mat(pre, n))
    want_pre = _ref(a, b, False, False) + bias
    assert_close(pre, want_pre, tol=2e-6)
    np.testing.assert_array_equal(y.numpy(), np.maximum(pre.numpy(), 0.0))
    # relu mask (x >= 0 passes, including x == 0)
    mask_src = rng.standard_normal((m, n)).astype(np.float32)
    mask_src[0, :7] = 0.0
    out = D.empty([m, n])
    D.gemm(m, n, k, D.Mat(da, k), D.Mat(db, n), D.Mat(out, n), relu_mask=D.Mat(D.from_host(mask_src), n))
    want = np.where(mask_src >= 0, _ref(a, b, False, False), 0.0)
    assert_close(out, want, tol=2e-6)
    assert np.all(out.numpy()[0, :7] != 0.0)


@pytest.mark.parametrize('split', [0, 1, 3, 7])
def test_split_k_tall(D, split):
    """Weight-gradient shape: small M, N and a tall K; slabs are summed in fixed order."""
    rng = np.random.default_rng(11)
    m, n, k = 96, 160, 4096 + 40
    a = rng.standard_normal((k, m)).astype(np.float32)
    b = rng.standard_normal((k, n)).astype(np.float32)
    c = D.empty([m, n])
    D.gemm(m, n, k, D.Mat(D.from_host(a), m), D.Mat(D.from_host(b), n), D.Mat(c, n), trans_a=True, split_k=split)
    assert_close(c, _ref(a, b, True, False), tol=3e-6)
    c_again = D.empty([m, n])
    D.gemm(m, n, k, D.Mat(D.from_host(a), m), D.Mat(D.from_host(b), n), D.Mat(c_again, n), trans_a=True,
           split_k=split)
    np.testing.assert_array_equal(c.numpy(), c_again.numpy())       # reproducible


def test_batched_head_strided(D):
    """q k^T and p v addressed as head slices of [B, S, H, D] (no transposes)."""
    rng = np.random.default_rng(5)
    bsz, h, sq, skv, d = 3, 4, 40, 24, 16
    q = rng.standard_normal((bsz, sq, h, d)).astype(np.float32)
    k = rng.standard_normal((bsz, skv, h, d)).astype(np.float32)
    v = rng.standard_normal((bsz, skv, h, d)).astype(np.float32)
    dq, dk, dv = D.from_host(q), D.from_host(k), D.from_host(v)
    s = D.empty([bsz, h, sq, skv])
    D.gemm(sq, skv, d, D.Mat(dq, h * d, sq * h * d, d), D.Mat(dk, h * d, skv * h * d, d),
           D.Mat(s, skv, h * sq * skv, sq * skv), trans_b=True, batch=(bsz, h))
    want = np.einsum('bqhd,bkhd->bhqk', q.astype(np.float64), k.astype(np.float64))
    assert_close(s, want, tol=2e-6)
    ctx = D.empty([bsz, sq, h, d])
    D.gemm(sq, d, skv, D.Mat(s, skv, h * sq * skv, sq * skv), D.Mat(dv, h * d, skv * h * d, d),
           D.Mat(ctx, h * d, sq * h * d, d), batch=(bsz, h))
    want_ctx = np.einsum('bhqk,bkhd->bqhd', s.numpy().astype(np.float64), v.astype(np.float64))
    assert_close(ctx, want_ctx, tol=2e-6)
    # TN batched: dv = p^T dctx
    dvv = D.empty([bsz, skv, h, d])
    D.gemm(skv, d, sq, D.Mat(s, skv, h * sq * skv, sq * skv), D.Mat(ctx, h * d, sq * h * d, d),
           D.Mat(dvv, h * d, skv * h * d, d), trans_a=True, batch=(bsz, h))
    want_dv = np.einsum('bhqk,bqhd->bkhd', s.numpy().astype(np.float64), ctx.numpy().astype(np.float64))
    assert_close(dvv, want_dv, tol=2e-6)


def test_large_tile_aligned(D):
    """A shape that fills the chip (2048 blocks) with tile-aligned fast-path staging."""
    rng = np.random.default_rng(9)
    m, n, k = 4096, 1024, 512
    a = rng.standard_normal((m, k)).astype(np.float32)
    b = rng.standard_normal((n, k)).astype(np.float32)
    c = D.empty([m, n])
    D.gemm(m, n, k, D.Mat(D.from_host(a), k), D.Mat(D.from_host(b), k), D.Mat(c, n), trans_b=True)
    assert_close(c, _ref(a, b, False, True), tol=2e-6)


def test_bad_arguments_fail_loudly(D):
    from np_modeling_amd import _C
    with pytest.raises(_C.NpmError):
        D.gemm(8, 8, 8, D.Mat(0, 8), D.Mat(0, 8), D.Mat(0, 8))


@pytest.mark.parametrize('ta,tb', [(False, False), (False, True), (True, False)])
@pytest.mark.parametrize('m,n,k', [(64, 16, 32), (100, 200, 48), (129, 132, 16), (300, 4, 64)])
def test_ragged_tiles_do_not_write_outside_c(D, ta, tb, m, n, k):
    """The LDS-DMA kernel's epilogue relies on the buffer descriptor's range check to drop rows
    >= M and columns >= N of edge tiles: C sits inside a guarded allocation whose sentinels must
    survive, and C itself is pitched (ldc > n) with sentinels in the pitch gap."""
    rng = np.random.default_rng(m + n + k)
    a = rng.standard_normal((k, m) if ta else (m, k)).astype(np.float32)
    b = rng.standard_normal((n, k) if tb else (k, n)).astype(np.float32)
    ldc = n + 4
    guard = 4096
    whole = D.full([guard + m * ldc + guard], -777.0)
    c = whole.flat_view(guard, [m, ldc])
    D.gemm(m, n, k, D.Mat(D.from_host(a), a.shape[1]), D.Mat(D.from_host(b), b.shape[1]), D.Mat(c, ldc),
           trans_a=ta, trans_b=tb)
    host = whole.numpy()
    assert np.all(host[:guard] == -777.0) and np.all(host[guard + m * ldc:] == -777.0)
    body = host[guard:guard + m * ldc].reshape(m, ldc)
    assert np.all(body[:, n:] == -777.0)
    assert_close(body[:, :n], _ref(a, b, ta, tb), tol=2e-6)


@pytest.mark.parametrize('m,n,k,batch', [(256, 384, 64, (1, 1)), (100, 72, 48, (1, 1)), (130, 20, 33, (1, 1)),
                                         (40, 16, 32, (3, 4)), (512, 128, 128, (2, 8))])
def test_colsum_in_epilogue(D, m, n, k, batch):
    """Bias gradients taken in the producing GEMM's epilogue: per head (z1) sums over z0 and rows,
    of the STORED values (after the relu mask), on the DMA path and on the fallback path."""
    rng = np.random.default_rng(m + n)
    b0, b1 = batch
    a = rng.standard_normal((b0, m, b1, k)).astype(np.float32)            # [B, S, H, D] head slices
    w = rng.standard_normal((b0, k, b1, n)).astype(np.float32)
    mask = rng.standard_normal((b0, m, b1, n)).astype(np.float32)
    c = D.empty([b0, m, b1, n])
    sums = D.full([b1, n], np.nan)
    D.gemm(m, n, k, D.Mat(D.from_host(a), b1 * k, m * b1 * k, k), D.Mat(D.from_host(w), b1 * n, k * b1 * n, n),
           D.Mat(c, b1 * n, m * b1 * n, n), batch=batch, relu_mask=D.Mat(D.from_host(mask), b1 * n),
           colsum_out=sums)
    want = np.einsum('zmhk,zkhn->zmhn', a.astype(np.float64), w.astype(np.float64))
    want = np.where(mask >= 0, want, 0.0)
    assert_close(c, want, tol=2e-6)
    assert_close(sums, c.numpy().astype(np.float64).sum(axis=(0, 1)), tol=3e-6)


@pytest.mark.parametrize('m,n,k', [(128, 128, 64), (256, 384, 4096),      # LDS-DMA path, one split and split-K
                                   (1024, 256, 32768),                     # many splits: 8 tile rows x 2 tile columns
                                   (4096, 256, 2048), (256, 4096, 2048),   # more than 16 blocks share an operand tile
                                   (200, 72, 1040), (130, 20, 33), (5, 3, 7), (64, 640, 16), (384, 16, 2048)])
@pytest.mark.parametrize('which', ['bsum', 'asum'])
def test_bias_gradient_beside_weight_gradient(D, m, n, k, which):
    """dw = x^T dy with db = sum_k dy[k, :] (bsum: mlp.py:34-35) or sum_k of the first operand
    (asum: attentions.py:167-197) taken from the operand tiles of the same GEMM; fallback shapes too."""
    rng = np.random.default_rng(m * 31 + n * 7 + k)
    a = rng.standard_normal((k, m)).astype(np.float32)
    b = rng.standard_normal((k, n)).astype(np.float32)
    c = D.full([m, n], np.nan)
    length = n if which == 'bsum' else m
    sums = D.from_host(np.full(length + 2, 55.0, dtype=np.float32))         # guard words either side
    out = sums.flat_view(1, [length])
    D.gemm(m, n, k, D.Mat(D.from_host(a), m), D.Mat(D.from_host(b), n), D.Mat(c, n), trans_a=True,
           **{which + '_out': out})
    assert_close(c, _ref(a, b, True, False), tol=3e-6)
    src = b if which == 'bsum' else a
    host = sums.numpy()
    assert host[0] == 55.0 and host[-1] == 55.0
    assert_close(host[1:-1], src.astype(np.float64).sum(axis=0), tol=3e-6)


def test_split_math_error_statistics(D):
    """The split-bf16 modes carry fp32-class error: against fp64 at K = 4096 the rms error of 'bf16x3' is not above
    the exact-f32 MFMA's and its mean error is as small (no bias); 'bf16x3_fast' has the documented negative bias
    of a fraction of an ulp, far below its rms error."""
    import np_modeling_amd as npm
    rng = np.random.default_rng(0)
    m, n, k = 256, 256, 4096
    a = rng.standard_normal((m, k), dtype=np.float32)
    b = (rng.standard_normal((k, n), dtype=np.float32) / 64).astype(np.float32)      # outputs ~ N(0, 1)
    ref = a.astype(np.float64) @ b.astype(np.float64)
    da, db = D.from_host(a), D.from_host(b)
    stats = {}
    for mode in ('f32', 'bf16x3', 'bf16x3_fast'):
        npm.set_math(mode)
        c = D.empty([m, n])
        D.gemm(m, n, k, D.Mat(da, k), D.Mat(db, n), D.Mat(c, n))
        err = c.numpy().astype(np.float64) - ref
        stats[mode] = (err.mean(), np.sqrt((err ** 2).mean()))
    ulp = 2.0 ** -23
    assert stats['f32'][1] < 4 * ulp and abs(stats['f32'][0]) < 0.05 * ulp
    assert stats['bf16x3'][1] <= 1.05 * stats['f32'][1] and abs(stats['bf16x3'][0]) < 0.05 * ulp
    assert stats['bf16x3_fast'][1] <= 1.05 * stats['f32'][1] and abs(stats['bf16x3_fast'][0]) < 1.0 * ulp


def test_f16x2_error_contract(D):
    """NPM_MATH_F16X2 (csrc/npm_gemm_f16x2.hip): fp32-class error ROW-NORMWISE.  Against fp64 at K = 4096 the error of an
    output element relative to the largest element of its output row stays below the exact-f32 MFMA's, for N(0,1) rows,
    for rows whose magnitudes differ by 2^40 and for elements 2^30 apart inside a row; zero rows give exact zeros; a nan
    or an inf poisons exactly its own output row (A) or column (B)."""
    import np_modeling_amd as npm
    from oracle import np_oracle as O
    rng = np.random.default_rng(0)
    m, n, k = 256, 384, 4096
    b = (rng.standard_normal((k, n), dtype=np.float32) / 64).astype(np.float32)
    db = D.from_host(b)

    def rel_rms(a, mode):
        npm.set_math(mode)
        c = D.empty([m, n])
        D.gemm(m, n, k, D.Mat(D.from_host(a), k), D.Mat(db, n), D.Mat(c, n))
        assert npm.last_math() == mode
        ref = a.astype(np.float64) @ b.astype(np.float64)
        err = (c.numpy().astype(np.float64) - ref) / np.abs(ref).max(axis=1, keepdims=True)
        return np.sqrt((err ** 2).mean()), np.abs(err).max(), err.mean()

    try:
        base = rng.standard_normal((m, k), dtype=np.float32)
        rows = (base * np.exp2(rng.integers(-20, 21, size=(m, 1)))).astype(np.float32)          # rows 2^40 apart
        cols = (base * np.exp2(-rng.integers(0, 31, size=(1, k)))).astype(np.float32)           # elements 2^30 apart
        for name, a in (('gaussian', base), ('row spread', rows), ('element spread', cols)):
            f32, f16 = rel_rms(a, 'f32'), rel_rms(a, 'f16x2')
            assert f16[0] <= f32[0] and f16[1] <= 1.5 * f32[1] and abs(f16[2]) < 2e-8, (name, f32, f16)
        # the kernel against the NumPy restatement of its arithmetic (same scales, same fp16 parts; fp64 sums there, fp32
        # sums on the matrix pipe here): what is left is accumulation order
        npm.set_math('f16x2')
        c = D.empty([m, n])
        D.gemm(m, n, k, D.Mat(D.from_host(rows), k), D.Mat(db, n), D.Mat(c, n))
        want = O.gemm_f16x2(rows, b)
        dev = (c.numpy().astype(np.float64) - want) / np.abs(want).max(axis=1, keepdims=True)
        assert np.sqrt((dev ** 2).mean()) < 1.5e-7 and np.abs(dev).max() < 1.5e-6, (np.sqrt((dev ** 2).mean()), np.abs(dev).max())
        # magnitudes at both ends of the fp32 range: the scales stay normal numbers, nothing overflows in fp16
        ext = base.copy()
        ext[0::2] *= np.float32(2.0 ** 110)
        ext[1::2] *= np.float32(2.0 ** -120)
        D.gemm(m, n, k, D.Mat(D.from_host(ext), k), D.Mat(db, n), D.Mat(c, n))
        ref = ext.astype(np.float64) @ b.astype(np.float64)
        dev = np.abs(c.numpy().astype(np.float64) - ref) / np.abs(ref).max(axis=1, keepdims=True)
        assert np.isfinite(c.numpy()).all() and dev.max() < 2e-6, dev.max()
        # zero rows / columns, exactly
        a = base.copy()
        a[5] = 0.0
        c = D.empty([m, n])
        D.gemm(m, n, k, D.Mat(D.from_host(a), k), D.Mat(db, n), D.Mat(c, n))
        assert np.all(c.numpy()[5] == 0.0) and np.isfinite(c.numpy()).all()
        # non-finite operands: their own row / column, nothing else
        a = base.copy()
        a[7, 100] = np.nan
        a[9, 200] = np.inf
        bb = b.copy()
        bb[300, 11] = np.inf
        D.gemm(m, n, k, D.Mat(D.from_host(a), k), D.Mat(D.from_host(bb), n), D.Mat(c, n))
        out = c.numpy()
        bad = ~np.isfinite(out)
        want = np.zeros_like(bad)
        want[7] = want[9] = True
        want[:, 11] = True
        np.testing.assert_array_equal(bad, want)
        # launches it does not cover run (and report) the bf16 split
        a3 = D.from_host(rng.standard_normal((2, 128, 64)).astype(np.float32))
        b3 = D.from_host(rng.standard_normal((2, 64, 128)).astype(np.float32))
        c3 = D.empty([2, 128, 128])
        D.gemm(128, 128, 64, D.Mat(a3, 64, 128 * 64), D.Mat(b3, 128, 64 * 128), D.Mat(c3, 128, 128 * 128), batch=(2, 1))
        assert npm.last_math() == 'bf16x3'
        np.testing.assert_allclose(c3.numpy(), np.einsum('zmk,zkn->zmn', a3.numpy().astype(np.float64), b3.numpy().astype(np.float64)), rtol=0, atol=2e-4)
    finally:
        npm.set_math('f32')


@pytest.mark.parametrize('sweep_math', ['f32', 'f16x2'])
@pytest.mark.parametrize('seed', range(6))
def test_random_sweep(D, seed, sweep_math):
    """Randomised descriptors against NumPy fp64: layouts, sizes on and off the tile / DMA boundaries, padded and
    unaligned row pitches, two batch dimensions with head-style strides ([b0, rows, b1, cols] storage), every
    epilogue the ABI accepts, split-K requests, the operand column sums; guard words around C and the padding
    columns inside it check that nothing else is written."""
    import np_modeling_amd as npm
    npm.set_math(sweep_math)           # 'f16x2': its kernel where it applies (single products, aligned, K % 16 == 0), the bf16 split or the f32 fallback elsewhere
    rng = np.random.default_rng(1000 + seed)
    sizes = [1, 2, 3, 5, 16, 17, 31, 32, 33, 48, 64, 65, 100, 127, 128, 129, 160, 200, 256, 272, 384]
    guard = 5
    for case in range(40):
        ta, tb = [(False, False), (False, True), (True, False)][rng.integers(3)]
        m, n = int(rng.choice(sizes)), int(rng.choice(sizes))
        k = int(rng.choice(sizes + [512, 1024, 2048 + 16]))
        b0, b1 = (1, 1) if rng.random() < 0.6 else (int(rng.integers(1, 4)), int(rng.integers(1, 4)))
        a_rows, a_cols = (k, m) if ta else (m, k)
        b_rows, b_cols = (n, k) if tb else (k, n)

        def stored(rows, cols):                       # [b0, rows, b1, cols + pad] -> (array, ld, stride0, stride1)
            cp = cols + int(rng.choice([0, 0, 4, 8, 1, 3]))
            arr = rng.standard_normal((b0, rows, b1, cp)).astype(np.float32)
            return arr, b1 * cp, rows * b1 * cp, cp

        a, lda, sa0, sa1 = stored(a_rows, a_cols)
        b, ldb, sb0, sb1 = stored(b_rows, b_cols)
        c_init, ldc, sc0, sc1 = stored(m, n)
        c_init[:] = -777.0
        res = aux = None
        mode = int(rng.integers(7))        # 0 plain 1 bias 2 bias+res 3 bias+relu_save 4 res 5 relu_mask 6 res+relu_mask
        alpha = float(rng.choice([1.0, 1.0, 0.5, -2.0]))
        kwargs = dict(trans_a=ta, trans_b=tb, batch=(b0, b1), alpha=alpha)
        bias = None
        if mode in (1, 2, 3):
            bias = rng.standard_normal(n).astype(np.float32)
            kwargs['bias'] = D.from_host(bias)
        if mode in (2, 4, 6):
            res = rng.standard_normal(c_init.shape).astype(np.float32)
            dres = D.from_host(res)
            kwargs['residual'] = D.Mat(dres, ldc)
        if mode in (3, 5, 6):
            aux = rng.standard_normal(c_init.shape).astype(np.float32)
            daux = D.from_host(aux)
            kwargs['relu_save' if mode == 3 else 'relu_mask'] = D.Mat(daux, ldc)
        if mode in (0, 1, 2, 4) and rng.random() < 0.3:
            kwargs['split_k'] = int(rng.choice([1, 2, 5]))
        sums = which = None
        if (b0, b1) == (1, 1) and ta and not tb and mode in (0, 1, 2, 4) and rng.random() < 0.6:
            which = 'asum_out' if rng.random() < 0.5 else 'bsum_out'
            sums = D.full([m if which == 'asum_out' else n], np.nan)
            kwargs[which] = sums
        flat = np.concatenate([np.full(guard, -777.0, np.float32), c_init.ravel(), np.full(guard, -777.0, np.float32)])
        da, db, dc = D.from_host(a), D.from_host(b), D.from_host(flat)
        D.gemm(m, n, k, D.Mat(da, lda, sa0, sa1), D.Mat(db, ldb, sb0, sb1),
               D.Mat(dc.flat_view(guard, [c_init.size]), ldc, sc0, sc1), **kwargs)
        what = (f'seed {seed} case {case}: ta={ta} tb={tb} m={m} n={n} k={k} batch={(b0, b1)} ld={(lda, ldb, ldc)} '
                f'mode={mode} alpha={alpha} {sorted(kwargs)}')
        got = dc.numpy()
        assert np.all(got[:guard] == -777.0) and np.all(got[-guard:] == -777.0), what
        body = got[guard:-guard].reshape(c_init.shape)
        assert np.all(body[..., n:] == -777.0), what                     # padding columns of every head
        a64, b64 = a[..., :a_cols].astype(np.float64), b[..., :b_cols].astype(np.float64)
        eq = ('zkhm' if ta else 'zmhk') + ',' + ('znhk' if tb else 'zkhn') + '->zmhn'
        ref = alpha * np.einsum(eq, a64, b64)
        if bias is not None:
            ref = ref + bias
        if res is not None:
            ref = ref + res[..., :n]
        if mode == 3:
            assert_close(daux.numpy()[..., :n], ref, tol=3e-6, what=what + ' (saved pre-activation)')
            ref = np.maximum(ref, 0.0)
        if mode in (5, 6):
            ref = np.where(aux[..., :n] >= 0, ref, 0.0)
        assert_close(body[..., :n], ref, tol=3e-6, what=what)
        if sums is not None:
            src = a64 if which == 'asum_out' else b64
            assert_close(sums, src.reshape(k, -1).sum(axis=0), tol=3e-6, what=what + ' (operand sums)')


def test_math_mode_api(D):
    import np_modeling_amd as npm
    from np_modeling_amd import _C
    with pytest.raises(ValueError):
        npm.set_math('tf32')
    with pytest.raises(_C.NpmError):
        _C.check(_C.lib().npm_set_math(7), 'npm_set_math')
    npm.set_math('bf16x3')
    assert npm.get_math() == 'bf16x3'
    npm.set_math('f32')
    assert npm.get_math() == 'f32'


def test_last_math_reports_what_ran(D, math_mode):
    """npm_get_math() is the mode requested; npm_last_math() the one the last launch ran: the split-bf16 modes exist
    on the LDS-DMA pipeline only, every other launch runs (and reports) the exact-f32 MFMA."""
    import np_modeling_amd as npm
    rng = np.random.default_rng(1)

    def run(m, n, k, ld_pad=0, **kw):
        a = D.from_host(rng.standard_normal((m, k + ld_pad)).astype(np.float32))
        b = D.from_host(rng.standard_normal((k, n)).astype(np.float32))
        c = D.empty([m, n])
        D.gemm(m, n, k, D.Mat(a, k + ld_pad), D.Mat(b, n), D.Mat(c, n), **kw)
        return npm.last_math()

    assert npm.get_math() == math_mode
    assert run(256, 256, 64) == math_mode                      # aligned, K % 16 == 0: the DMA pipeline
    assert run(256, 256, 72) == 'f32'                          # K % 16 != 0: register-staged fallback
    assert run(130, 129, 64, ld_pad=1) == 'f32'                # rows not 16-byte aligned
    assert run(256, 256, 64, colsum_out=D.empty([256])) == 'f32'      # column sums in the epilogue
    assert run(256, 256, 64) == math_mode


def test_bias_gradient_arguments(D):
    from np_modeling_amd import _C
    a, b, c, s = D.zeros([64, 64]), D.zeros([64, 64]), D.empty([64, 64]), D.empty([64])
    with pytest.raises(_C.NpmError):          # bsum needs B stored [k, n]
        D.gemm(64, 64, 64, D.Mat(a, 64), D.Mat(b, 64), D.Mat(c, 64), trans_b=True, bsum_out=s)
    with pytest.raises(_C.NpmError):          # asum needs A stored [k, m]
        D.gemm(64, 64, 64, D.Mat(a, 64), D.Mat(b, 64), D.Mat(c, 64), asum_out=s)


@pytest.mark.parametrize('bsz,h,sq,skv,d', [(2, 4, 64, 128, 16), (3, 2, 40, 24, 12), (1, 8, 512, 512, 128), (2, 3, 33, 17, 5)])
def test_softmax_backward_fused_into_the_dp_gemm(D, bsz, h, sq, skv, d):
    """datt = scale * P * (dctx v^T - rowdot(dctx, ctx)) in one GEMM epilogue equals the reference's
    Jacobian softmax backward of dP = dctx v^T (activations.py:32-45) scaled by 1/sqrt(dk)."""
    from oracle import np_oracle as O
    rng = np.random.default_rng(bsz * 7 + sq)
    logits = rng.standard_normal((bsz, h, sq, skv)) * 2
    p = O.softmax_fwd(logits).astype(np.float32)
    v = rng.standard_normal((bsz, skv, h, d)).astype(np.float32)
    dctx = rng.standard_normal((bsz, sq, h, d)).astype(np.float32)
    ctx = np.einsum('bhqk,bkhd->bqhd', p.astype(np.float64), v.astype(np.float64)).astype(np.float32)
    scale = 1.0 / np.sqrt(d)
    dp_, dv_, dd, dc = D.from_host(p), D.from_host(v), D.from_host(dctx), D.from_host(ctx)
    delta = D.attn_rowdot(dd, dc)
    want_delta = np.einsum('bqhd,bqhd->bhq', dctx.astype(np.float64), ctx.astype(np.float64))
    assert_close(delta, want_delta, tol=3e-6)
    datt = D.empty([bsz, h, sq, skv])
    D.gemm(sq, skv, d, D.Mat(dd, h * d, sq * h * d, d), D.Mat(dv_, h * d, skv * h * d, d),
           D.Mat(datt, skv, h * sq * skv, sq * skv), trans_b=True, batch=(bsz, h), alpha=scale,
           softmax_bwd=(D.Mat(dp_, skv), delta))
    dP = np.einsum('bqhd,bkhd->bhqk', dctx.astype(np.float64), v.astype(np.float64))
    want = scale * O.softmax_bwd(p.astype(np.float64), dP, verbatim=(sq * skv * skv < 2e6))
    assert_close(datt, want, tol=1e-5)


@pytest.mark.parametrize('m,n,k', [(128, 128, 16), (300, 256, 64), (1000, 1024, 128), (77, 384, 48)])
def test_rowdot_epilogue(D, m, n, k):
    """NPM_EPI_ROWDOT: C = A B stored as is, and out[n // 128, m] += scale * the row dots of C with X per block of 128 columns --
    the attention backward's row term dctx . ctx per head of size 128, taken where dctx = dy wo is produced (attentions.py:136,
    150-155).  Ragged M (rows beyond M add nothing, the guard rows stay zero), two atomic adds per element onto zeros: bitwise
    reproducible.  Exact fp32 only; any other mode, layout or shape is refused (NPM_E_UNSUPPORTED), never computed differently."""
    from np_modeling_amd import _C
    rng = np.random.default_rng(m + n + k)
    a = rng.standard_normal((m, k)).astype(np.float32)
    b = rng.standard_normal((k, n)).astype(np.float32)
    x = rng.standard_normal((m, n)).astype(np.float32)
    da, db, dx = D.from_host(a), D.from_host(b), D.from_host(x)
    runs = []
    for _ in range(2):
        dc, out = D.full([m, n], np.nan), D.zeros([n // 128, m + 8])
        if _C.current_math() != 'f32':
            with pytest.raises(_C.NpmError) as err:
                D.gemm(m, n, k, D.Mat(da, k), D.Mat(db, n), D.Mat(dc, n), rowdot=(D.Mat(dx, n), out, -0.25))
            assert err.value.code == 10003
            return
        # (out has M + 8 columns per block: the kernel is told M, so its rows are M apart -- a flat [n // 128, m] view)
        flat = out.flat_view(0, [n // 128, m])
        D.gemm(m, n, k, D.Mat(da, k), D.Mat(db, n), D.Mat(dc, n), rowdot=(D.Mat(dx, n), flat, -0.25))
        runs.append((dc.numpy(), flat.numpy(), out.numpy().ravel()[(n // 128) * m:]))
    c = _ref(a, b, False, False)
    assert_close(runs[0][0], c, tol=2e-6)
    want = -0.25 * (c * x.astype(np.float64)).reshape(m, n // 128, 128).sum(axis=2).T
    assert_close(runs[0][1], want, tol=3e-6)
    np.testing.assert_array_equal(runs[0][2], 0.0)                       # nothing written behind the last row
    np.testing.assert_array_equal(runs[0][0], runs[1][0])
    np.testing.assert_array_equal(runs[0][1], runs[1][1])                # two commutative adds per element: same bits every run
    if m == 128:
        for bad in (dict(n=192), dict(trans_b=True), dict(k=24)):
            nn, kk = bad.get('n', n), bad.get('k', k)
            with pytest.raises(_C.NpmError) as err:
                D.gemm(m, nn, kk, D.Mat(da, k), D.Mat(db, n), D.Mat(D.empty([m, n]), n), trans_b=bad.get('trans_b', False),
                       rowdot=(D.Mat(dx, n), D.zeros([2, m]), 1.0))
            assert err.value.code == 10003, bad
