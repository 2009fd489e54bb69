"""TEST INFRASTRUCTURE ONLY -- a host-memory simulator of the libnpm_hip.so C ABI.

It lets the CPU test-suite exercise the *host-side* logic of the product (Layer protocol,
DeviceArray/optimizer contract, gradient bucketing, the data-parallel exchange over gloo)
in a container without a GPU.  "Device pointers" are addresses of NumPy buffers; every
entry point is restated with NumPy / the oracle.  The product never loads this: tests
install it explicitly with ``hostsim.install()`` (which sets ``np_modeling_amd._C._LIB``).
Numerical parity of the real kernels is established on the GPU (tests marked ``gpu``).
"""

from __future__ import annotations

import ctypes as C

import numpy as np

from oracle import np_oracle as O


def _deref(arg):
    return arg._obj if hasattr(arg, '_obj') else arg


def _addr(p):
    if p is None:
        return 0
    if isinstance(p, int):
        return p
    v = getattr(p, 'value', None)
    return int(v) if v else 0


def _vec(ptr, n):
    n = int(n)
    if n == 0:
        return np.zeros(0, dtype=np.float32)
    return np.ctypeslib.as_array((C.c_float * n).from_address(_addr(ptr)))


def _mat(ptr, rows, cols, ld):
    rows, cols, ld = int(rows), int(cols), int(ld)
    if rows == 0 or cols == 0:
        return np.zeros((rows, cols), dtype=np.float32)
    flat = _vec(ptr, (rows - 1) * ld + cols)
    return np.lib.stride_tricks.as_strided(flat, shape=(rows, cols), strides=(4 * ld, 4))


class HostSim:
    def __init__(self):
        self._blocks = {}
        self._err = b''
        self.calls = []

    # ---- runtime ---------------------------------------------------------------------
    def npm_abi_version(self):
        return 2

    def npm_last_error(self):
        return self._err

    def npm_init(self, device):
        return 0

    def npm_shutdown(self):
        return 0

    def npm_sync(self):
        return 0

    def npm_stream(self):
        return 0

    def npm_malloc(self, out, nbytes):
        buf = np.empty(max(int(nbytes), 4) // 4 + 4, dtype=np.float32)
        buf.fill(np.nan)                                   # poison: catches reads of unwritten memory
        addr = buf.ctypes.data
        addr_aligned = (addr + 15) // 16 * 16
        self._blocks[addr_aligned] = buf
        _deref(out).value = addr_aligned
        return 0

    def npm_free(self, ptr):
        self._blocks.pop(_addr(ptr), None)
        return 0

    def npm_pool_stats(self, a, b):
        _deref(a).value = sum(v.nbytes for v in self._blocks.values())
        _deref(b).value = _deref(a).value
        return 0

    def npm_pool_trim(self):
        return 0

    def npm_h2d(self, dst, src, nbytes):
        C.memmove(_addr(dst), _addr(src), int(nbytes))
        return 0

    npm_d2h = npm_h2d
    npm_d2d = npm_h2d

    def npm_fill_f32(self, dst, value, n):
        _vec(dst, n)[:] = value
        return 0

    def npm_event_create(self, out):
        _deref(out).value = 1
        return 0

    def npm_event_destroy(self, ev):
        return 0

    npm_event_record = npm_event_sync = npm_event_destroy

    def npm_event_elapsed_ms(self, a, b, out):
        _deref(out).value = 0.0
        return 0

    # ---- GEMM ----------------------------------------------------------------------------
    def npm_sgemm(self, gref):
        g = _deref(gref)
        self.calls.append('npm_sgemm')
        sums = np.zeros((g.batch1, g.n), dtype=np.float64)
        for z0 in range(g.batch0):
            for z1 in range(g.batch1):
                oa = 4 * (z0 * g.stride_a0 + z1 * g.stride_a1)
                ob = 4 * (z0 * g.stride_b0 + z1 * g.stride_b1)
                oc = 4 * (z0 * g.stride_c0 + z1 * g.stride_c1)
                a = _mat(g.a + oa, g.k, g.m, g.lda).T if g.trans_a else _mat(g.a + oa, g.m, g.k, g.lda)
                b = _mat(g.b + ob, g.n, g.k, g.ldb).T if g.trans_b else _mat(g.b + ob, g.k, g.n, g.ldb)
                v = np.float64(g.alpha) * (a.astype(np.float64) @ b.astype(np.float64))
                if g.epilogue & 32:
                    z = z0 * g.batch1 + z1
                    v = np.float64(g.alpha) * _mat(g.aux + oc, g.m, g.n, g.ldaux) * (
                        a.astype(np.float64) @ b.astype(np.float64) - _vec(g.rowvec + 4 * z * g.m, g.m).astype(np.float64)[:, None])
                if g.epilogue & 1:
                    v = v + _vec(g.bias, g.n).astype(np.float64)
                if g.epilogue & 2:
                    v = v + _mat(g.residual + oc, g.m, g.n, g.ldr).astype(np.float64)
                if g.epilogue & 4:
                    _mat(g.aux + oc, g.m, g.n, g.ldaux)[:] = v
                    v = np.maximum(v, 0.0)
                if g.epilogue & 8:
                    v = np.where(_mat(g.aux + oc, g.m, g.n, g.ldaux) >= 0, v, 0.0)
                if g.epilogue & 16:
                    v = np.maximum(v, 0.0)
                _mat(g.c + oc, g.m, g.n, g.ldc)[:] = v
                if g.epilogue & 128:                     # NPM_EPI_ROWDOT: per block of 128 columns, row dots of C with aux, added to zeros
                    if g.epilogue != 128 or g.trans_a or g.trans_b or g.n % 128 or g.k % 16 or g.batch0 * g.batch1 != 1:
                        return 10003
                    prod = _mat(g.c + oc, g.m, g.n, g.ldc).astype(np.float64) * _mat(g.aux + oc, g.m, g.n, g.ldaux).astype(np.float64)
                    out = _vec(g.rowdot, (g.n // 128) * g.m).reshape(g.n // 128, g.m)
                    assert not np.isnan(out).any() and (out == 0).all(), 'NPM_EPI_ROWDOT: rowdot must be zero on entry'
                    out += (np.float64(g.rowdot_scale) * prod.reshape(g.m, g.n // 128, 128).sum(axis=2).T).astype(np.float32)
                sums[z1] += _mat(g.c + oc, g.m, g.n, g.ldc).astype(np.float64).sum(axis=0)
        if g.colsum:
            _vec(g.colsum, g.batch1 * g.n)[:] = sums.ravel()
        if g.bsum:
            assert not g.trans_b and g.batch0 == 1 and g.batch1 == 1
            _vec(g.bsum, g.n)[:] = _mat(g.b, g.k, g.n, g.ldb).astype(np.float64).sum(axis=0)
        if g.asum:
            assert g.trans_a and g.batch0 == 1 and g.batch1 == 1 and not g.bsum
            _vec(g.asum, g.m)[:] = _mat(g.a, g.k, g.m, g.lda).astype(np.float64).sum(axis=0)
        return 0

    # ---- elementwise ------------------------------------------------------------------------
    def npm_relu_fwd(self, x, y, n):
        _vec(y, n)[:] = np.maximum(_vec(x, n), 0)
        return 0

    def npm_relu_bwd(self, x, dy, dx, n):
        _vec(dx, n)[:] = np.where(_vec(x, n) >= 0, _vec(dy, n), 0)
        return 0

    def npm_add(self, a, b, out, n):
        self.calls.append('npm_add')
        _vec(out, n)[:] = _vec(a, n) + _vec(b, n)
        return 0

    def npm_add3(self, a, b, c, out, n):
        _vec(out, n)[:] = _vec(a, n) + _vec(b, n) + _vec(c, n)
        return 0

    def npm_axpy(self, y, x, alpha, n):
        self.calls.append('npm_axpy')
        _vec(y, n)[:] = _vec(y, n) + np.float32(alpha) * _vec(x, n)
        return 0

    def npm_scale(self, x, y, alpha, n):
        _vec(y, n)[:] = np.float32(alpha) * _vec(x, n)
        return 0

    def npm_colsum(self, x, out, rows, cols, ld):
        _vec(out, cols)[:] = _mat(x, rows, cols, ld).astype(np.float64).sum(axis=0)
        return 0

    def npm_relu_bwd_colsum(self, x, dy, dx, colsum, rows, cols):
        n = rows * cols
        g = np.where(_vec(x, n) >= 0, _vec(dy, n), 0)
        _vec(dx, n)[:] = g
        _vec(colsum, cols)[:] = g.reshape(rows, cols).astype(np.float64).sum(axis=0)
        return 0

    # ---- row kernels ---------------------------------------------------------------------------
    def npm_attn_rowdot(self, a, b, out, batch, seq, heads, dim):
        n = int(batch * seq * heads * dim)
        prod = (_vec(a, n).astype(np.float64) * _vec(b, n)).reshape(batch, seq, heads, dim).sum(axis=-1)
        _vec(out, batch * seq * heads)[:] = prod.transpose(0, 2, 1).ravel()
        return 0

    def npm_softmax_fwd(self, x, y, rows, n, scale):
        _mat(y, rows, n, n)[:] = O.softmax_fwd(np.float64(scale) * _mat(x, rows, n, n).astype(np.float64))
        return 0

    def npm_softmax_bwd(self, y, dy, dx, rows, n, scale):
        _mat(dx, rows, n, n)[:] = scale * O.softmax_bwd(_mat(y, rows, n, n), _mat(dy, rows, n, n))
        return 0

    def npm_layernorm_fwd(self, x, gamma, beta, eps, rows, d, z, mean, rstd):
        xv = _mat(x, rows, d, d).astype(np.float64)
        out, (mu, var, _) = O.layernorm_fwd(xv, _vec(gamma, d).astype(np.float64), _vec(beta, d).astype(np.float64), eps)
        _mat(z, rows, d, d)[:] = out
        _vec(mean, rows)[:] = mu[:, 0]
        _vec(rstd, rows)[:] = 1.0 / np.sqrt(var[:, 0] + eps)
        return 0

    def npm_layernorm_bwd(self, dz, x, mean, rstd, gamma, residual, rows, d, dx, dgamma, dbeta):
        xv = _mat(x, rows, d, d).astype(np.float64)
        mu = _vec(mean, rows).astype(np.float64)[:, None]
        rs = _vec(rstd, rows).astype(np.float64)[:, None]
        yhat = (xv - mu) * rs
        dzv = _mat(dz, rows, d, d).astype(np.float64)
        g = dzv * _vec(gamma, d).astype(np.float64)
        out = rs * (g - g.mean(axis=1, keepdims=True) - yhat * (g * yhat).mean(axis=1, keepdims=True))
        if _addr(residual):
            out = out + _mat(residual, rows, d, d)
        dg, db = (dzv * yhat).sum(axis=0), dzv.sum(axis=0)
        _mat(dx, rows, d, d)[:] = out
        _vec(dgamma, d)[:] = dg
        _vec(dbeta, d)[:] = db
        return 0

    def _dropped(self, x, mask, keep, rows, d):
        mk = np.ctypeslib.as_array((C.c_ubyte * int(rows * d)).from_address(_addr(mask))).reshape(rows, d) != 0
        return mk, np.where(mk, _mat(x, rows, d, d) / np.float32(keep), np.float32(0)).astype(np.float32)

    def npm_layernorm_dropout_fwd(self, x, mask, keep, gamma, beta, eps, rows, d, z, mean, rstd):
        """include/npm_hip.h: npm_mask_scale, then npm_layernorm_fwd, without the tensor in between."""
        self.calls.append('npm_layernorm_dropout_fwd')
        assert d % 4 == 0 and d <= 4096 and _addr(mask) % 4 == 0
        _, xd = self._dropped(x, mask, keep, rows, d)
        return self.npm_layernorm_fwd(xd.ctypes.data, gamma, beta, eps, rows, d, z, mean, rstd)

    def npm_layernorm_dropout_bwd(self, dz, x, mask, keep, mean, rstd, gamma, residual, rows, d, dx, dgamma, dbeta):
        self.calls.append('npm_layernorm_dropout_bwd')
        assert d % 4 == 0 and d <= 4096 and _addr(mask) % 4 == 0
        mk, xd = self._dropped(x, mask, keep, rows, d)
        inner = np.empty([rows, d], dtype=np.float32)
        rc = self.npm_layernorm_bwd(dz, xd.ctypes.data, mean, rstd, gamma, 0, rows, d, inner.ctypes.data, dgamma, dbeta)
        out = np.where(mk, inner / np.float32(keep), np.float32(0))
        if _addr(residual):
            out = out + _mat(residual, rows, d, d)
        _mat(dx, rows, d, d)[:] = out
        return rc

    # ---- fused attention core ------------------------------------------------------------------------
    def npm_mha_mask_summary(self, mask, sb, sh, sq_stride, nb, nh, seq_q, seq_kv, out):
        """byte (qt, kb) of plane (b, h): bit w = some allowed position in queries 32 qt.. x keys 128 kb + 16 w.. (include/npm_hip.h)"""
        self.calls.append('npm_mha_mask_summary')
        nqt, nkb = (seq_q + 31) // 32, (seq_kv + 127) // 128
        extent = (nb - 1) * sb + (nh - 1) * sh + (seq_q - 1) * sq_stride + seq_kv
        raw = np.ctypeslib.as_array((C.c_ubyte * int(extent)).from_address(_addr(mask)))
        planes = np.lib.stride_tricks.as_strided(raw, shape=(nb, nh, seq_q, seq_kv), strides=(sb, sh, sq_stride, 1))
        padded = np.zeros((nb, nh, nqt * 32, nkb * 128), dtype=bool)
        padded[:, :, :seq_q, :seq_kv] = planes != 0
        inside = np.zeros_like(padded)
        inside[:, :, :seq_q, :seq_kv] = True
        split = lambda a: a.reshape(nb, nh, nqt, 32, nkb, 8, 16)
        any16 = split(padded).any(axis=(3, 6))                                          # [nb, nh, nqt, nkb, 8]
        any16[~(planes != 0).any(axis=3).all(axis=2)] = True     # a plane with a key-less query row is never skipped (npm_hip.h)
        all16 = split(padded | ~inside).all(axis=(3, 6)) & split(inside).any(axis=(3, 6))
        pack = lambda bits: (bits << np.arange(8)).sum(axis=-1).astype(np.uint8).reshape(-1)
        both = np.concatenate([pack(any16), pack(all16)])
        np.ctypeslib.as_array((C.c_ubyte * both.size).from_address(_addr(out)))[:] = both
        return 0

    def npm_mha_core_supported(self, head_dim):
        return int(head_dim in (16, 32, 64, 128))

    @staticmethod
    def _heads(ptr, pitch, b, s, h, d):
        """[B, S, H, D] view of a pitched operand."""
        flat = _vec(ptr, ((b * s - 1) * pitch + h * d) if b * s else 0)
        return np.lib.stride_tricks.as_strided(flat, shape=(b, s, h, d), strides=(4 * s * pitch, 4 * pitch, 4 * d, 4))

    def _core_mask(self, c):
        if not c.mask:
            return None
        b, h, sq, skv = c.batch, c.heads, c.seq_q, c.seq_kv
        extent = (b - 1) * c.mask_stride_b + (h - 1) * c.mask_stride_h + (sq - 1) * c.mask_stride_q + skv
        raw = np.ctypeslib.as_array((C.c_ubyte * int(extent)).from_address(_addr(c.mask)))
        return np.lib.stride_tricks.as_strided(raw, shape=(b, h, sq, skv),
                                               strides=(c.mask_stride_b, c.mask_stride_h, c.mask_stride_q, 1)) != 0

    def npm_mha_core_fwd(self, cref):
        c = _deref(cref)
        self.calls.append('npm_mha_core_fwd')
        if c.head_dim not in (16, 32, 64, 128):
            return 10003
        b, h, sq, skv, d = c.batch, c.heads, c.seq_q, c.seq_kv, c.head_dim
        q, k, v = (self._heads(ptr, pitch, b, s, h, d).astype(np.float64)
                   for ptr, pitch, s in ((c.q, c.q_pitch, sq), (c.k, c.k_pitch, skv), (c.v, c.v_pitch, skv)))
        mask = self._core_mask(c)
        with np.errstate(invalid='ignore'):
            ctx, lse, _ = O.attention_core_fwd(q, k, v, float(c.scale), mask)
        self._heads(c.ctx, c.ctx_pitch, b, sq, h, d)[:] = ctx
        _vec(c.lse, b * h * sq)[:] = lse.ravel()
        if c.scores:
            raw = np.einsum('bqhd,bkhd->bhqk', q, k)
            _vec(c.scores, b * h * sq * skv)[:] = (raw if mask is None else np.where(mask, raw, -np.inf)).ravel()
        return 0

    def npm_mha_core_bwd(self, cref):
        c = _deref(cref)
        self.calls.append('npm_mha_core_bwd')
        b, h, sq, skv, d = c.batch, c.heads, c.seq_q, c.seq_kv, c.head_dim
        q, k, v = (self._heads(ptr, pitch, b, s, h, d).astype(np.float64)
                   for ptr, pitch, s in ((c.q, c.q_pitch, sq), (c.k, c.k_pitch, skv), (c.v, c.v_pitch, skv)))
        dctx = self._heads(c.dctx, c.dctx_pitch, b, sq, h, d).astype(np.float64)
        lse = _vec(c.lse, b * h * sq).astype(np.float64).reshape(b, h, sq)
        scaled = float(c.scale) * np.einsum('bqhd,bkhd->bhqk', q, k)
        mask = self._core_mask(c)
        if mask is not None:
            scaled = np.where(mask, scaled, -np.inf)
        probs = np.exp(scaled - lse[..., None])             # from the saved log-sum-exp, as the kernel does
        if c.neg_delta:                                     # the caller's row terms: what the kernel would have computed itself
            ctx = self._heads(c.ctx, c.ctx_pitch, b, sq, h, d).astype(np.float64)
            want = -float(c.scale) * np.einsum('bqhd,bqhd->bhq', dctx, ctx)
            extent = (b - 1) * c.neg_delta_stride_b + (h - 1) * c.neg_delta_stride_h + sq
            got = np.lib.stride_tricks.as_strided(_vec(c.neg_delta, extent), shape=(b, h, sq),
                                                  strides=(4 * c.neg_delta_stride_b, 4 * c.neg_delta_stride_h, 4))
            np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-5 * (np.abs(want).max() + 1e-30), err_msg='neg_delta')
        dq, dk, dv = O.attention_core_bwd(q, k, v, probs, dctx, float(c.scale))
        self._heads(c.dq, c.dq_pitch, b, sq, h, d)[:] = dq
        self._heads(c.dk, c.dk_pitch, b, skv, h, d)[:] = dk
        self._heads(c.dv, c.dv_pitch, b, skv, h, d)[:] = dv
        return 0

    # ---- conv -----------------------------------------------------------------------------------
    def npm_conv2d_fwd(self, cref):
        c = _deref(cref)
        x = _vec(c.x, c.n * c.h * c.w * c.c_in).reshape(c.n, c.h, c.w, c.c_in)
        f = _vec(c.filt, c.ksize * c.ksize * c.c_in * c.c_out).reshape(c.ksize, c.ksize, c.c_in, c.c_out)
        v = O.conv2d_fwd(x, f)
        if c.bias:
            v = v + _vec(c.bias, c.c_out)
        if c.relu:
            if c.pre:
                _vec(c.pre, v.size)[:] = v.ravel()
            v = np.maximum(v, 0)
        _vec(c.y, v.size)[:] = v.ravel()
        return 0

    def npm_conv2d_bwd_x(self, dy, filt, dx, n, h, w, c0, c1, k):
        g = _vec(dy, n * h * w * c1).reshape(n, h, w, c1)
        f = _vec(filt, k * k * c0 * c1).reshape(k, k, c0, c1)
        _vec(dx, n * h * w * c0)[:] = O.conv2d_grad_x(g, f).ravel()
        return 0

    def npm_conv2d_bwd_w(self, dy, x, dw, n, h, w, c0, c1, k):
        g = _vec(dy, n * h * w * c1).reshape(n, h, w, c1)
        xv = _vec(x, n * h * w * c0).reshape(n, h, w, c0)
        _vec(dw, k * k * c0 * c1)[:] = O.conv2d_grad_w(g, xv, k).ravel()
        return 0

    def npm_conv2d_bwd_w_relu(self, dy, pre, x, g, dw, db, n, h, w, c0, c1, k):
        size = n * h * w * c1
        gv = np.where(_vec(pre, size) >= 0, _vec(dy, size), 0).astype(np.float32)
        _vec(g, size)[:] = gv
        _vec(db, c1)[:] = gv.reshape(-1, c1).sum(axis=0)
        return self.npm_conv2d_bwd_w(g, x, dw, n, h, w, c0, c1, k)

    # ---- around the path -------------------------------------------------------------------------
    def npm_fill_f64(self, dst, value, n):
        np.ctypeslib.as_array((C.c_double * int(n)).from_address(_addr(dst)))[:] = value
        return 0

    def npm_adam_step(self, var, grad, m, v, n, lr, beta1, beta2, eps, step):
        self.calls.append('npm_adam_step')
        n = int(n)
        mm = np.ctypeslib.as_array((C.c_double * n).from_address(_addr(m)))
        vv = np.ctypeslib.as_array((C.c_double * n).from_address(_addr(v)))
        g = _vec(grad, n).astype(np.float64)
        mm[:] = beta1 * mm + (1 - beta1) * g
        vv[:] = beta2 * vv + (1 - beta2) * g ** 2
        upd = lr * ((mm / (1 - beta1 ** step)) / np.sqrt(vv / (1 - beta2 ** step) + eps))
        w = _vec(var, n)
        w -= upd                                            # fp64 loop, one rounding (as NumPy's f32 -= f64)
        return 0

    def npm_mse_fwd(self, y, t, n, out):
        d = _vec(y, n).astype(np.float64) - _vec(t, n)
        _deref(out).value = float((d * d).sum() / int(n))
        return 0

    def npm_mse_bwd(self, y, t, dy, n):
        _vec(dy, n)[:] = np.float32(2.0 / int(n)) * (_vec(y, n) - _vec(t, n))
        return 0

    def npm_xent_fwd(self, y, t, n, out):
        _deref(out).value = float(-(_vec(t, n).astype(np.float64) * np.log(_vec(y, n).astype(np.float64))).sum())
        return 0

    def npm_xent_bwd(self, y, t, dy, n):
        _vec(dy, n)[:] = -_vec(t, n) / _vec(y, n)
        return 0

    def npm_mask_scale(self, x, mask, y, n, keep):
        self.calls.append('npm_mask_scale')
        mk = np.ctypeslib.as_array((C.c_ubyte * int(n)).from_address(_addr(mask)))
        _vec(y, n)[:] = np.where(mk != 0, _vec(x, n) / np.float32(keep), 0)
        return 0

    def npm_dropout_philox(self, x, y, mask, n, keep, seed, offset):
        self.calls.append('npm_dropout_philox')
        keep_mask = O.dropout_philox_mask(int(n), float(keep), int(seed), int(offset))
        np.ctypeslib.as_array((C.c_ubyte * int(n)).from_address(_addr(mask)))[:] = keep_mask
        if _addr(y):                               # x = y = NULL: the mask only
            _vec(y, n)[:] = np.where(keep_mask, _vec(x, n) / np.float32(keep), 0)
        return 0

    def npm_set_math(self, mode):
        self.math = int(mode)
        return 0

    def npm_get_math(self):
        return getattr(self, 'math', 0)

    def npm_last_math(self):
        return 0

    def npm_set_tuning(self, knob, value):
        return 0

    def npm_last_attn_kernel(self):
        last = [c for c in self.calls if c.startswith('npm_mha_core_')]
        return (('hostsim ' + last[-1]) if last else '').encode()


def install():
    """Install a fresh simulator as the product's library handle; returns it."""
    from np_modeling_amd import _C
    sim = HostSim()
    _C._LIB = sim
    _C._DEVICE = 0
    return sim


def uninstall():
    from np_modeling_amd import _C
    _C._LIB = None
    _C._DEVICE = None
