"""Multi-head attention on the fp32 MFMA GEMM (reference layers/attentions.py:11-199).

Every einsum of the reference is a GEMM view, addressed in place -- no transposes are
materialised:

* projections ``'...ab,cdb->...acd'`` (attentions.py:88-100): [B*S, F] x w[H*D, F]^T, bias in
  the epilogue; q/k/v stay in the reference's [B, S, H, D] layout;
* ``QK^T`` / ``PV`` and their four gradients (attentions.py:103-112,146-162): GEMMs batched
  over (B, H) whose operands are head slices of [B, S, H, D] (row pitch H*D, batch strides
  (S*H*D, D));
* the context ("values") is written as [B, Sq, H, Dv] so that the output projection
  ``'...abc,...dac->...bd'`` (attentions.py:116) is a plain [B*Sq, H*Dv] x wo[F, H*Dv]^T GEMM
  (the reference's own cache is head-major [B, H, Sq, Dv]; this one is private state);
* softmax forward/backward with the 1/sqrt(Dk) scaling fused (attentions.py:104,150-155).

Where the head size allows (Dk == Dv in {16, 32, 64, 128}, exact-fp32 math) the score / softmax / context chain and
its gradient run as ONE kernel each (``npm_mha_core_fwd`` / ``npm_mha_core_bwd``, csrc/npm_attn.hip): the
[B, H, Sq, Skv] probabilities are never stored, only a log-sum-exp per query row.  The GEMM composition above stays
as the path for other head sizes and for the split-bf16 math modes.

Masks.  The reference tests ``if mask:`` (attentions.py:84,106), which raises ValueError for any array of more than
one element, and its backward raises NotImplementedError (attentions.py:152-153): masked attention is unreachable
there.  This layer implements the evident intent instead: ``mask`` is a boolean array broadcastable to
[B, H, Sq, Skv]; excluded positions get ``-inf`` before the softmax (``np.where(mask, scaled, -inf)``) and
probability exactly 0, in forward and backward; a query row with no position left yields NaN, as that ``np.where``
followed by the reference's softmax would.  A ``device.AttnMask`` made once (``AttnMask(mask, B, H, Sq, Skv)``) may be
passed instead of the array: its bytes and its tile summary (which lets the kernels skip tiles without an allowed
position) then stay on the device between steps.

Parameter layouts are the reference's: wq/wk [H, Dk, H*Dk], wv [H, Dv, H*Dv], wo [H*Dk, H, Dv],
bq/bk [H, Dk], bv [H, Dv], bo [H*Dk] (attentions.py:46-65), drawn in that order.
"""

from __future__ import annotations

import math
from typing import Optional

import numpy as np

from np_modeling_amd import device as D
from np_modeling_amd import parallel
from np_modeling_amd.device import Mat
from np_modeling_amd.layers import activations, layer

_PARAMS = ('_wq', '_wk', '_wv', '_wo', '_bq', '_bk', '_bv', '_bo')


class MultiHeadAttention(layer.StatefulLayer):
    def __init__(self, num_heads: int, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._num_heads = num_heads
        self._softmax = activations.Softmax()

    def initialize(self, query, key=None, value=None, *args, **kwargs) -> None:
        # query [B, Sq, H*Dk]; key [B, Skv, H*Dk]; value [B, Skv, H*Dv]
        if key is None:
            key = query
        if value is None:
            value = key
        assert query.shape[0] == key.shape[0]
        assert query.shape[2] == key.shape[2]
        assert query.shape[0] == value.shape[0]
        assert key.shape[1] == value.shape[1]
        self._seq_len_q = query.shape[1]
        self._seq_len_kv = key.shape[1]
        h = self._num_heads
        assert key.shape[2] % h == 0
        self._key_dim = dk = key.shape[2] // h
        assert value.shape[2] % h == 0
        self._value_dim = dv = value.shape[2] // h
        # Draw order wq, wk, wv, wo, bq, bk, bv, bo (attentions.py:46-65).  When the three in-projections
        # have one shape they are stored back to back (views of one [3, H, D, F] buffer), so that
        # self-attention can run them as ONE GEMM; rebinding any of them (weight binders do) just
        # falls back to three GEMMs.
        draws = [self._initializer([h, dk, h * dk]), self._initializer([h, dk, h * dk]),
                 self._initializer([h, dv, h * dv])]
        wo = self._initializer([h * dk, h, dv])
        bias_draws = [self._initializer([h, dk]), self._initializer([h, dk]), self._initializer([h, dv])]
        bo = self._initializer([h * dk])
        if dk == dv:
            packed_w, packed_b = D.empty([3, h, dk, h * dk]), D.empty([3, h, dk])
            for i, (name, bname) in enumerate((('_wq', '_bq'), ('_wk', '_bk'), ('_wv', '_bv'))):
                setattr(self, name, packed_w.flat_view(i * h * dk * h * dk, [h, dk, h * dk]).set(draws[i]))
                setattr(self, bname, packed_b.flat_view(i * h * dk, [h, dk]).set(bias_draws[i]))
        else:
            self._wq, self._wk, self._wv = (D.as_device(a) for a in draws)
            self._bq, self._bk, self._bv = (D.as_device(a) for a in bias_draws)
        self._wo = D.as_device(wo)
        self._bo = D.as_device(bo)
        # one arena in the order backward produces the gradients (dbo, dwo, the in-projection weights, their biases)
        self._pack_parameters(self._segments())

    def _params_adjacent(self) -> bool:
        """wq/wk/wv (and bq/bk/bv) still back to back in memory?  Checked at EVERY use: parameters may be
        rebound between a forward and its backward (weight binders, tests)."""
        w = [self._param(p) for p in ('_wq', '_wk', '_wv')]
        b = [self._param(p) for p in ('_bq', '_bk', '_bv')]
        return (w[0].shape == w[1].shape == w[2].shape and
                w[1].ptr == w[0].ptr + w[0].nbytes and w[2].ptr == w[1].ptr + w[1].nbytes and
                b[1].ptr == b[0].ptr + b[0].nbytes and b[2].ptr == b[1].ptr + b[1].nbytes)

    def _packed_qkv(self, query, key, value) -> bool:
        """Self-attention with the in-projection parameters adjacent in memory: one GEMM makes q, k and v."""
        if not (query is key and key is value) or self._key_dim != self._value_dim or not D.PACK_QKV:
            return False
        return self._params_adjacent()

    def _numel(self) -> int:
        return sum(self._param(p).size for p in _PARAMS) + 4 * len(_PARAMS)

    def _segments(self):
        """Parameters in the order ``_backward_impl`` produces their gradients (device.ParamArena segments)."""
        return [[(self, '_bo')], [(self, '_wo')], [(self, '_wq'), (self, '_wk'), (self, '_wv')],
                [(self, '_bq'), (self, '_bk'), (self, '_bv')]]

    # -- forward -------------------------------------------------------------------------
    def forward(self, query, key=None, value=None, mask=None):
        query = D.as_device(query)
        key = query if key is None else D.as_device(key)
        value = key if value is None else D.as_device(value)
        return self._forward_impl(query, key, value, mask=mask)

    def _forward_impl(self, query, key, value, residual: Optional[D.DeviceArray] = None, mask=None):
        h, dk, dv = self._num_heads, self._key_dim, self._value_dim
        b, sq, f = query.shape
        skv = key.shape[1]
        fv = value.shape[2]
        assert f == h * dk and key.shape == (b, skv, f) and value.shape[:2] == (b, skv) and fv == h * dv
        wq, wk, wv, wo = (self._param(p) for p in ('_wq', '_wk', '_wv', '_wo'))
        bq, bk, bv, bo = (self._param(p) for p in ('_bq', '_bk', '_bv', '_bo'))
        self._query, self._key, self._value = query, key, value
        if mask is not None and not isinstance(mask, D.AttnMask) and np.ndim(mask) == 0 and not mask:   # `if mask:` false (attentions.py:84,106)
            mask = None
        if isinstance(mask, D.AttnMask):                               # made once by the caller (its bytes and tile summary stay
            assert mask.dims == (b, h, sq, skv), f'AttnMask made for {mask.dims}, used with {(b, h, sq, skv)}'   # on the device)
            self._mask = mask
        else:
            self._mask = None if mask is None else D.AttnMask(mask, b, h, sq, skv)
        core = D.mha_core_supported(dk, dv, any_math=self._mask is not None)
        if self._mask is not None and not core:
            raise NotImplementedError('masked attention needs head sizes Dk == Dv in {16, 32, 64, 128} (fused kernels)')
        self._core = core

        # in-projections: [rows, F] x w[H*D, F]^T + b.  q/k/v are [B, S, H, D] head slices addressed through
        # (array, element offset, row pitch): separate tensors, or thirds of one packed [B, S, 3, H, D].
        packed = self._packed_qkv(query, key, value)
        self._packed = packed
        if packed:
            qkv = D.empty([b, sq, 3, h, dk])
            D.gemm(b * sq, 3 * f, f, Mat(query, f), Mat(wq, f), Mat(qkv, 3 * f), trans_b=True, bias=bq)
            pitch = 3 * f
            # k and v are the same buffer entered f and 2f elements later (row pitch 3f); the views span to the end
            q, k, v = qkv, qkv.flat_view(f, [qkv.size - f]), qkv.flat_view(2 * f, [qkv.size - 2 * f])
        else:
            pitch = None
            q = D.empty([b, sq, h, dk])
            k = D.empty([b, skv, h, dk])
            v = D.empty([b, skv, h, dv])
            D.gemm(b * sq, h * dk, f, Mat(query, f), Mat(wq, f), Mat(q, h * dk), trans_b=True, bias=bq)
            D.gemm(b * skv, h * dk, f, Mat(key, f), Mat(wk, f), Mat(k, h * dk), trans_b=True, bias=bk)
            D.gemm(b * skv, h * dv, fv, Mat(value, fv), Mat(wv, fv), Mat(v, h * dv), trans_b=True, bias=bv)
        self._q, self._k, self._v = q, k, v
        pq, pk, pv = pitch or h * dk, pitch or h * dk, pitch or h * dv       # row pitches of q, k, v
        self._pitches = (pq, pk, pv)

        self._scale = 1.0 / math.sqrt(dk)
        if core:
            # scores, softmax and context in one kernel; what the backward needs is the log-sum-exp per row
            ctx, self._lse, self._raw_scores = D.mha_core_fwd(
                Mat(q, pq), Mat(k, pk), Mat(v, pv), (b, h, sq, skv, dk), self._scale, self._mask,
                save_scores=D.attn_save_scores(dk))
            self._softmax._y = self._attention_scores = None
        else:
            # attention[b, h] = q_h k_h^T ; scores = softmax(attention / sqrt(dk))
            scores = D.empty([b, h, sq, skv])
            D.gemm(sq, skv, dk, Mat(q, pq, sq * pq, dk), Mat(k, pk, skv * pk, dk),
                   Mat(scores, skv, h * sq * skv, sq * skv), trans_b=True, batch=(b, h))
            D.softmax_fwd(scores, self._scale, out=scores)
            self._softmax._y = scores
            self._attention_scores = scores
            # context[b, :, h, :] = scores[b, h] v_h      -> [B, Sq, H, Dv]
            ctx = D.empty([b, sq, h, dv])
            D.gemm(sq, dv, skv, Mat(scores, skv, h * sq * skv, sq * skv), Mat(v, pv, skv * pv, dv),
                   Mat(ctx, h * dv, sq * h * dv, dv), batch=(b, h))
        self._context = ctx

        # output projection: [B*Sq, H*Dv] x wo[F, H*Dv]^T + bo (+ skip connection)
        out = D.empty([b, sq, f])
        D.gemm(b * sq, f, h * dv, Mat(ctx, h * dv), Mat(wo, h * dv), Mat(out, f), trans_b=True, bias=bo,
               residual=None if residual is None else Mat(residual, f))
        return out

    # -- backward --------------------------------------------------------------------------
    def backward(self, dy, optimizer_):
        with parallel.grad_scope(self._numel(), self._arena) as scope:
            return self._backward_impl(D.as_device(dy), optimizer_, scope)

    def _backward_impl(self, dy, optimizer_, scope, *, sum_inputs: bool = False, sum_kv: bool = False,
                       residual: Optional[D.DeviceArray] = None):
        """Returns (dquery, dkey, dvalue) (attentions.py:199), or -- for a composite that feeds
        one tensor as query, key and value -- their sum (+ residual) accumulated in the GEMM
        epilogues when ``sum_inputs`` is set (reference layers/transformer.py:84-85).  ``sum_kv``
        (cross-attention over one kv tensor): returns (dquery (+ residual), dkey + dvalue), the second
        accumulated in its GEMMs' epilogues (transformer.py:186)."""
        h, dk, dv = self._num_heads, self._key_dim, self._value_dim
        query, key, value = self._query, self._key, self._value
        q, k, v, scores, ctx = self._q, self._k, self._v, self._attention_scores, self._context
        b, sq, f = dy.shape
        skv, fv = key.shape[1], value.shape[2]
        wq, wk, wv, wo = (self._param(p) for p in ('_wq', '_wk', '_wv', '_wo'))
        m_q, m_kv = b * sq, b * skv

        # output projection (attentions.py:129-136)
        # dbo = sum over (batch, position) of dy: taken from the dy tiles of the weight-gradient GEMM
        dbo = scope.take([f], owner=(self, '_bo'))
        dwo = scope.take(wo.shape, owner=(self, '_wo'))
        D.gemm(f, h * dv, m_q, Mat(dy, f), Mat(ctx, h * dv), Mat(dwo, h * dv), trans_a=True, asum_out=dbo)   # dy^T ctx
        dctx = D.empty([b, sq, h, dv])
        # dy wo.  At head size 128 a column block of that GEMM's tiles IS a head: its epilogue also takes the row terms
        # -scale * (dctx_i . ctx_i) the fused backward needs (the Jacobian-vector product of Softmax.backward,
        # activations.py:42-45 / attentions.py:150-155), [H, B * Sq], instead of a pass over dctx and ctx behind it.
        neg_delta = None
        if self._core and dv == 128 and D.ATTN_ROWDOT and D.attn_save_scores(dk) and f % 16 == 0 and sq % 4 == 0 \
                and D._C.current_math() == 'f32':
            rows = D.zeros([h, m_q])
            try:
                D.gemm(m_q, h * dv, f, Mat(dy, f), Mat(wo, h * dv), Mat(dctx, h * dv), rowdot=(Mat(ctx, h * dv), rows, -self._scale))
                neg_delta = (rows, sq, m_q)
            except D._C.NpmError as err:                   # operands the row-dot instance does not take (alignment): the plain product
                if err.code != 10003:
                    raise
        if neg_delta is None:
            D.gemm(m_q, h * dv, f, Mat(dy, f), Mat(wo, h * dv), Mat(dctx, h * dv))

        packed = self._packed
        pq, pk, pv = self._pitches
        # dbq/dbk/dbv = sum over (batch, position) of dq/dk/dv (attentions.py:186-188): taken from the dq/dk/dv
        # tiles of the in-projection weight-gradient GEMMs below
        if packed:      # gradients of the packed parameters and of q/k/v live in packed buffers too
            dw_all, db_all = scope.take([3, h, dk, f], owner=(self, '_wq')), scope.take([3, h, dk], owner=(self, '_bq'))
            dwq, dwk, dwv = (dw_all.flat_view(i * h * dk * f, [h, dk, f]) for i in range(3))
            dbq, dbk, dbv = (db_all.flat_view(i * h * dk, [h, dk]) for i in range(3))
            dqkv = D.empty([b, sq, 3, h, dk])
            dq, dk_, dv_ = dqkv, dqkv.flat_view(f, [dqkv.size - f]), dqkv.flat_view(2 * f, [dqkv.size - 2 * f])
            gq = gk = gv = 3 * f
        else:
            dwq, dwk, dwv = (scope.take(w_.shape, owner=(self, a_)) for w_, a_ in ((wq, '_wq'), (wk, '_wk'), (wv, '_wv')))
            dbq, dbk, dbv = (scope.take([h, d_], owner=(self, a_)) for d_, a_ in ((dk, '_bq'), (dk, '_bk'), (dv, '_bv')))
            dq, dk_, dv_ = D.empty([b, sq, h, dk]), D.empty([b, skv, h, dk]), D.empty([b, skv, h, dv])
            gq, gk, gv = h * dk, h * dk, h * dv
        if self._core:
            # attentions.py:146-162 in one kernel: P is recomputed from the saved log-sum-exp, tile by tile
            D.mha_core_bwd(Mat(q, pq), Mat(k, pk), Mat(v, pv), ctx, self._lse, dctx, Mat(dq, gq), Mat(dk_, gk),
                           Mat(dv_, gv), (b, h, sq, skv, dk), self._scale, self._mask, self._raw_scores, neg_delta=neg_delta)
        else:
            # softmax @ V (attentions.py:146-148)
            # dP = dctx_h v_h^T followed by the softmax backward and the 1/sqrt(dk) of attentions.py:150-155.
            # The row term sum_j dP_ij P_ij equals dctx_i . ctx_i (ctx = P v), so it is one cheap row-dot and the
            # rest, datt = scale * P * (dP - row term), is elementwise: it rides the epilogue of the dP GEMM
            # and dP itself never goes to memory.
            datt = D.empty([b, h, sq, skv])
            if D.FUSE_SOFTMAX_BWD:
                delta = D.attn_rowdot(dctx, ctx)
                D.gemm(sq, skv, dv, Mat(dctx, h * dv, sq * h * dv, dv), Mat(v, pv, skv * pv, dv),
                       Mat(datt, skv, h * sq * skv, sq * skv), trans_b=True, batch=(b, h), alpha=self._scale,
                       softmax_bwd=(Mat(scores, skv), delta))
            else:
                D.gemm(sq, skv, dv, Mat(dctx, h * dv, sq * h * dv, dv), Mat(v, pv, skv * pv, dv),
                       Mat(datt, skv, h * sq * skv, sq * skv), trans_b=True, batch=(b, h))        # dctx_h v_h^T
                D.softmax_bwd(scores, datt, self._scale, out=datt)
            D.gemm(skv, dv, sq, Mat(scores, skv, h * sq * skv, sq * skv), Mat(dctx, h * dv, sq * h * dv, dv),
                   Mat(dv_, gv, skv * gv, dv), trans_a=True, batch=(b, h))                        # P_h^T dctx_h
            # Q K^T (attentions.py:161-162)
            D.gemm(sq, dk, skv, Mat(datt, skv, h * sq * skv, sq * skv), Mat(k, pk, skv * pk, dk),
                   Mat(dq, gq, sq * gq, dk), batch=(b, h))                                        # datt_h k_h
            D.gemm(skv, dk, sq, Mat(datt, skv, h * sq * skv, sq * skv), Mat(q, pq, sq * pq, dk),
                   Mat(dk_, gk, skv * gk, dk), trans_a=True, batch=(b, h))                        # datt_h^T q_h

        # in-projections (attentions.py:167-188): dw = dproj^T x ; dx = dproj w
        if packed:
            D.gemm(3 * f, f, m_q, Mat(dqkv, 3 * f), Mat(query, f), Mat(dw_all, f), trans_a=True, asum_out=db_all)
        else:
            D.gemm(h * dk, f, m_q, Mat(dq, gq), Mat(query, f), Mat(dwq, f), trans_a=True, asum_out=dbq)
            D.gemm(h * dk, f, m_kv, Mat(dk_, gk), Mat(key, f), Mat(dwk, f), trans_a=True, asum_out=dbk)
            D.gemm(h * dv, fv, m_kv, Mat(dv_, gv), Mat(value, fv), Mat(dwv, fv), trans_a=True, asum_out=dbv)
        # every parameter gradient of this layer now exists: start exchanging them (data parallel) under
        # the remaining input-gradient GEMMs
        scope.flush()
        if sum_inputs:
            assert query is key and key is value
            total = D.empty([b, sq, f])
            res = None if residual is None else Mat(residual, f)
            if packed and self._params_adjacent():      # dq wq + dk wk + dv wv: one contraction over the packed 3F axis
                D.gemm(m_q, f, 3 * f, Mat(dqkv, 3 * f), Mat(wq, f), Mat(total, f), residual=res)
            else:
                D.gemm(m_q, f, h * dk, Mat(dq, gq), Mat(wq, f), Mat(total, f), residual=res)
                D.gemm(m_kv, f, h * dk, Mat(dk_, gk), Mat(wk, f), Mat(total, f), residual=Mat(total, f))
                D.gemm(m_kv, fv, h * dv, Mat(dv_, gv), Mat(wv, fv), Mat(total, fv), residual=Mat(total, fv))
            result = total
        elif sum_kv:
            assert key is value and fv == f
            dquery, dkv = D.empty([b, sq, f]), D.empty([b, skv, f])
            D.gemm(m_q, f, h * dk, Mat(dq, gq), Mat(wq, f), Mat(dquery, f),
                   residual=None if residual is None else Mat(residual, f))
            D.gemm(m_kv, f, h * dk, Mat(dk_, gk), Mat(wk, f), Mat(dkv, f))
            D.gemm(m_kv, f, h * dv, Mat(dv_, gv), Mat(wv, f), Mat(dkv, f), residual=Mat(dkv, f))
            result = (dquery, dkv)
        else:
            assert residual is None
            dquery, dkey, dvalue = D.empty([b, sq, f]), D.empty([b, skv, f]), D.empty([b, skv, fv])
            D.gemm(m_q, f, h * dk, Mat(dq, gq), Mat(wq, f), Mat(dquery, f))
            D.gemm(m_kv, f, h * dk, Mat(dk_, gk), Mat(wk, f), Mat(dkey, f))
            assert value.shape == (b, skv, h * dv)
            D.gemm(m_kv, fv, h * dv, Mat(dv_, gv), Mat(wv, fv), Mat(dvalue, fv))
            result = (dquery, dkey, dvalue)

        # update order of attentions.py:190-197
        for attribute, grad in (('_wq', dwq), ('_wk', dwk), ('_wv', dwv), ('_wo', dwo),
                                ('_bq', dbq), ('_bk', dbk), ('_bv', dbv), ('_bo', dbo)):
            scope.defer(optimizer_, self, attribute, grad)
        return result
