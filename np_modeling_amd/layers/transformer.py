"""TransformerEncoder / TransformerDecoder (reference layers/transformer.py:8-203).

Composition only: the sub-layers do the arithmetic.  In the encoder -- with or without dropout: a DropOut always sits
directly in front of a LayerNormalization there and rides inside that norm's kernels (normalizations.py
``LayerNormalization._forward_impl``) -- and in the decoder with ``drop_rate == 0`` the residual additions of
transformer.py:39,53,78,90 (encoder; 130,141,152,170-198 decoder) and the three-way sum of transformer.py:85 are folded into GEMM
epilogues / the LayerNorm backward kernel, and the ReLU backward of ``dense1`` is folded
into the epilogue of ``dense2``'s dx GEMM, so no standalone elementwise pass runs.
All parameter updates are deferred to the end of ``backward`` (every dx is computed from
pre-update weights in the reference too), which lets the data-parallel gradient all-reduce
overlap the rest of the backward pass.
"""

from __future__ import annotations

from np_modeling_amd import device as D
from np_modeling_amd import parallel
from np_modeling_amd.layers import attentions, layer, mlp, normalizations


def _identity_dropout(*dropouts) -> bool:
    return all(d._drop_prob == 0.0 for d in dropouts)


def _dropout_folds(features: int, *dropouts) -> bool:
    """Whether the dropouts can ride inside the LayerNorm kernels behind them (device.layernorm_dropout_supported): always
    when they are the identity."""
    return _identity_dropout(*dropouts) or D.layernorm_dropout_supported(features)


def _linear_segments(lin):
    return [[(lin, '_b')], [(lin, '_w')]]              # Linear._backward_impl takes db, then dw


def _norm_segments(norm):
    return [[(norm, '_gamma')], [(norm, '_beta')]]


def _block_segments(norm, body, norm_first: bool):
    """A residual block's parameters in the order its backward produces their gradients: the norm's come after the
    body's with pre-norm, before them with post-norm."""
    return body + _norm_segments(norm) if norm_first else _norm_segments(norm) + body


# A transformer layer is a chain of residual blocks around a body (attention or the feed-forward pair):
#     pre-norm :  y = x + body(norm(drop(x)))          post-norm:  y = norm(drop(x + body(x)))
# The two helpers below are that block and its mirror image, built from standalone device adds -- the literal
# (unfused) composition, used when dropout is active, by the decoder, and as the reference the fused encoder is
# tested against.
def _block_forward(x, norm, dropout, norm_first: bool, body):
    x = D.as_device(x)
    if norm_first:
        return D.add(D.as_device(body(norm(dropout(x)))), x)
    return D.as_device(norm(dropout(D.add(D.as_device(body(x)), x))))


def _block_backward(dy, norm, dropout, norm_first: bool, body_backward, optimizer_, scope):
    """``body_backward(dy)`` returns the gradient w.r.t. the body's (first) input; the skip path adds ``dy``."""
    if not norm_first:
        dy = dropout.backward(norm._backward_impl(D.as_device(dy), optimizer_, scope))
    through_skip = D.as_device(dy)
    through_body = D.as_device(body_backward(through_skip))
    if norm_first:
        through_body = D.as_device(dropout.backward(norm._backward_impl(through_body, optimizer_, scope)))
    return D.add(through_body, through_skip)


class TransformerEncoder(layer.Layer):
    def __init__(self, num_heads: int, hidden_units: int, norm_first: bool, drop_rate: float = 0.0,
                 *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._self_attention = attentions.MultiHeadAttention(num_heads)
        self._dense1 = mlp.Dense(units=hidden_units)
        self._norm1 = normalizations.LayerNormalization()
        self._norm2 = normalizations.LayerNormalization()
        self._norm_first = norm_first
        self._dropout1 = normalizations.DropOut(drop_rate)
        self._dropout2 = normalizations.DropOut(drop_rate)
        self._fused = True      # which composition the last forward ran (the backward mirrors it)

    def initialize(self, qkv):
        features = qkv.shape[-1]
        self._dense2 = mlp.Linear(units=features)  # no activation (transformer.py:25-27)

    def _numel(self) -> int:
        att = self._self_attention._numel()
        lin1, lin2 = self._dense1._linear, self._dense2
        norms = 2 * (self._norm1._param('_gamma').size + self._norm2._param('_gamma').size)
        return att + lin1._w.size + lin1._b.size + lin2._w.size + lin2._b.size + norms + 64

    def _pack(self) -> None:
        """After the first forward (every sub-layer has drawn its parameters, in the reference's order): all 16 parameters
        into ONE arena, ordered as ``backward`` produces their gradients -- feed-forward block, then attention block --
        so that gradient bucket, exchange and the deferred updates walk one range (one optimizer launch per step)."""
        if self._arena is None:
            pre = self._norm_first
            ffn = _linear_segments(self._dense2) + _linear_segments(self._dense1._linear)
            if pre:     # dense2 | dense1, norm2 | attention | norm1 -- the four flushes of _backward_fused
                segments = ffn + _norm_segments(self._norm2) + self._self_attention._segments() + _norm_segments(self._norm1)
            else:       # norm2, dense2 | dense1 | norm1, attention
                segments = (_norm_segments(self._norm2) + ffn + _norm_segments(self._norm1)
                            + self._self_attention._segments())
            self._pack_parameters(segments)

    # Sub-layers initialise lazily at their first call, in call order, exactly as in the
    # reference (layer.py:33-35) -- that fixes the global-RNG draw order of the parameters.
    @staticmethod
    def _ensure(sub, *args):
        if not sub._initialized:
            sub.initialize(*args)
            sub._initialized = True

    def forward(self, qkv):
        qkv = D.as_device(qkv)
        batch, seq_len_q, features = qkv.shape
        self._fused = _dropout_folds(features, self._dropout1, self._dropout2) and self._dense1._fused_relu()
        if not self._fused:
            return self._forward_unfused(qkv)
        att, dense1, dense2 = self._self_attention, self._dense1, self._dense2
        norm1, norm2 = self._norm1, self._norm2
        skip = qkv
        h = qkv
        if self._norm_first:
            h = norm1._forward_impl(h, self._dropout1)                   # dropout1 inside the norm (transformer.py:35-36)
        self._ensure(att, h)
        out = att._forward_impl(h, h, h, residual=skip)                  # ... + skip (transformer.py:39)
        if not self._norm_first:
            out = norm1._forward_impl(out, self._dropout1)               # transformer.py:40-41
        out = out.reshape(-1, features)
        skip = out
        if self._norm_first:
            out = norm2._forward_impl(out, self._dropout2)               # transformer.py:49-50
        out = dense1(out)
        self._ensure(dense2, out)
        out = dense2._forward_impl(out, residual=skip)                   # ... + skip (transformer.py:53)
        if not self._norm_first:
            out = norm2._forward_impl(out, self._dropout2)               # transformer.py:55-56
        self._pack()
        return out.reshape(batch, seq_len_q, features)

    def _feed_forward(self, x):
        return self._dense2(self._dense1(x))

    def _feed_forward_backward(self, dy, optimizer_, scope):
        dy = self._dense2._backward_impl(D.as_device(dy), optimizer_, scope)
        dy = self._dense1._activation.backward(dy)
        return self._dense1._linear._backward_impl(D.as_device(dy), optimizer_, scope)

    def _forward_unfused(self, qkv):
        """The reference's composition (transformer.py:29-59) out of standalone kernels: two residual blocks."""
        batch, seq_len_q, features = qkv.shape
        out = _block_forward(qkv, self._norm1, self._dropout1, self._norm_first, self._self_attention)
        out = _block_forward(out.reshape(-1, features), self._norm2, self._dropout2, self._norm_first,
                             self._feed_forward)
        self._pack()
        return out.reshape(batch, seq_len_q, features)

    def backward(self, dy, optimizer_):
        dy = D.as_device(dy)
        with parallel.grad_scope(self._numel(), self._arena) as scope:
            if self._fused and self._dense1._fused_relu():           # the composition the forward ran
                return self._backward_fused(dy, optimizer_, scope)
            return self._backward_unfused(dy, optimizer_, scope)

    def _backward_fused(self, dy, optimizer_, scope):
        batch, seq_len_q, features = dy.shape
        att, lin1, lin2 = self._self_attention, self._dense1._linear, self._dense2
        pre1 = self._dense1._activation._x
        dy = dy.reshape(-1, features)
        if not self._norm_first:
            dy = self._norm2._backward_impl(dy, optimizer_, scope)
        dskip = dy
        # dense2: dx masked by dense1's ReLU (activations.py:19) in the GEMM epilogue; dense1's bias gradient
        # (the column sums of that masked dx, mlp.py:34) comes out of dense1's weight-gradient GEMM
        dh = lin2._backward_impl(dy, optimizer_, scope, relu_mask_pre=pre1)
        scope.flush()
        if self._norm_first:
            dy = lin1._backward_impl(dh, optimizer_, scope)
            dy = self._norm2._backward_impl(dy, optimizer_, scope, residual=dskip)     # dy += dskip
        else:
            dy = lin1._backward_impl(dh, optimizer_, scope, residual=dskip)            # dy += dskip
        scope.flush()
        dy = dy.reshape(batch, seq_len_q, features)
        if not self._norm_first:
            dy = self._norm1._backward_impl(dy, optimizer_, scope)
        dskip = dy
        if self._norm_first:
            dy = att._backward_impl(dy, optimizer_, scope, sum_inputs=True)            # dq + dk + dv
            dy = self._norm1._backward_impl(dy, optimizer_, scope, residual=dskip)     # dy += dskip
        else:
            dy = att._backward_impl(dy, optimizer_, scope, sum_inputs=True, residual=dskip)
        return dy

    def _backward_unfused(self, dy, optimizer_, scope):
        """Mirror of ``_forward_unfused`` (transformer.py:61-92); the attention block sums dquery + dkey + dvalue."""
        batch, seq_len_q, features = dy.shape
        dy = _block_backward(dy.reshape(-1, features), self._norm2, self._dropout2, self._norm_first,
                             lambda g: self._feed_forward_backward(g, optimizer_, scope), optimizer_, scope)
        return _block_backward(dy.reshape(batch, seq_len_q, features), self._norm1, self._dropout1, self._norm_first,
                               lambda g: D.add3(*self._self_attention._backward_impl(g, optimizer_, scope)),
                               optimizer_, scope)


class TransformerDecoder(layer.Layer):
    """Self-attention, cross-attention over ``kv``, feed-forward; three LayerNorms
    (transformer.py:95-203).  ``backward`` returns ``(dq, dkv)`` with
    ``dkv = dkey + dvalue`` of the cross-attention (transformer.py:186)."""

    def __init__(self, num_heads: int, hidden_units: int, norm_first: bool, drop_rate: float = 0.0,
                 *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._self_attention = attentions.MultiHeadAttention(num_heads)
        self._cross_attention = attentions.MultiHeadAttention(num_heads)
        self._dense1 = mlp.Dense(units=hidden_units)
        self._norm1 = normalizations.LayerNormalization()
        self._norm2 = normalizations.LayerNormalization()
        self._norm3 = normalizations.LayerNormalization()
        self._norm_first = norm_first
        self._dropout1 = normalizations.DropOut(drop_rate)
        self._dropout2 = normalizations.DropOut(drop_rate)
        self._dropout3 = normalizations.DropOut(drop_rate)

    def initialize(self, q, kv):
        features = q.shape[-1]
        self._dense2 = mlp.Linear(units=features)  # no activation

    def _feed_forward(self, x):
        return self._dense2(self._dense1(x))

    @staticmethod
    def _ensure(sub, *args):
        if not sub._initialized:
            sub.initialize(*args)
            sub._initialized = True

    _fused = True          # which composition the last forward ran (the backward mirrors it)

    def _fusable(self, features: int = 0) -> bool:
        """The fused composition: always without dropout; with dropout when the LayerNorm kernels can apply it (row
        length: device.layernorm_dropout_supported)."""
        return _dropout_folds(features, self._dropout1, self._dropout2, self._dropout3) and self._dense1._fused_relu()

    def _numel(self) -> int:
        """Floats the gradients of one backward need (26 parameter tensors): ONE bucket for the exchange, like the
        encoder's (round 4 opened the scope without a size: every gradient was a collective of its own at N > 1)."""
        lin1, lin2 = self._dense1._linear, self._dense2
        norms = sum(2 * n._param('_gamma').size for n in (self._norm1, self._norm2, self._norm3))
        return (self._self_attention._numel() + self._cross_attention._numel() + lin1._w.size + lin1._b.size
                + lin2._w.size + lin2._b.size + norms + 128)

    def _pack(self) -> None:
        """See TransformerEncoder._pack: feed-forward block, cross-attention block, self-attention block."""
        if self._arena is None:
            pre = self._norm_first
            ffn = _linear_segments(self._dense2) + _linear_segments(self._dense1._linear)
            ca, sa = self._cross_attention._segments(), self._self_attention._segments()
            self._pack_parameters(_block_segments(self._norm3, ffn, pre) + _block_segments(self._norm2, ca, pre)
                                  + _block_segments(self._norm1, sa, pre))

    def forward(self, q, kv):
        """Three residual blocks: self-attention, cross-attention over ``kv``, feed-forward (transformer.py:120-157).
        The residual additions ride the producing GEMMs' epilogues and the dropouts the LayerNorm kernels, as in the encoder."""
        q, kv = D.as_device(q), D.as_device(kv)
        batch, seq_len_q, features = q.shape
        self._fused = self._fusable(features)
        if not self._fused:
            return self._forward_unfused(q, kv)
        pre = self._norm_first
        sa, ca, dense1, dense2 = self._self_attention, self._cross_attention, self._dense1, self._dense2
        self._kv = kv
        # each DropOut sits directly in front of a LayerNormalization (transformer.py:125-126,131-132,137-138,142-143,
        # 149-150,154-155) and is applied inside that norm's kernels
        h = self._norm1._forward_impl(q, self._dropout1) if pre else q
        self._ensure(sa, h)
        out = sa._forward_impl(h, h, h, residual=q)                       # ... + skip (transformer.py:130)
        if not pre:
            out = self._norm1._forward_impl(out, self._dropout1)
        skip = out
        h = self._norm2._forward_impl(out, self._dropout2) if pre else out
        self._ensure(ca, h, kv)
        out = ca._forward_impl(h, kv, kv, residual=skip)                  # ... + skip (transformer.py:141)
        if not pre:
            out = self._norm2._forward_impl(out, self._dropout2)
        out = out.reshape(-1, features)
        skip = out
        h = self._norm3._forward_impl(out, self._dropout3) if pre else out
        h = dense1(h)
        self._ensure(dense2, h)
        out = dense2._forward_impl(h, residual=skip)                      # ... + skip (transformer.py:152)
        if not pre:
            out = self._norm3._forward_impl(out, self._dropout3)
        self._pack()
        return out.reshape(batch, seq_len_q, features)

    def _forward_unfused(self, q, kv):
        batch, seq_len_q, features = q.shape
        out = _block_forward(q, self._norm1, self._dropout1, self._norm_first, self._self_attention)
        out = _block_forward(out, self._norm2, self._dropout2, self._norm_first,
                             lambda x: self._cross_attention(x, kv))
        out = _block_forward(out.reshape(-1, features), self._norm3, self._dropout3, self._norm_first,
                             self._feed_forward)
        self._pack()
        return out.reshape(batch, seq_len_q, features)

    def backward(self, dy, optimizer_):
        """Returns ``(dq, dkv)``; ``dkv`` is the cross-attention's dkey + dvalue (transformer.py:159-203)."""
        dy = D.as_device(dy)
        with parallel.grad_scope(self._numel(), self._arena) as scope:
            if self._fused and self._dense1._fused_relu():           # the composition the forward ran
                return self._backward_fused(dy, optimizer_, scope)
            return self._backward_unfused(dy, optimizer_, scope)

    def _backward_fused(self, dy, optimizer_, scope):
        """Mirror of the fused forward: skip-connection gradients ride the LayerNorm backward (pre-norm) or the
        input-gradient GEMMs (post-norm); dense1's ReLU backward is the mask epilogue of dense2's dx GEMM; the
        cross-attention's dkey + dvalue and the self-attention's dq + dk + dv accumulate in GEMM epilogues."""
        batch, seq_len_q, features = dy.shape
        pre = self._norm_first
        sa, ca, lin1, lin2 = self._self_attention, self._cross_attention, self._dense1._linear, self._dense2
        dy = dy.reshape(-1, features)
        if not pre:
            dy = self._norm3._backward_impl(dy, optimizer_, scope)
        dskip = dy
        dh = lin2._backward_impl(dy, optimizer_, scope, relu_mask_pre=self._dense1._activation._x)
        scope.flush()
        if pre:
            dy = self._norm3._backward_impl(lin1._backward_impl(dh, optimizer_, scope), optimizer_, scope, residual=dskip)
        else:
            dy = lin1._backward_impl(dh, optimizer_, scope, residual=dskip)
        scope.flush()
        dy = dy.reshape(batch, seq_len_q, features)
        if not pre:
            dy = self._norm2._backward_impl(dy, optimizer_, scope)
        dskip = dy
        if pre:
            dquery, dkv = ca._backward_impl(dy, optimizer_, scope, sum_kv=True)
            dy = self._norm2._backward_impl(dquery, optimizer_, scope, residual=dskip)
        else:
            dy, dkv = ca._backward_impl(dy, optimizer_, scope, sum_kv=True, residual=dskip)
        if not pre:
            dy = self._norm1._backward_impl(dy, optimizer_, scope)
        dskip = dy
        if pre:
            dy = sa._backward_impl(dy, optimizer_, scope, sum_inputs=True)
            dy = self._norm1._backward_impl(dy, optimizer_, scope, residual=dskip)
        else:
            dy = sa._backward_impl(dy, optimizer_, scope, sum_inputs=True, residual=dskip)
        return dy, dkv

    def _backward_unfused(self, dy, optimizer_, scope):
        batch, seq_len_q, features = dy.shape
        kv_grad = []

        def feed_forward_backward(g):
            g = self._dense2._backward_impl(D.as_device(g), optimizer_, scope)
            g = self._dense1._activation.backward(g)
            return self._dense1._linear._backward_impl(D.as_device(g), optimizer_, scope)

        def cross_attention_backward(g):
            dquery, dkey, dvalue = self._cross_attention._backward_impl(g, optimizer_, scope)
            kv_grad.append(D.add(dkey, dvalue))
            return dquery

        def self_attention_backward(g):
            return D.add3(*self._self_attention._backward_impl(g, optimizer_, scope))

        dy = _block_backward(dy.reshape(-1, features), self._norm3, self._dropout3, self._norm_first,
                             feed_forward_backward, optimizer_, scope)
        dy = _block_backward(dy.reshape(batch, seq_len_q, features), self._norm2, self._dropout2,
                             self._norm_first, cross_attention_backward, optimizer_, scope)
        dy = _block_backward(dy, self._norm1, self._dropout1, self._norm_first, self_attention_backward,
                             optimizer_, scope)
        return dy, kv_grad[0]
