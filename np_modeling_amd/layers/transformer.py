"""TransformerEncoder / TransformerDecoder (reference layers/transformer.py:8-203).

Composition only: the sub-layers do the arithmetic.  With ``drop_rate == 0`` (the only
configuration the reference's tests and the benchmark use) the residual additions of
transformer.py:39,53,78,90 and the three-way sum of transformer.py:85 are folded into GEMM
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

    def initialize(self, qkv):
        features = qkv.shape[-1]
        self._dense2 = mlp.Linear(units=features)  # no activation (transformer.py:25-27)

    def _numel(self) -> int:
        att = self._self_attention._numel()
        lin1, lin2 = self._dense1._linear, self._dense2
        norms = 2 * (self._norm1._param('_gamma').size + self._norm2._param('_gamma').size)
        return att + lin1._w.size + lin1._b.size + lin2._w.size + lin2._b.size + norms + 64

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
        if not (_identity_dropout(self._dropout1, self._dropout2)
                and self._dense1._fused_relu()):
            return self._forward_unfused(qkv)
        att, dense1, dense2 = self._self_attention, self._dense1, self._dense2
        skip = qkv
        h = qkv
        if self._norm_first:
            h = self._norm1(h)
        self._ensure(att, h)
        out = att._forward_impl(h, h, h, residual=skip)                  # ... + skip (transformer.py:39)
        if not self._norm_first:
            out = self._norm1(out)
        out = out.reshape(-1, features)
        skip = out
        if self._norm_first:
            out = self._norm2(out)
        out = dense1(out)
        self._ensure(dense2, out)
        out = dense2._forward_impl(out, residual=skip)                   # ... + skip (transformer.py:53)
        if not self._norm_first:
            out = self._norm2(out)
        return out.reshape(batch, seq_len_q, features)

    def _forward_unfused(self, qkv):
        """Literal transcription of the reference order with standalone adds (dropout > 0)."""
        batch, seq_len_q, features = qkv.shape
        skip = qkv
        if self._norm_first:
            qkv = self._norm1(self._dropout1(qkv))
        out = D.add(D.as_device(self._self_attention(qkv)), D.as_device(skip))
        if not self._norm_first:
            out = self._norm1(self._dropout1(out))
        out = D.as_device(out).reshape(-1, features)
        skip = out
        if self._norm_first:
            out = self._norm2(self._dropout2(out))
        out = self._dense2(self._dense1(out))
        out = D.add(out, skip)
        if not self._norm_first:
            out = self._norm2(self._dropout2(out))
        return D.as_device(out).reshape(batch, seq_len_q, features)

    def backward(self, dy, optimizer_):
        dy = D.as_device(dy)
        with parallel.grad_scope(self._numel()) as scope:
            if _identity_dropout(self._dropout1, self._dropout2) and self._dense1._fused_relu():
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
        batch, seq_len_q, features = dy.shape
        dy = dy.reshape(-1, features)
        if not self._norm_first:
            dy = self._dropout2.backward(self._norm2._backward_impl(dy, optimizer_, scope))
        dskip = D.as_device(dy)
        dy = self._dense2._backward_impl(dskip, optimizer_, scope)
        dy = self._dense1._activation.backward(dy)
        dy = self._dense1._linear._backward_impl(D.as_device(dy), optimizer_, scope)
        if self._norm_first:
            dy = self._dropout2.backward(self._norm2._backward_impl(dy, optimizer_, scope))
        dy = D.add(D.as_device(dy), dskip).reshape(batch, seq_len_q, features)
        if not self._norm_first:
            dy = self._dropout1.backward(self._norm1._backward_impl(dy, optimizer_, scope))
        dskip = D.as_device(dy)
        dq, dk, dv = self._self_attention._backward_impl(dskip, optimizer_, scope)
        dy = D.add3(dq, dk, dv)
        if self._norm_first:
            dy = self._dropout1.backward(self._norm1._backward_impl(dy, optimizer_, scope))
        return D.add(D.as_device(dy), dskip)


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

    def forward(self, q, kv):
        q, kv = D.as_device(q), D.as_device(kv)
        batch, seq_len_q, features = q.shape
        skip = q
        if self._norm_first:
            q = self._norm1(self._dropout1(q))
        out = D.add(D.as_device(self._self_attention(q)), D.as_device(skip))
        if not self._norm_first:
            out = self._norm1(self._dropout1(out))
        skip = out
        if self._norm_first:
            out = self._norm2(self._dropout2(out))
        out = D.add(D.as_device(self._cross_attention(out, kv)), D.as_device(skip))
        if not self._norm_first:
            out = self._norm2(self._dropout2(out))
        out = D.as_device(out).reshape(-1, features)
        skip = out
        if self._norm_first:
            out = self._norm3(self._dropout3(out))
        out = self._dense2(self._dense1(out))
        out = D.add(out, skip)
        if not self._norm_first:
            out = self._norm3(self._dropout3(out))
        return D.as_device(out).reshape(batch, seq_len_q, features)

    def backward(self, dy, optimizer_):
        dy = D.as_device(dy)
        batch, seq_len_q, features = dy.shape
        with parallel.grad_scope(0) as scope:
            dy = dy.reshape(-1, features)
            if not self._norm_first:
                dy = self._dropout3.backward(self._norm3._backward_impl(dy, optimizer_, scope))
            dskip = D.as_device(dy)
            dy = self._dense2._backward_impl(dskip, optimizer_, scope)
            dy = self._dense1._activation.backward(dy)
            dy = self._dense1._linear._backward_impl(D.as_device(dy), optimizer_, scope)
            if self._norm_first:
                dy = self._dropout3.backward(self._norm3._backward_impl(dy, optimizer_, scope))
            dy = D.add(D.as_device(dy), dskip).reshape(batch, seq_len_q, features)

            if not self._norm_first:
                dy = self._dropout2.backward(self._norm2._backward_impl(dy, optimizer_, scope))
            dskip = D.as_device(dy)
            dq, dk, dv = self._cross_attention._backward_impl(dskip, optimizer_, scope)
            dkv = D.add(dk, dv)
            dy = dq
            if self._norm_first:
                dy = self._dropout2.backward(self._norm2._backward_impl(dy, optimizer_, scope))
            dy = D.add(D.as_device(dy), dskip)

            if not self._norm_first:
                dy = self._dropout1.backward(self._norm1._backward_impl(dy, optimizer_, scope))
            dskip = D.as_device(dy)
            dq, dk, dv = self._self_attention._backward_impl(dskip, optimizer_, scope)
            dy = D.add3(dq, dk, dv)
            if self._norm_first:
                dy = self._dropout1.backward(self._norm1._backward_impl(dy, optimizer_, scope))
            dy = D.add(D.as_device(dy), dskip)
        return dy, dkv
