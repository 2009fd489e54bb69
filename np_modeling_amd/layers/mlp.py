"""Linear / Dense on the fp32 MFMA GEMM (reference layers/mlp.py:11-82).

``Linear``: y = x @ w + b as one GEMM with the bias in the epilogue; backward is two
GEMMs (dw = x^T dy with split-K, dx = dy w^T) plus a column sum for db.
``Dense`` with the default ReLU fuses the activation into the forward epilogue (the
pre-activation is kept for the ``x >= 0`` test of the backward, activations.py:19).
"""

from __future__ import annotations

from typing import Optional

from np_modeling_amd import device as D
from np_modeling_amd import parallel
from np_modeling_amd.device import Mat
from np_modeling_amd.layers import activations, layer


class Linear(layer.StatefulLayer):
    def __init__(self, units: int, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._output_units = units

    def initialize(self, x) -> None:
        self._input_units = x.shape[-1]
        # draw order w then b (mlp.py:18-19)
        self._w = self._new_param([self._input_units, self._output_units])
        self._b = self._new_param([self._output_units])
        self._pack_parameters([[(self, '_b')], [(self, '_w')]])      # the order backward produces db, dw in

    # -- forward ---------------------------------------------------------------------
    def forward(self, x):
        return self._forward_impl(D.as_device(x))

    def _forward_impl(self, x: D.DeviceArray, residual: Optional[D.DeviceArray] = None,
                      relu_pre: Optional[D.DeviceArray] = None) -> D.DeviceArray:
        """y = x @ w + b (+ residual) ; with ``relu_pre`` given: relu_pre = y, return max(y, 0)."""
        w, b = self._param('_w'), self._param('_b')
        k, n = w.shape
        assert x.shape[-1] == k, f'{x.shape} vs {w.shape}'
        self._x = x
        m = x.size // k if k else 0
        y = D.empty(tuple(x.shape[:-1]) + (n,))
        D.gemm(m, n, k, Mat(x, k), Mat(w, n), Mat(y, n), bias=b,
               residual=None if residual is None else Mat(residual, n),
               relu_save=None if relu_pre is None else Mat(relu_pre, n))
        return y

    # -- backward --------------------------------------------------------------------
    def backward(self, dy, optimizer_):
        with parallel.grad_scope(self._w.size + self._b.size + 8, self._arena) as scope:
            return self._backward_impl(D.as_device(dy), optimizer_, scope)

    def _backward_impl(self, dy: D.DeviceArray, optimizer_, scope, *,
                       relu_mask_pre: Optional[D.DeviceArray] = None,
                       residual: Optional[D.DeviceArray] = None,
                       need_dx: bool = True, db: Optional[D.DeviceArray] = None) -> Optional[D.DeviceArray]:
        """db, dw, dx of mlp.py:34-36.  ``relu_mask_pre``: zero dx where that (the producer's
        pre-activation) is negative -- the upstream Dense's ReLU backward fused into this
        GEMM's epilogue.  ``residual``: dx += residual (a skip connection's gradient).
        ``db``: the bias gradient already taken by whoever produced ``dy`` (skips the column sum)."""
        w = self._param('_w')
        x = self._x
        k, n = w.shape
        # dy: [m, n]; x: [m, k] -- 2-D only, like the reference (mlp.py:33)
        assert dy.shape == (x.shape[0], n), f'{dy.shape} vs {(x.shape[0], n)}'
        m = dy.shape[0]
        have_db = db is not None
        if not have_db:
            db = scope.take([n], owner=(self, '_b'))
        dw = scope.take([k, n], owner=(self, '_w'))
        # x^T @ dy; the same GEMM sums its dy tiles over the batch: db = np.sum(dy, axis=0) (mlp.py:34)
        D.gemm(k, n, m, Mat(x, k), Mat(dy, n), Mat(dw, n), trans_a=True, bsum_out=None if have_db else db)
        dx = None
        if need_dx:
            dx = D.empty([m, k])
            D.gemm(m, k, n, Mat(dy, n), Mat(w, n), Mat(dx, k), trans_b=True,       # dy @ w^T
                   residual=None if residual is None else Mat(residual, k),
                   relu_mask=None if relu_mask_pre is None else Mat(relu_mask_pre, k))
            assert dx.shape == x.shape
        scope.defer(optimizer_, self, '_w', dw)
        scope.defer(optimizer_, self, '_b', db)
        return dx

    @property
    def w(self):
        assert self._initialized
        return self._param('_w')

    @property
    def b(self):
        assert self._initialized
        return self._param('_b')


class Dense(layer.StatefulLayer):
    """Linear followed by an activation, ReLU by default (mlp.py:53-82)."""

    def __init__(self, units: int, activation: Optional[activations.Activation] = None, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._linear = Linear(units=units)
        self._activation = activation or activations.ReLU()

    def initialize(self, x) -> None:
        self._linear.initialize(x)
        self._linear._initialized = True
        self._activation.initialize()
        self._activation._initialized = True

    def _fused_relu(self) -> bool:
        return type(self._activation) is activations.ReLU

    def forward(self, x):
        x = D.as_device(x)
        if self._fused_relu():
            pre = D.empty(tuple(x.shape[:-1]) + (self._linear._output_units,))
            y = self._linear._forward_impl(x, relu_pre=pre)
            self._activation._x = pre            # what ReLU.forward would have cached
            return y
        return self._activation.forward(self._linear._forward_impl(x))

    def backward(self, dy, optimizer_):
        lin = self._linear
        with parallel.grad_scope(lin._w.size + lin._b.size + 8, lin._arena) as scope:
            dy = D.as_device(dy)
            if self._fused_relu():           # relu' (mlp.py:74) and db (mlp.py:34) in one pass over dy
                db = scope.take([lin._output_units], owner=(lin, '_b'))
                g = D.relu_bwd_colsum(self._activation._x, dy, lin._output_units, db)
                return lin._backward_impl(g, optimizer_, scope, db=db)
            return lin._backward_impl(self._activation.backward(dy), optimizer_, scope)

    @property
    def linear(self) -> Linear:
        assert self._initialized
        return self._linear
