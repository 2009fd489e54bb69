"""LayerNormalization and DropOut (reference layers/normalizations.py:9-75)."""

from __future__ import annotations

from typing import Optional

import numpy as np

from np_modeling_amd import _C
from np_modeling_amd import device as D
from np_modeling_amd import parallel
from np_modeling_amd.layers import layer


class _DeviceRng:
    """State of the device-side mask generator (csrc/npm_optim.hip dropout_philox_kernel): a seed and a running
    offset, one offset per forward call, so every call draws an independent stream and a run is reproducible from
    its seed.  Off by default: the reference draws masks with the HOST generator (np.random.binomial), and seeded
    parity with it needs that path."""

    enabled = False
    seed = 0
    offset = 0


def set_dropout_rng(kind: str = 'host', seed: int = 0) -> None:
    """``'host'``: masks from NumPy's global generator with the reference's exact call (default; seeded runs drop the
    same elements as the reference).  ``'device'``: masks drawn on the GPU by Philox4x32-10 -- no host draw, no PCIe
    upload of a mask per forward; ``_mask`` stays readable (copied to the host on access)."""
    if kind not in ('host', 'device'):
        raise ValueError("set_dropout_rng: kind is 'host' or 'device'")
    _DeviceRng.enabled = kind == 'device'
    _DeviceRng.seed, _DeviceRng.offset = int(seed), 0


class DropOut(layer.Layer):
    """``drop_prob == 0`` is the identity and returns its argument unchanged
    (normalizations.py:14-23) -- the only path the encoder exercises.  For p > 0 the mask is
    drawn on the HOST with the reference's exact call (``np.random.binomial``), so a seeded run
    drops the same elements and ``_mask`` stays a readable NumPy array (reference
    layers/normalizations_test.py:15-30 reads it); applying it is one device kernel.
    :func:`set_dropout_rng` switches to masks drawn on the device."""

    def __init__(self, drop_prob: float, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._drop_prob = drop_prob
        self._host_mask = None
        self._mask_dev = None

    @property
    def _mask(self):
        """The 0/1 mask of the last forward as a NumPy array (device-drawn masks are copied to the host here)."""
        if self._host_mask is None and self._mask_dev is not None:
            raw = np.empty(self._mask_dev.nbytes, dtype=np.uint8)
            _C.check(_C.lib().npm_d2h(raw.ctypes.data, self._mask_dev.ptr, raw.nbytes), 'npm_d2h')
            self._host_mask = raw.reshape(self._mask_shape).astype(np.int64)
        return self._host_mask

    @_mask.setter
    def _mask(self, value):
        self._host_mask = value

    def _apply(self, x):
        keep = 1 - self._drop_prob
        x = D.as_device(x)
        out = D.empty(x.shape)
        _C.check(_C.lib().npm_mask_scale(x.ptr, self._mask_dev.ptr, out.ptr, x.size, float(keep)), 'npm_mask_scale')
        return out

    def forward(self, x, training: bool = True):
        if training and self._drop_prob != 0.0:
            keep = 1 - self._drop_prob
            if _DeviceRng.enabled:
                x = D.as_device(x)
                out = D.empty(x.shape)
                self._mask_dev, self._mask_shape, self._host_mask = D.ByteBuffer(x.size), tuple(x.shape), None
                _C.check(_C.lib().npm_dropout_philox(x.ptr, out.ptr, self._mask_dev.ptr, x.size, float(keep),
                                                     _DeviceRng.seed, _DeviceRng.offset), 'npm_dropout_philox')
                _DeviceRng.offset += 1
                return out
            self._mask = np.random.binomial(n=1, p=keep, size=x.size).reshape(x.shape)
            self._mask_dev = D.bytes_from_host(self._mask.astype(np.uint8))
            return self._apply(x)
        return x

    def _draw(self, shape):
        """The mask of a forward over a tensor of ``shape`` WITHOUT applying it -- for a consumer that applies it on its way in
        (LayerNormalization._forward_impl: the encoder's fused path).  Same draws as ``forward``: the host generator's
        ``np.random.binomial`` call of the reference, or the next Philox offset.  Returns ``(mask bytes, keep_prob)``, or None
        when this dropout is the identity."""
        if self._drop_prob == 0.0:
            return None
        keep = 1 - self._drop_prob
        size = D._prod(shape)
        if _DeviceRng.enabled:
            self._mask_dev, self._mask_shape, self._host_mask = D.ByteBuffer(size), tuple(shape), None
            _C.check(_C.lib().npm_dropout_philox(None, None, self._mask_dev.ptr, size, float(keep),
                                                 _DeviceRng.seed, _DeviceRng.offset), 'npm_dropout_philox')
            _DeviceRng.offset += 1
        else:
            self._mask = np.random.binomial(n=1, p=keep, size=size).reshape(shape)
            self._mask_dev = D.bytes_from_host(self._mask.astype(np.uint8))
        return self._mask_dev, keep

    def backward(self, dl_dy, *args, **kwargs):
        if self._drop_prob != 0.0:
            return self._apply(dl_dy)
        return dl_dy


class LayerNormalization(layer.StatefulLayer):
    """Row-wise normalisation over the last axis with biased variance and epsilon inside the
    square root (normalizations.py:45-48).  gamma and beta are RANDOM-initialised (not 1/0),
    gamma first (normalizations.py:40-41).  One wavefront per row; the backward is the closed
    form of the reference's [rows, d, d] Jacobian einsum (normalizations.py:58-71)."""

    _drop = None            # (mask bytes, keep_prob) of the DropOut folded into the last forward, or None

    def __init__(self, epsilon: float = 1e-3, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._epsilon = epsilon

    def initialize(self, x):
        self._col = x.shape[-1]
        self._gamma = self._new_param([self._col])
        self._beta = self._new_param([self._col])
        self._pack_parameters([[(self, '_gamma')], [(self, '_beta')]])

    def forward(self, x):
        return self._forward_impl(x)

    def _forward_impl(self, x, dropout: Optional[DropOut] = None):
        """``dropout``: the DropOut layer that sits directly in front of this norm in a composite (transformer.py:35-36,
        40-41,49-50,55-56).  Its mask is drawn here and applied INSIDE the kernels -- forward on the way in, backward on the
        way in (to rebuild the norm's input) and on the way out (DropOut.backward) -- so ``x`` stays the dropout's input and
        the dropped tensor is never stored; ``_backward_impl`` then returns the gradient with respect to that input."""
        x = D.as_device(x)
        drop = dropout._draw(x.shape) if dropout is not None else None      # the dropout runs first: its draws precede this
        if not self._initialized:                                            # layer's lazy parameter draws (layer.py:33-35)
            self.initialize(x)
            self._initialized = True
        gamma, beta = self._param('_gamma'), self._param('_beta')
        d = x.shape[-1]
        assert gamma.shape == (d,), f'{gamma.shape} vs {d}'
        self._x = x
        self._drop = drop
        z, self._mean, self._rstd = D.layernorm_fwd(x, gamma, beta, self._epsilon, drop=self._drop)
        return z

    def backward(self, dl_dz, optimizer_):
        with parallel.grad_scope(2 * self._param('_gamma').size + 8, self._arena) as scope:
            return self._backward_impl(D.as_device(dl_dz), optimizer_, scope)

    def _backward_impl(self, dz: D.DeviceArray, optimizer_, scope,
                       residual: Optional[D.DeviceArray] = None) -> D.DeviceArray:
        """dx (+ residual), and deferred updates of gamma then beta (normalizations.py:73-74)."""
        x = self._x
        gamma = self._param('_gamma')
        d = x.shape[-1]
        assert dz.size == x.size, f'{dz.shape} vs {x.shape}'
        dgamma, dbeta = scope.take([d], owner=(self, '_gamma')), scope.take([d], owner=(self, '_beta'))
        dx = D.layernorm_bwd(dz, x, self._mean, self._rstd, gamma, dgamma, dbeta, residual=residual, drop=self._drop)
        scope.defer(optimizer_, self, '_gamma', dgamma)
        scope.defer(optimizer_, self, '_beta', dbeta)
        return dx
