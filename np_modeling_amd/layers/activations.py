"""ReLU and Softmax on the GPU (reference layers/activations.py:8-45)."""

from __future__ import annotations

from np_modeling_amd import _C
from np_modeling_amd import device as D
from np_modeling_amd.layers import layer


class Activation(layer.Layer):
    pass


class ReLU(Activation):
    """``max(x, 0)``; the gradient passes where ``x >= 0`` (activations.py:15,19).

    ``backward(dy)`` takes no optimizer argument, exactly like the reference."""

    def forward(self, x):
        x = D.as_device(x)
        self._x = x
        y = D.empty(x.shape)
        _C.check(_C.lib().npm_relu_fwd(x.ptr, y.ptr, x.size), 'npm_relu_fwd')
        return y

    def backward(self, dy):
        dy = D.as_device(dy)
        assert dy.shape == self._x.shape, f'{dy.shape} vs {self._x.shape}'
        dx = D.empty(dy.shape)
        _C.check(_C.lib().npm_relu_bwd(self._x.ptr, dy.ptr, dx.ptr, dy.size), 'npm_relu_bwd')
        return dx


class Softmax(Activation):
    """Max-shifted softmax over the last axis (activations.py:26-29).  The backward is the
    closed form ``y * (dy - sum(dy * y))`` of the reference's Jacobian einsum
    (activations.py:32-45): one wavefront per row, O(rows * n) instead of O(rows * n^2)."""

    def forward(self, x):
        x = D.as_device(x)
        self._x = x
        self._y = self._run_forward(x, 1.0)
        return self._y

    def backward(self, dy, *args, **kwargs):
        return self._run_backward(self._y, D.as_device(dy), 1.0)

    @staticmethod
    def _run_forward(x, scale, out=None):
        n = x.shape[-1] if x.ndim else 1
        rows = x.size // n if n else 0
        out = D.empty(x.shape) if out is None else out
        _C.check(_C.lib().npm_softmax_fwd(x.ptr, out.ptr, rows, n, float(scale)), 'npm_softmax_fwd')
        return out

    @staticmethod
    def _run_backward(y, dy, scale, out=None):
        assert dy.shape == y.shape, f'{dy.shape} vs {y.shape}'
        n = y.shape[-1] if y.ndim else 1
        rows = y.size // n if n else 0
        out = D.empty(y.shape) if out is None else out
        _C.check(_C.lib().npm_softmax_bwd(y.ptr, dy.ptr, out.ptr, rows, n, float(scale)), 'npm_softmax_bwd')
        return out
