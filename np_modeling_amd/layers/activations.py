"""ReLU and Softmax on the GPU (reference layers/activations.py:8-45)."""

from __future__ import annotations

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
        return D.relu_fwd(x)

    def backward(self, dy):
        dy = D.as_device(dy)
        assert dy.shape == self._x.shape, f'{dy.shape} vs {self._x.shape}'
        return D.relu_bwd(self._x, dy)


class Softmax(Activation):
    """Max-shifted softmax over the last axis (activations.py:26-29).  The backward is the
    closed form ``y * (dy - sum(dy * y))`` of the reference's Jacobian einsum
    (activations.py:32-45): one wavefront per row, O(rows * n) instead of O(rows * n^2)."""

    def forward(self, x):
        x = D.as_device(x)
        self._x = x
        self._y = D.softmax_fwd(x)
        return self._y

    def backward(self, dy, *args, **kwargs):
        dy = D.as_device(dy)
        assert dy.shape == self._y.shape, f'{dy.shape} vs {self._y.shape}'
        return D.softmax_bwd(self._y, dy)
