"""Losses with the reference's interface (reference loss.py:10-39).

When the network output is a ``DeviceArray`` the loss value is reduced on the device (fp64
accumulation, one scalar copied back) and the gradient stays on the device, so a Trainer step
never copies activations to the host; host inputs take the NumPy path.
"""

from __future__ import annotations

import abc
import ctypes as C

import numpy as np

from np_modeling_amd import _C
from np_modeling_amd import device as D
from np_modeling_amd.layers import layer


class Loss(layer.Layer):
    @abc.abstractmethod
    def forward(self, *args, **kwargs) -> float:
        pass

    @abc.abstractmethod
    def backward(self, *args, **kwargs):
        pass

    def _device_targets(self, targets, shape):
        """Host targets are read afresh at every forward, as the reference does (a caller may refill the same
        array in place between steps); ``Trainer.train`` uploads them once before its loop instead."""
        if isinstance(targets, D.DeviceArray):
            if tuple(targets.shape) == tuple(shape):
                return targets
            # ``y - targets`` broadcasts in the reference (loss.py:24,28): resident targets of another, broadcastable
            # shape ([B, 1], a scalar) are expanded once per (targets, shape) -- the kernels read y.size elements
            cached = getattr(self, '_expanded', None)
            if cached is None or cached[0] is not targets or cached[1] != tuple(shape):
                wide = np.broadcast_to(np.asarray(targets, dtype=np.float32), shape)    # ValueError if not broadcastable
                self._expanded = cached = (targets, tuple(shape), D.from_host(wide))
            return cached[2]
        return D.from_host(np.broadcast_to(np.asarray(targets, dtype=np.float32), shape))


class MSELoss(Loss):
    """sum((y - t)^2) / y.size and its gradient 2 (y - t) / y.size (loss.py:21-29)."""

    def forward(self, y, targets) -> float:
        self._y = y
        if isinstance(y, D.DeviceArray):
            self._targets = self._device_targets(targets, y.shape)
            value = C.c_double()
            _C.check(_C.lib().npm_mse_fwd(y.ptr, self._targets.ptr, y.size, C.byref(value)), 'npm_mse_fwd')
            return np.float64(value.value)
        self._y = np.asarray(y)
        self._targets = np.asarray(targets)
        delta = self._y - self._targets
        return np.sum(delta ** 2) / self._y.size

    def backward(self, *args, **kwargs):
        if isinstance(self._y, D.DeviceArray):
            dy = D.empty(self._y.shape)
            _C.check(_C.lib().npm_mse_bwd(self._y.ptr, self._targets.ptr, dy.ptr, dy.size), 'npm_mse_bwd')
            return dy
        return 2 * (self._y - self._targets) / self._y.size


class CrossEntropyLoss(Loss):
    """-sum(t * log(y)) and -t / y (loss.py:33-39)."""

    def forward(self, y, targets) -> float:
        self._y = y
        if isinstance(y, D.DeviceArray):
            self._targets = self._device_targets(targets, y.shape)
            value = C.c_double()
            _C.check(_C.lib().npm_xent_fwd(y.ptr, self._targets.ptr, y.size, C.byref(value)), 'npm_xent_fwd')
            return np.float64(value.value)
        self._y = np.asarray(y)
        self._targets = np.asarray(targets)
        return -np.sum(self._targets * np.log(self._y))

    def backward(self, *args, **kwargs):
        if isinstance(self._y, D.DeviceArray):
            dy = D.empty(self._y.shape)
            _C.check(_C.lib().npm_xent_bwd(self._y.ptr, self._targets.ptr, dy.ptr, dy.size), 'npm_xent_bwd')
            return dy
        return -self._targets / self._y
