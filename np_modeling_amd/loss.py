"""Losses with the reference's interface (reference loss.py:10-39).

O(N) once per step and outside the hot path: evaluated on the
host from a device-to-host copy of the network output; the returned gradient goes back to
the device when the last layer's ``backward`` consumes it.
"""

from __future__ import annotations

import abc

import numpy as np

from np_modeling_amd.layers import layer


class Loss(layer.Layer):
    @abc.abstractmethod
    def forward(self, *args, **kwargs) -> float:
        pass

    @abc.abstractmethod
    def backward(self, *args, **kwargs):
        pass


class MSELoss(Loss):
    """sum((y - t)^2) / y.size and its gradient 2 (y - t) / y.size (loss.py:21-29)."""

    def forward(self, y, targets) -> float:
        self._y = np.asarray(y)
        self._targets = np.asarray(targets)
        delta = self._y - self._targets
        return np.sum(delta ** 2) / self._y.size

    def backward(self, *args, **kwargs):
        return 2 * (self._y - self._targets) / self._y.size


class CrossEntropyLoss(Loss):
    """-sum(t * log(y)) and -t / y (loss.py:33-39)."""

    def forward(self, y, targets) -> float:
        self._y = np.asarray(y)
        self._targets = np.asarray(targets)
        return -np.sum(self._targets * np.log(self._y))

    def backward(self, *args, **kwargs):
        return -self._targets / self._y
