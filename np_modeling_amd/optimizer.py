"""Optimizers with the reference's call contract (reference optimizer.py:12-69).

A layer calls ``optimizer_.update(layer, '_w', grad)`` once per parameter; ``update`` looks
the parameter up, lets ``update_variable`` change it IN PLACE and stores it back.  The
reference's own unchanged ``optimizer.py`` also drives the device layers (``DeviceArray``
implements ``lr * grad`` and ``-=``); these classes are the same contract for machines
where the reference is not on the path, with the SGD step as a single device axpy.
Adam keeps the reference's numerics -- epsilon INSIDE the square root, bias correction,
fp64 moments keyed by ``f'{id(obj)}.{attribute}'``.  When parameter and gradient live on the
device the whole update is one kernel (``npm_adam_step``) on device-resident fp64 moments;
host arrays take the NumPy path.
"""

from __future__ import annotations

import abc
import dataclasses

import numpy as np


class Optimizer(metaclass=abc.ABCMeta):
    def update(self, obj: object, attribute: str, gradient) -> None:
        key = f'{id(obj)}.{attribute}'
        updated = self.update_variable(key, getattr(obj, attribute), gradient)
        setattr(obj, attribute, updated)

    @abc.abstractmethod
    def update_variable(self, identifier: str, variable, gradient):
        """Return the updated variable (mutated in place)."""


class SGDOptimizer(Optimizer):
    def __init__(self, learning_rate: float):
        self._learning_rate = learning_rate

    def update_variable(self, identifier, variable, gradient):
        step = self._learning_rate * gradient      # DeviceArray -> symbolic; consumed by one axpy
        variable -= step
        return variable


@dataclasses.dataclass
class AdamOptimizerConfig:
    learning_rate: float
    beta1: float = 0.9
    beta2: float = 0.999
    epsilon: float = 1e-7

    def __post_init__(self, *args, **kwargs):
        self._steps = {}
        self._momentums = {}
        self._velocities = {}


class _DeviceMoments:
    """fp64 first/second moments of one parameter, resident in HBM."""

    def __init__(self, n: int):
        from np_modeling_amd import _C, device as D
        self.n = n
        self.m = D._Buffer(8 * n)
        self.v = D._Buffer(8 * n)
        _C.check(_C.lib().npm_fill_f64(self.m.ptr, 0.0, n), 'npm_fill_f64')
        _C.check(_C.lib().npm_fill_f64(self.v.ptr, 0.0, n), 'npm_fill_f64')


class AdamOptimizer(AdamOptimizerConfig, Optimizer):
    def _device_step(self, identifier, variable, gradient):
        from np_modeling_amd import _C, device as D
        if isinstance(gradient, D.Scaled):
            gradient = gradient.materialize()
        step = self._steps.get(identifier, 1)
        state = self._momentums.get(identifier)
        if not isinstance(state, _DeviceMoments) or state.n != variable.size:
            state = _DeviceMoments(variable.size)
        _C.check(_C.lib().npm_adam_step(variable.ptr, gradient.ptr, state.m.ptr, state.v.ptr, variable.size,
                                        float(self.learning_rate), float(self.beta1), float(self.beta2),
                                        float(self.epsilon), int(step)), 'npm_adam_step')
        self._steps[identifier] = step + 1
        self._momentums[identifier] = state
        self._velocities[identifier] = state
        return variable

    def update_variable(self, identifier, variable, gradient):
        from np_modeling_amd import device as D
        if isinstance(variable, D.DeviceArray) and isinstance(gradient, (D.DeviceArray, D.Scaled)) \
                and gradient.size == variable.size:
            return self._device_step(identifier, variable, gradient)
        grad = np.asarray(gradient, dtype=np.float64)
        step = self._steps.get(identifier, 1)
        first = self._momentums.get(identifier)
        second = self._velocities.get(identifier)
        if first is None:
            first = np.zeros(grad.shape)
            second = np.zeros(grad.shape)
        first = self.beta1 * first + (1 - self.beta1) * grad
        second = self.beta2 * second + (1 - self.beta2) * grad ** 2
        first_hat = first / (1 - self.beta1 ** step)
        second_hat = second / (1 - self.beta2 ** step)
        variable -= self.learning_rate * (first_hat / np.sqrt(second_hat + self.epsilon))
        self._steps[identifier] = step + 1
        self._momentums[identifier] = first
        self._velocities[identifier] = second
        return variable
