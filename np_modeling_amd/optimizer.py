"""Optimizers under the reference's call contract (reference optimizer.py:12-69).

Contract, as the layers use it: ``optimizer_.update(layer, '_w', grad)`` once per parameter at the end of
``backward``; the optimizer fetches the parameter with ``getattr``, changes it IN PLACE (aliases taken before the
backward must see the new values) and stores it back with ``setattr``.  The reference's own, unchanged
``optimizer.py`` drives the device layers just as well (``DeviceArray`` implements ``lr * grad`` and ``-=``); the
classes here are the same contract for machines where the reference is not importable, with

* SGD as one device ``axpy`` per parameter (``lr * grad`` stays symbolic until ``-=`` consumes it), and
* Adam as one kernel per parameter (``npm_adam_step``) on fp64 moments resident in HBM, with the reference's
  numerics: bias-corrected moments, epsilon INSIDE the square root, state keyed by ``f'{id(obj)}.{attribute}'``.
  Host arrays take an equivalent NumPy path.
"""

from __future__ import annotations

import numpy as np


class Optimizer:
    """``update`` is what layers call; subclasses implement ``update_variable``."""

    def update(self, obj: object, attribute: str, gradient) -> None:
        slot = f'{id(obj)}.{attribute}'
        setattr(obj, attribute, self.update_variable(slot, getattr(obj, attribute), gradient))

    def update_variable(self, identifier: str, variable, gradient):
        """Apply one step to ``variable`` in place and return it."""
        raise NotImplementedError


class SGDOptimizer(Optimizer):
    def __init__(self, learning_rate: float):
        self._learning_rate = learning_rate

    def update_variable(self, identifier, variable, gradient):
        variable -= self._learning_rate * gradient        # device operands: a single axpy kernel
        return variable


class _HostMoments:
    """Adam state of one parameter on the host (fp64, like the reference's np.zeros defaults)."""

    def __init__(self, shape):
        self.first = np.zeros(shape)
        self.second = np.zeros(shape)


class _DeviceMoments:
    """Adam state of one parameter in HBM: two fp64 vectors."""

    def __init__(self, count: int):
        from np_modeling_amd import _C, device as D
        self.count = count
        self.first, self.second = D._Buffer(8 * count), D._Buffer(8 * count)
        for buf in (self.first, self.second):
            _C.check(_C.lib().npm_fill_f64(buf.ptr, 0.0, count), 'npm_fill_f64')


class AdamOptimizer(Optimizer):
    """Adam as reference optimizer.py:36-69 computes it.  Positional order of the constructor: learning rate,
    beta1, beta2, epsilon."""

    def __init__(self, learning_rate: float, beta1: float = 0.9, beta2: float = 0.999, epsilon: float = 1e-7):
        self.learning_rate, self.beta1, self.beta2, self.epsilon = learning_rate, beta1, beta2, epsilon
        self._state = {}          # identifier -> [next step number (from 1), moments]

    def _entry(self, identifier, make):
        entry = self._state.get(identifier)
        if entry is None or not make(entry[1]):
            entry = self._state[identifier] = [1, make(None)]
        return entry

    def update_variable(self, identifier, variable, gradient):
        from np_modeling_amd import device as D
        on_device = isinstance(variable, D.DeviceArray) and isinstance(gradient, (D.DeviceArray, D.Scaled)) \
            and gradient.size == variable.size
        if on_device:
            return self._step_on_device(identifier, variable, gradient)
        return self._step_on_host(identifier, variable, np.asarray(gradient, dtype=np.float64))

    def _step_on_device(self, identifier, variable, gradient):
        from np_modeling_amd import _C, device as D
        if isinstance(gradient, D.Scaled):
            gradient = gradient.materialize()
        count = variable.size

        def make(old):          # reuse matching state, else (re)create it
            if old is None:
                return _DeviceMoments(count)
            return isinstance(old, _DeviceMoments) and old.count == count

        entry = self._entry(identifier, make)
        moments = entry[1]
        _C.check(_C.lib().npm_adam_step(variable.ptr, gradient.ptr, moments.first.ptr, moments.second.ptr, count,
                                        float(self.learning_rate), float(self.beta1), float(self.beta2),
                                        float(self.epsilon), int(entry[0])), 'npm_adam_step')
        entry[0] += 1
        return variable

    def _step_on_host(self, identifier, variable, grad):
        def make(old):
            if old is None:
                return _HostMoments(grad.shape)
            return isinstance(old, _HostMoments) and old.first.shape == grad.shape

        entry = self._entry(identifier, make)
        step, moments = entry
        b1, b2 = self.beta1, self.beta2
        moments.first = b1 * moments.first + (1.0 - b1) * grad
        moments.second = b2 * moments.second + (1.0 - b2) * np.square(grad)
        unbiased_first = moments.first / (1.0 - b1 ** step)
        unbiased_second = moments.second / (1.0 - b2 ** step)
        variable -= self.learning_rate * (unbiased_first / np.sqrt(unbiased_second + self.epsilon))
        entry[0] = step + 1
        return variable
