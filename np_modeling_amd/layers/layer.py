"""Layer protocol -- same call semantics as reference layers/layer.py:11-69.

``layer(*inputs)`` runs ``forward``; ``layer(dy, backprop=True, learning_rate=lr)`` or
``layer(dy, backprop=True, optimizer_=opt)`` runs ``backward`` with the optimizer passed
positionally last.  The first call of either kind triggers ``initialize(*args)``.
Parameters are drawn on the HOST with NumPy's global RNG in the reference's order, so a
seeded run builds bit-identical parameters, then live in HBM as ``DeviceArray``.
"""

from __future__ import annotations

import abc
from typing import Optional, Sequence

import numpy as np

from np_modeling_amd import device as D
from np_modeling_amd import optimizer


class Layer(metaclass=abc.ABCMeta):
    def __init__(self, name: str = '', *args, **kwargs):
        self._name = name
        self._initialized = False

    def initialize(self, *args, **kwargs) -> None:
        pass

    @abc.abstractmethod
    def forward(self, *args, **kwargs):
        pass

    @abc.abstractmethod
    def backward(self, *args, optimizer_, **kwargs):
        pass

    def __call__(self, *args, backprop: bool = False, learning_rate: Optional[float] = None,
                 optimizer_: Optional['optimizer.Optimizer'] = None, **kwargs):
        # reference layer.py:33-45 -- lazy initialisation fires on the first call of any kind
        if not self._initialized:
            self.initialize(*args, **kwargs)
            self._initialized = True
        if not backprop:
            return self.forward(*args, **kwargs)
        if learning_rate is not None and optimizer_ is not None:
            raise ValueError('Optimizer and learning rate cannot both be specified!')
        if learning_rate is not None:
            optimizer_ = optimizer.SGDOptimizer(learning_rate)
        return self.backward(*args, optimizer_, **kwargs)

    @property
    def name(self):
        return self._name

    # -- helpers shared by the device layers ------------------------------------------
    def _param(self, attribute: str) -> D.DeviceArray:
        """Current value of a parameter as a DeviceArray.  Tests and weight binders assign
        arbitrary array-likes into the private attributes (reference layers/utils.py:52-88);
        those are moved to the device on first use and the attribute is rebound."""
        value = getattr(self, attribute)
        if not isinstance(value, D.DeviceArray):
            value = D.as_device(value)
            setattr(self, attribute, value)
        return value


class Initializer(metaclass=abc.ABCMeta):
    def __call__(self, shape: Sequence[int]) -> np.ndarray:
        pass


class RandomInitializer(Initializer):
    """N(0, 1) clipped to [-1, 1] in fp32 from the global NumPy RNG (reference layer.py:57-60)."""

    def __call__(self, shape: Sequence[int]) -> np.ndarray:
        sample = np.random.normal(size=shape).astype(np.float32)
        return np.clip(sample, -1.0, 1.0)


class StatefulLayer(Layer):
    """``initializer`` is the FIRST positional argument, as in reference layer.py:64-69."""

    def __init__(self, initializer: Optional[Initializer] = None, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._initializer = initializer or RandomInitializer()

    def _new_param(self, shape: Sequence[int]) -> D.DeviceArray:
        return D.as_device(self._initializer(list(shape)))
