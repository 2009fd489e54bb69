"""The call protocol every device layer speaks (what reference layers/layer.py:11-69 defines).

One entry point, three behaviours:

* ``layer(x, ...)``                                   -> ``forward``
* ``layer(dy, backprop=True, learning_rate=lr)``      -> ``backward`` with a throw-away SGD optimizer
* ``layer(dy, backprop=True, optimizer_=opt)``        -> ``backward`` with that optimizer (``None`` is allowed)

``backward`` always receives the optimizer as its LAST positional argument.  Whatever the first call is, it first
runs ``initialize`` with the very same arguments (lazy shapes).  Parameters are drawn on the HOST from NumPy's global
RNG in the reference's order -- a seeded run builds bit-identical parameters -- and then live in HBM.
"""

from __future__ import annotations

import abc
from typing import Any, Optional, Sequence

import numpy as np

from np_modeling_amd import device as D
from np_modeling_amd import optimizer as optimizers

_BOTH_GIVEN = 'Optimizer and learning rate cannot both be specified!'


def _pick_optimizer(learning_rate: Optional[float], optimizer_: Any):
    """The optimizer a backward call runs with (reference layer.py:38-44)."""
    if learning_rate is None:
        return optimizer_
    if optimizer_ is not None:
        raise ValueError(_BOTH_GIVEN)
    return optimizers.SGDOptimizer(learning_rate)


class Layer(abc.ABC):
    """Base of all layers; concrete layers provide ``forward`` and ``backward`` (and usually ``initialize``)."""

    def __init__(self, name: str = '', *_ignored, **_ignored_kw):
        self._initialized = False
        self._name = name

    @property
    def name(self) -> str:
        return self._name

    # ---- what a concrete layer implements ---------------------------------------------------------
    def initialize(self, *inputs, **kwargs) -> None:
        """Create parameters from the first call's arguments.  Stateless layers keep this no-op."""

    @abc.abstractmethod
    def forward(self, *inputs, **kwargs):
        raise NotImplementedError

    @abc.abstractmethod
    def backward(self, *grads, optimizer_, **kwargs):
        raise NotImplementedError

    # ---- dispatch -----------------------------------------------------------------------------------------
    def __call__(self, *inputs, backprop: bool = False, learning_rate: Optional[float] = None,
                 optimizer_: Optional['optimizers.Optimizer'] = None, **kwargs):
        if not self._initialized:                       # first call of ANY kind (a backward call included)
            self.initialize(*inputs, **kwargs)
            self._initialized = True
        if backprop:
            return self.backward(*inputs, _pick_optimizer(learning_rate, optimizer_), **kwargs)
        return self.forward(*inputs, **kwargs)

    # ---- shared by the device layers --------------------------------------------------------------------
    _arena = None           # device.ParamArena of this layer's parameters (built by layers that have several)

    def _pack_parameters(self, segments) -> None:
        """Move the parameters named by ``segments`` (lists of (layer, attribute), in the order ``backward`` produces their
        gradients) into one device.ParamArena, so that the deferred updates of a backward are one launch.  Called where
        no outside alias of a parameter can be stale afterwards: from ``initialize`` or at the end of the first forward."""
        if D.COALESCE_UPDATES:
            self._arena = D.ParamArena(segments)
            # a sub-layer whose parameters all moved here would keep its own (now dead) arena block alive -- a second copy
            # of every parameter, deep-copied along with the layer
            for obj in {id(o): o for segment in segments for o, _ in segment}.values():
                if obj is not self and obj._arena is not None and not obj._arena.live():
                    obj._arena = None

    def _param(self, attribute: str) -> D.DeviceArray:
        """Current value of a parameter as a DeviceArray.  Tests and weight binders assign arbitrary array-likes
        into the private attributes (reference layers/utils.py:52-88); such a value moves to the device on first
        use and the attribute is rebound to the device copy."""
        value = getattr(self, attribute)
        if isinstance(value, D.DeviceArray):
            return value
        on_device = D.as_device(value)
        setattr(self, attribute, on_device)
        return on_device


class Initializer(abc.ABC):
    """Maps a shape to a host array of initial values."""

    def __call__(self, shape: Sequence[int]) -> np.ndarray:
        return None


class RandomInitializer(Initializer):
    """Standard normal draws from the GLOBAL NumPy generator, fp32, clipped to [-1, 1] (reference layer.py:57-60)."""

    def __call__(self, shape: Sequence[int]) -> np.ndarray:
        return np.clip(np.random.normal(size=shape).astype(np.float32), -1.0, 1.0)


class StatefulLayer(Layer):
    """A layer with parameters.  NB the constructor takes ``initializer`` as its FIRST positional argument and the
    name after it, as reference layer.py:64-69 does."""

    def __init__(self, initializer: Optional[Initializer] = None, *args, **kwargs):
        Layer.__init__(self, *args, **kwargs)
        self._initializer = initializer if initializer is not None else RandomInitializer()

    def _new_param(self, shape: Sequence[int]) -> D.DeviceArray:
        return D.as_device(self._initializer(list(shape)))
