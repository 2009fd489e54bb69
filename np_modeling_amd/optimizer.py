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

Inside ``device.coalesced_updates()`` (what ``parallel.GradScope`` wraps a backward's deferred updates in) the launches
of parameters that are neighbours in memory -- a layer's ``device.ParamArena`` -- are joined: ONE axpy / ONE Adam kernel
per encoder step.  For that the Adam moments of a parameter live at the parameter's own
offset inside two fp64 buffers that mirror the pool block the parameter is a view of, so neighbours in the arena are
neighbours in the moments too; ``update`` is still called once per parameter and the state is still keyed per
``id(obj).attribute``.
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


class _BlockMoments:
    """fp64 first / second moments for EVERY element of one pool block (zero-initialised): a parameter that is a view
    of the block -- a slice of a ParamArena, one of the packed wq / wk / wv -- has its moments at the same element offset."""

    def __init__(self, block):
        from np_modeling_amd import _C, device as D
        self.count = max(block.nbytes // 4, 1)
        self.first, self.second = D._Buffer(8 * self.count), D._Buffer(8 * self.count)
        for buf in (self.first, self.second):
            _C.check(_C.lib(ordered=False).npm_fill_f64(buf.ptr, 0.0, self.count), 'npm_fill_f64')       # fresh buffers: no queued update touches them


class _DeviceMoments:
    """Adam state of one parameter in HBM: ``count`` doubles at ``first_ptr`` / ``second_ptr`` inside ``owner``."""

    def __init__(self, count: int, owner: _BlockMoments, offset: int):
        self.count, self.owner = count, owner
        self.first_ptr, self.second_ptr = owner.first.ptr + 8 * offset, owner.second.ptr + 8 * offset


class AdamOptimizer(Optimizer):
    """Adam as reference optimizer.py:36-69 computes it.  Positional order of the constructor: learning rate,
    beta1, beta2, epsilon."""

    def __init__(self, learning_rate: float, beta1: float = 0.9, beta2: float = 0.999, epsilon: float = 1e-7):
        self.learning_rate, self.beta1, self.beta2, self.epsilon = learning_rate, beta1, beta2, epsilon
        self._state = {}          # identifier -> [next step number (from 1), moments]
        self._blocks = None       # pool block -> _BlockMoments (weak keys: the moments go when the block does)

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

    def _moments_for(self, variable, old):
        """The moments of ``variable`` at its offset inside its block's moment buffers.  A parameter that moved (into an
        arena; rebound to an array of the same size) takes its moments along."""
        import weakref
        from np_modeling_amd import _C
        if self._blocks is None:
            self._blocks = weakref.WeakKeyDictionary()
        block = variable._buf
        owner = self._blocks.get(block)
        if owner is None:
            owner = self._blocks[block] = _BlockMoments(block)
        new = _DeviceMoments(variable.size, owner, (variable.ptr - block.ptr) // 4)
        if isinstance(old, _DeviceMoments) and old.count == new.count:
            if old.first_ptr == new.first_ptr and old.owner is owner:
                return old
            if new.count:
                _C.check(_C.lib().npm_d2d(new.first_ptr, old.first_ptr, 8 * new.count), 'npm_d2d')
                _C.check(_C.lib().npm_d2d(new.second_ptr, old.second_ptr, 8 * new.count), 'npm_d2d')
        return new

    def _step_on_device(self, identifier, variable, gradient):
        from np_modeling_amd import _C, device as D
        if isinstance(gradient, D.Scaled):
            gradient = gradient.materialize()
        entry = self._state.get(identifier)
        if entry is None or not isinstance(entry[1], _DeviceMoments) or entry[1].count != variable.size:
            entry = self._state[identifier] = [1, None]             # fresh state (also after a host step or a resize)
        moments = entry[1] = self._moments_for(variable, entry[1])
        hyper = (float(self.learning_rate), float(self.beta1), float(self.beta2), float(self.epsilon), int(entry[0]))
        queue = D.UpdateQueue.active
        if queue is not None:                    # inside device.coalesced_updates(): joined with its neighbours at the end
            queue.adam(variable, gradient, moments.first_ptr, moments.second_ptr, hyper, moments.owner)
        else:
            _C.check(_C.lib().npm_adam_step(variable.ptr, gradient.ptr, moments.first_ptr, moments.second_ptr, variable.size,
                                            *hyper), 'npm_adam_step')
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
