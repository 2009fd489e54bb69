"""Sequential trainer with the interface and the printed output of reference train.py:13-46 (``Step:`` / ``Loss:``
lines, which the golden loss trajectories are parsed from).  Activations and gradients stay on the device from the
first layer to the loss and back; only the scalar loss comes to the host."""

from __future__ import annotations

import logging
from typing import Optional, Sequence

from np_modeling_amd import loss as losses
from np_modeling_amd.layers.layer import Layer

log = logging.getLogger(__name__)


class Trainer:
    def __init__(self, layers: Sequence[Layer], loss_: Optional['losses.Loss'] = None):
        self._loss = losses.MSELoss() if loss_ is None else loss_
        self._layers = layers

    def _predict(self, batch):
        for stage in self._layers:
            log.debug('forward: %s', stage.name)
            batch = stage(batch)
        return batch

    def _backpropagate(self, optimizer_) -> None:
        grad = self._loss(backprop=True)
        for stage in self._layers[::-1]:
            log.debug('backward: %s', stage.name)
            grad = stage(grad, backprop=True, optimizer_=optimizer_)

    def train(self, inputs, targets, steps: int, optimizer_) -> None:
        inputs, targets = self._resident(inputs), self._resident(targets)     # one upload for the whole loop
        for index in range(steps):
            print('Step: ', index)
            print('Loss: ', self._loss(self._predict(inputs), targets))
            self._backpropagate(optimizer_)

    @staticmethod
    def _resident(array):
        """A device copy of a host array, made ONCE per call (the arguments are fixed for the duration of a call)."""
        from np_modeling_amd import device as D
        return D.as_device(array)

    def eval(self, inputs, targets) -> None:
        print('Loss: ', self._loss(self._predict(inputs), targets))
