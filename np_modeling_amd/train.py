"""Sequential trainer with the reference's interface (reference train.py:13-46)."""

from __future__ import annotations

import logging
from typing import Optional, Sequence

from np_modeling_amd import loss as loss_lib
from np_modeling_amd import optimizer
from np_modeling_amd.layers import layer

_LOG = logging.getLogger(__name__)


class Trainer:
    def __init__(self, layers: Sequence[layer.Layer], loss_: Optional[loss_lib.Loss] = None):
        self._layers = layers
        self._loss = loss_ or loss_lib.MSELoss()

    def _forward(self, inputs):
        activation = inputs
        for layer_ in self._layers:
            _LOG.debug('forward %s', layer_.name)
            activation = layer_(activation)
        return activation

    def train(self, inputs, targets, steps: int, optimizer_: optimizer.Optimizer) -> None:
        for step in range(steps):
            print('Step: ', step)
            value = self._loss(self._forward(inputs), targets)
            print('Loss: ', value)
            grad = self._loss(backprop=True)
            for layer_ in reversed(self._layers):
                _LOG.debug('backward %s', layer_.name)
                grad = layer_(grad, backprop=True, optimizer_=optimizer_)

    def eval(self, inputs, targets) -> None:
        print('Loss: ', self._loss(self._forward(inputs), targets))
