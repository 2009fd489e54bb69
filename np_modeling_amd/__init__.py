"""np_modeling_amd -- the layer forward/backward hot path of levendlee/np-modeling on MI355X.

Same ``Layer.forward / backward / params`` surface as the reference's ``layers`` package,
computed by hand-written HIP kernels for gfx950 through a C ABI (``include/npm_hip.h``).
There is no CPU path: constructing device state without the built library or without a
GPU raises.

    import np_modeling_amd
    np_modeling_amd.install()          # optional: `from layers import mlp` now resolves here
    from np_modeling_amd import layers, optimizer, loss, train
"""

from __future__ import annotations

import sys

__version__ = '0.1.0'

from np_modeling_amd import _C, device, parallel          # noqa: E402,F401
from np_modeling_amd import optimizer, layers, loss, train  # noqa: E402,F401
from np_modeling_amd.device import DeviceArray, as_device, synchronize  # noqa: E402,F401
from np_modeling_amd._C import set_math, get_math, last_math  # noqa: E402,F401
from np_modeling_amd.layers.normalizations import set_dropout_rng  # noqa: E402,F401


def install(include_support_modules: bool = False) -> None:
    """Make the reference's import names resolve to this package: ``layers`` (and its
    submodules) always; ``optimizer``, ``loss`` and ``train`` only on request, because the
    reference's own unchanged versions of those drive the device layers as they are."""
    sys.modules['layers'] = layers
    for sub in ('layer', 'mlp', 'activations', 'normalizations', 'attentions', 'conv', 'transformer'):
        sys.modules['layers.' + sub] = getattr(layers, sub)
    if include_support_modules:
        sys.modules['optimizer'] = optimizer
        sys.modules['loss'] = loss
        sys.modules['train'] = train
