"""Conv2D, NHWC x HWIO, 'SAME' padding, stride 1, odd kernel (reference layers/conv.py:11-194).

Forward, input gradient and filter gradient are implicit-im2col GEMMs on the fp32 MFMA
(the k*k shifted matmuls of conv.py:101-105,189-193 collapsed into one K = k*k*C0
contraction; the padded copy of conv.py:97 is never built -- out-of-image taps read zero).
The reference computes in fp64 (np.zeros default dtype); this layer is fp32 end to end.
"""

from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

from np_modeling_amd import _C
from np_modeling_amd import device as D
from np_modeling_amd import parallel
from np_modeling_amd.layers import activations, layer


class Conv2D(layer.StatefulLayer):
    """Conv2D followed by an activation, ReLU by default."""

    def __init__(self, channels: int, kernel_size: int, padding: str = 'SAME',
                 strides: Sequence[int] = (1, 1), activation: Optional[activations.Activation] = None,
                 *args, **kwargs):
        super().__init__(*args, **kwargs)
        assert padding == 'SAME'
        assert strides == (1, 1)
        self._output_channels = channels
        self._kernel_size = kernel_size
        self._activation = activation or activations.ReLU()

    def initialize(self, x) -> None:
        # x is NHWC, filters are HWIO; draw order w then b (conv.py:38-42)
        self._input_channels = x.shape[-1]
        k = self._kernel_size
        self._w = self._new_param([k, k, self._input_channels, self._output_channels])
        self._b = self._new_param([self._output_channels])
        self._pack_parameters([[(self, '_b')], [(self, '_w')]])
        self._activation.initialize()

    def _fused_relu(self) -> bool:
        return type(self._activation) is activations.ReLU

    def forward(self, x):
        x = D.as_device(x)
        w, b = self._param('_w'), self._param('_b')
        n, h, wd, c0 = x.shape
        k, k2, wc0, c1 = w.shape
        assert k == k2 and wc0 == c0
        assert k % 2
        self._x = x
        fused = self._fused_relu()
        y = D.empty([n, h, wd, c1])
        pre = D.empty([n, h, wd, c1]) if fused else None
        desc = _C.npm_conv2d(n=n, h=h, w=wd, c_in=c0, c_out=c1, ksize=k, x=x.ptr, filt=w.ptr, bias=b.ptr,
                             y=y.ptr, pre=pre.ptr if fused else None, relu=int(fused))
        flops = 2.0 * n * h * wd * c1 * k * k * c0
        with D._timed('conv2d_fwd', flops):
            _C.check(_C.lib().npm_conv2d_fwd(C.byref(desc)), 'npm_conv2d_fwd')
        if fused:
            self._activation._x = pre
            return y
        return self._activation.forward(y)

    def backward(self, dy, optimizer_):
        dy = D.as_device(dy)
        x, w = self._x, self._param('_w')
        assert dy.shape[:3] == x.shape[:3]
        assert dy.shape[3] == self._output_channels
        n, h, wd, c0 = x.shape
        k, c1 = self._kernel_size, self._output_channels
        with parallel.grad_scope(w.size + c1 + 8, self._arena) as scope:
            db = scope.take([c1], owner=(self, '_b'))
            dw = scope.take(w.shape, owner=(self, '_w'))
            flops = 2.0 * n * h * wd * c1 * k * k * c0
            if self._fused_relu():
                # relu' (activations.py:19), db and dw (conv.py:54-56) in one call: the mask is applied where the
                # filter-gradient kernel stages its dy tiles; g = relu'(dy) comes back for the grad_x convolution
                g = D.empty(dy.shape)
                with D._timed('conv2d_bwd_w', flops, nbytes=4.0 * (3 * g.size + x.size)):
                    _C.check(_C.lib().npm_conv2d_bwd_w_relu(dy.ptr, self._activation._x.ptr, x.ptr, g.ptr, dw.ptr, db.ptr,
                                                           n, h, wd, c0, c1, k), 'npm_conv2d_bwd_w_relu')
            else:
                g = D.as_device(self._activation.backward(dy))
                D.colsum(g, n * h * wd, c1, out=db)
                with D._timed('conv2d_bwd_w', flops):
                    _C.check(_C.lib().npm_conv2d_bwd_w(g.ptr, x.ptr, dw.ptr, n, h, wd, c0, c1, k), 'npm_conv2d_bwd_w')
            dx = D.empty(x.shape)
            with D._timed('conv2d_bwd_x', flops):
                _C.check(_C.lib().npm_conv2d_bwd_x(g.ptr, w.ptr, dx.ptr, n, h, wd, c0, c1, k), 'npm_conv2d_bwd_x')
            assert dx.shape == x.shape
            scope.defer(optimizer_, self, '_w', dw)
            scope.defer(optimizer_, self, '_b', db)
        return dx

    @property
    def w(self):
        assert self._initialized
        return self._param('_w')

    @property
    def b(self):
        assert self._initialized
        return self._param('_b')
