"""``DeviceArray``: an fp32 tensor resident in MI355X HBM, plus typed wrappers over the C ABI.

The reference passes ``np.ndarray`` everywhere and relies on three behaviours that a
device tensor has to keep:

* parameters are updated IN PLACE by the optimizer (``variable -= lr * gradient``,
  reference optimizer.py:32) and tests hold aliases to them taken before ``backward``
  (reference layers/mlp_test.py:50-51,93-94) -> ``__isub__`` mutates device memory and
  returns ``self``; ``np.asarray(alias)`` always reads the current device contents;
* array-likes are assigned straight into private attributes (reference
  layers/utils.py:52-88) -> layers convert lazily with :func:`as_device`;
* ``copy.deepcopy(layer)`` (reference layers/attentions_test.py:72) -> ``__deepcopy__``.

Anything that is not on the hot path (losses, Adam's fp64 moment math) works through
``__array__``: the value is copied to the host and NumPy does the arithmetic.
"""

from __future__ import annotations

import ctypes as C
import math
import numbers
import os
from typing import Optional, Sequence, Tuple, Union

import numpy as np

from np_modeling_amd import _C

Shape = Tuple[int, ...]


def _prod(shape: Sequence[int]) -> int:
    return int(math.prod(int(s) for s in shape))


class _Buffer:
    """Owns one pool block; returns it to the stream-ordered pool when unreferenced."""

    __slots__ = ('ptr', 'nbytes', '__weakref__')

    def __init__(self, nbytes: int):
        lib = _C.lib(ordered=False)              # an allocation reads and writes no array
        out = C.c_void_p()
        _C.check(lib.npm_malloc(C.byref(out), max(int(nbytes), 4)), 'npm_malloc')
        self.ptr = out.value
        self.nbytes = int(nbytes)

    def __del__(self):
        ptr, self.ptr = getattr(self, 'ptr', None), None
        if ptr and _C._LIB is not None:
            try:
                _C._LIB.npm_free(ptr)
            except Exception:       # interpreter shutdown
                pass

    def __deepcopy__(self, memo):
        """A block of its own with the same bytes, ONCE per deepcopy (``memo``): everything that shared this block
        shares the copy.  (The default protocol would duplicate the owner of ``ptr``: two frees of one block.)"""
        key = ('np_modeling_amd._Buffer', id(self))
        buf = memo.get(key)
        if buf is None:
            buf = _Buffer(self.nbytes)
            if self.nbytes:
                _C.check(_C.lib().npm_d2d(buf.ptr, self.ptr, self.nbytes), 'npm_d2d')
            memo[key] = buf
            memo[id(buf)] = buf                  # keeps `buf` alive for the duration of the deepcopy
        return buf

    def __copy__(self):
        raise TypeError('a device block has one owner: share the _Buffer object or deepcopy it')

    def __reduce__(self):
        raise TypeError('device memory cannot be pickled: copy the array to the host (numpy()) first')


class Scaled:
    """``alpha * array`` kept symbolic so that ``variable -= lr * gradient`` is ONE axpy
    kernel instead of a temporary plus a subtraction.  Materialises on any other use."""

    __slots__ = ('array', 'alpha')

    def __init__(self, array: 'DeviceArray', alpha: float):
        self.array = array
        self.alpha = float(alpha)

    @property
    def shape(self):
        return self.array.shape

    @property
    def size(self):
        return self.array.size

    dtype = np.dtype(np.float32)

    def materialize(self) -> 'DeviceArray':
        out = empty(self.array.shape)
        _C.check(_C.lib().npm_scale(self.array.ptr, out.ptr, self.alpha, self.array.size), 'npm_scale')
        return out

    def __array__(self, dtype=None, copy=None):
        host = self.array.numpy() * np.float32(self.alpha)
        return host if dtype is None else host.astype(dtype)

    def __mul__(self, other):
        if isinstance(other, numbers.Real):
            return Scaled(self.array, self.alpha * float(other))
        return np.asarray(self) * other

    __rmul__ = __mul__

    def __neg__(self):
        return Scaled(self.array, -self.alpha)


class DeviceArray:
    """Contiguous row-major fp32 tensor in device memory (a view into a pool block)."""

    __slots__ = ('_buf', 'ptr', 'shape', '__weakref__')
    dtype = np.dtype(np.float32)
    __array_priority__ = 100.0

    def __init__(self, shape: Sequence[int], _buf: Optional[_Buffer] = None, _ptr: Optional[int] = None):
        self.shape = tuple(int(s) for s in shape)
        if _buf is None:
            _buf = _Buffer(4 * _prod(self.shape))
            _ptr = _buf.ptr
        self._buf = _buf
        self.ptr = _ptr

    # ---- metadata ---------------------------------------------------------------
    @property
    def size(self) -> int:
        return _prod(self.shape)

    @property
    def ndim(self) -> int:
        return len(self.shape)

    @property
    def nbytes(self) -> int:
        return 4 * self.size

    def __len__(self) -> int:
        if not self.shape:
            raise TypeError('len() of unsized object')
        return self.shape[0]

    def __repr__(self) -> str:
        return f'DeviceArray(shape={self.shape}, dtype=float32, ptr=0x{self.ptr:x})'

    # ---- views --------------------------------------------------------------------
    def reshape(self, *shape) -> 'DeviceArray':
        if len(shape) == 1 and not isinstance(shape[0], numbers.Integral):
            shape = tuple(shape[0])
        shape = [int(s) for s in shape]
        if shape.count(-1) > 1:
            raise ValueError('can only specify one unknown dimension')
        if -1 in shape:
            known = _prod([s for s in shape if s != -1])
            shape[shape.index(-1)] = self.size // known if known else 0
        if _prod(shape) != self.size:
            raise ValueError(f'cannot reshape array of size {self.size} into shape {tuple(shape)}')
        return DeviceArray(shape, self._buf, self.ptr)

    def flat_view(self, offset: int, shape: Sequence[int]) -> 'DeviceArray':
        """View of ``prod(shape)`` elements starting ``offset`` elements into this array."""
        n = _prod(shape)
        if offset < 0 or offset + n > self.size:
            raise ValueError('flat_view out of range')
        return DeviceArray(shape, self._buf, self.ptr + 4 * int(offset))

    # ---- host <-> device ----------------------------------------------------------
    def numpy(self) -> np.ndarray:
        out = np.empty(self.shape, dtype=np.float32)
        if self.size:
            _C.check(_C.lib().npm_d2h(out.ctypes.data, self.ptr, out.nbytes), 'npm_d2h')
        return out

    def __array__(self, dtype=None, copy=None):
        host = self.numpy()
        return host if dtype is None or np.dtype(dtype) == np.float32 else host.astype(dtype)

    def set(self, value) -> 'DeviceArray':
        host = np.ascontiguousarray(np.broadcast_to(np.asarray(value, dtype=np.float32), self.shape))
        if host.size:
            _C.check(_C.lib().npm_h2d(self.ptr, host.ctypes.data, host.nbytes), 'npm_h2d')
        return self

    def copy(self) -> 'DeviceArray':
        out = DeviceArray(self.shape)
        if self.size:
            _C.check(_C.lib().npm_d2d(out.ptr, self.ptr, self.nbytes), 'npm_d2d')
        return out

    def __copy__(self):
        return self.copy()

    def __deepcopy__(self, memo):
        """Copies the OWNING pool block once per deepcopy (through ``memo``) and rebases this view into the copy:
        views that shared a block still share one afterwards -- the packed q/k/v projection parameters stay
        adjacent, and pointer views used as strided GEMM operands keep everything they address."""
        buf = self._buf.__deepcopy__(memo)
        return DeviceArray(self.shape, buf, buf.ptr + (self.ptr - self._buf.ptr))

    def astype(self, dtype, copy=True):
        if np.dtype(dtype) == np.float32:
            return self.copy() if copy else self
        return self.numpy().astype(dtype)

    # ---- in-place updates (the optimizer contract) -----------------------------------
    def _axpy(self, other, alpha: float) -> 'DeviceArray':
        if isinstance(other, Scaled):
            alpha, other = alpha * other.alpha, other.array
        if isinstance(other, numbers.Real):
            other = full(self.shape, float(other))
        elif not isinstance(other, DeviceArray) or other.size != self.size:
            host = np.asarray(other, dtype=np.float32)
            other = from_host(np.broadcast_to(host, self.shape))
        queue = UpdateQueue.active
        if queue is not None:                       # inside coalesced_updates(): launched, merged with its neighbours, at the end
            queue.axpy(self, other, float(alpha))
            return self
        _C.check(_C.lib().npm_axpy(self.ptr, other.ptr, float(alpha), self.size), 'npm_axpy')
        return self

    def __isub__(self, other):
        return self._axpy(other, -1.0)

    def __iadd__(self, other):
        return self._axpy(other, 1.0)

    def __imul__(self, other):
        if isinstance(other, numbers.Real):
            _C.check(_C.lib().npm_scale(self.ptr, self.ptr, float(other), self.size), 'npm_scale')
            return self
        return self.set(self.numpy() * np.asarray(other))

    # ---- arithmetic ---------------------------------------------------------------------
    def __mul__(self, other):
        if isinstance(other, numbers.Real):
            return Scaled(self, float(other))
        return self.numpy() * np.asarray(other)

    __rmul__ = __mul__

    def __neg__(self):
        return Scaled(self, -1.0)

    def __add__(self, other):
        if isinstance(other, DeviceArray) and other.shape == self.shape:
            out = empty(self.shape)
            _C.check(_C.lib().npm_add(self.ptr, other.ptr, out.ptr, self.size), 'npm_add')
            return out
        return self.numpy() + np.asarray(other)

    __radd__ = __add__

    def __sub__(self, other):
        return self.numpy() - np.asarray(other)

    def __rsub__(self, other):
        return np.asarray(other) - self.numpy()

    def __truediv__(self, other):
        if isinstance(other, numbers.Real):
            return Scaled(self, 1.0 / float(other))
        return self.numpy() / np.asarray(other)

    def __rtruediv__(self, other):
        return np.asarray(other) / self.numpy()

    def __pow__(self, other):
        return self.numpy() ** other

    def __getitem__(self, idx):
        return self.numpy()[idx]

    def __iter__(self):
        return iter(self.numpy())

    def sum(self, *a, **k):
        return self.numpy().sum(*a, **k)

    def mean(self, *a, **k):
        return self.numpy().mean(*a, **k)

    def max(self, *a, **k):
        return self.numpy().max(*a, **k)

    def min(self, *a, **k):
        return self.numpy().min(*a, **k)

    def __eq__(self, other):
        return self.numpy() == np.asarray(other)

    def __ne__(self, other):
        return self.numpy() != np.asarray(other)

    def __lt__(self, other):
        return self.numpy() < np.asarray(other)

    def __le__(self, other):
        return self.numpy() <= np.asarray(other)

    def __gt__(self, other):
        return self.numpy() > np.asarray(other)

    def __ge__(self, other):
        return self.numpy() >= np.asarray(other)

    __hash__ = None


ArrayLike = Union[DeviceArray, np.ndarray, Sequence]


# ---- constructors ------------------------------------------------------------------------
# ---- parameters back to back, updates in one launch ------------------------------------------------------------------
COALESCE_UPDATES = os.environ.get('NPM_COALESCE_UPDATES', '1') != '0'   # A/B switch: 0 = one optimizer launch per parameter (rounds 1-4)
_ALIGN = 4                                                              # floats: 16-byte aligned slices (GEMM / DMA operands)


class _LayerRef:
    """Weak reference to a layer that survives ``copy.deepcopy`` of the structure holding it by pointing at the COPY of
    the layer (through the deepcopy memo).  A ParamArena is an attribute of the layer whose parameters it lists: strong
    references would make every layer a reference cycle, and its device memory would wait for the cycle collector."""
    __slots__ = ('_ref',)

    def __init__(self, obj):
        import weakref
        self._ref = weakref.ref(obj)

    def __call__(self):
        return self._ref()

    def __deepcopy__(self, memo):
        import copy
        target = self._ref()
        return _LayerRef(copy.deepcopy(target, memo)) if target is not None else self


class ParamArena:
    """The parameters of one layer (a composite's sub-layers included) back to back in ONE device buffer, in the order
    their gradients are produced by ``backward`` -- so that the flat gradient bucket of parallel.GradScope mirrors it
    slot for slot and ``parameter -= lr * gradient`` (reference optimizer.py:26-33; Adam: :53-67) is ONE launch over the
    whole range instead of one per parameter.

    ``segments`` is a list of lists of ``(layer, attribute)``; the parameters of one segment stay gap-free (the packed
    wq / wk / wv of MultiHeadAttention must remain adjacent), every segment starts 16-byte aligned.  Building the arena
    COPIES the current values and rebinds the attributes to views of it, so it is built where no outside alias of a
    parameter can exist yet (inside ``initialize``) or where one would be re-taken anyway (end of a composite's first
    forward; the reference's tests take their aliases after the first call).  A parameter that is rebound later
    (weight binders assign arrays into the private attributes, reference layers/utils.py:52-88) simply leaves the
    arena: ``offset_of`` returns None for it and it takes the per-parameter path again."""

    def __init__(self, segments):
        self.entries = []                            # [layer, attribute, offset, size]
        offset = 0
        for segment in segments:
            offset = (offset + _ALIGN - 1) // _ALIGN * _ALIGN
            for obj, attribute in segment:
                value = obj._param(attribute)
                self.entries.append([_LayerRef(obj), attribute, offset, value.size])
                offset += value.size
        self.size = (offset + _ALIGN - 1) // _ALIGN * _ALIGN
        self.flat = zeros([max(self.size, 1)])       # zeros: the alignment gaps are updated along with their neighbours
        for ref, attribute, off, n in self.entries:
            obj = ref()
            old = obj._param(attribute)
            view = self.flat.flat_view(off, old.shape)
            if n:
                _C.check(_C.lib().npm_d2d(view.ptr, old.ptr, 4 * n), 'npm_d2d')
            setattr(obj, attribute, view)

    def offset_of(self, obj, attribute: str, size: int) -> Optional[int]:
        """Where (in floats) ``obj.attribute`` starts inside the arena if ``size`` elements from there are still what
        the attribute(s) hold -- None once a binder has rebound it."""
        for ref, a, off, n in self.entries:
            if a == attribute and ref() is obj:
                value = getattr(obj, attribute)
                if isinstance(value, DeviceArray) and value.ptr == self.flat.ptr + 4 * off and value._buf is self.flat._buf \
                        and off + size <= self.size:
                    return off
                return None
        return None

    def live(self) -> int:
        """How many of the parameters are still where the arena put them."""
        return sum(ref() is not None and self.offset_of(ref(), a, n) == off for ref, a, off, n in self.entries)


class _Pending:
    """One queued update: ``n`` elements at ``var`` and ``grad`` (byte addresses inside the blocks ``var_buf`` / ``grad_buf``),
    side arrays ``extra`` = ((byte address, bytes per element), ...) and ``key`` = everything two updates must agree on to
    be one launch (kind, step size, Adam's hyper-parameters and step number)."""
    __slots__ = ('var', 'grad', 'n', 'key', 'var_buf', 'grad_buf', 'extra')

    def __init__(self, var, grad, n, key, var_buf, grad_buf, extra=()):
        self.var, self.grad, self.n, self.key, self.var_buf, self.grad_buf, self.extra = var, grad, n, key, var_buf, grad_buf, extra

    def continued_by(self, other: '_Pending') -> bool:
        """``other`` starts where this ends -- or up to 3 floats later, the alignment padding between two slices of an
        arena, which the blocks own on every side and which may be updated along (nothing reads it) -- on the parameter
        side, the gradient side and every side array alike."""
        gap = other.var - (self.var + 4 * self.n)
        if gap < 0 or gap > 4 * (_ALIGN - 1) or gap % 4 or self.key != other.key:
            return False
        if self.var_buf is not other.var_buf or self.grad_buf is not other.grad_buf or len(self.extra) != len(other.extra):
            return False
        if other.grad - (self.grad + 4 * self.n) != gap:
            return False
        return all(width == w2 and ptr2 - (ptr + width * self.n) == gap // 4 * width
                   for (ptr, width), (ptr2, w2) in zip(self.extra, other.extra))


class UpdateQueue:
    """Parameter updates issued inside ``coalesced_updates()`` are collected instead of launched; on exit, updates whose
    parameters AND gradients (AND Adam moments) are neighbours in memory -- a ParamArena and the gradient bucket that
    mirrors it -- run as one launch over the joined range.  The optimizer is not involved: the reference's unchanged
    ``SGDOptimizer.update_variable`` (``variable -= lr * gradient``, optimizer.py:32) reaches ``DeviceArray._axpy``
    exactly as before, once per ``Optimizer.update(obj, attribute, grad)``, with its ``id(obj).attribute`` keying.
    Elementwise updates do not care where a range is cut: results are bit-identical to the per-parameter launches.

    Program order is kept for everything an observer can tell apart (any ``Optimizer`` subclass may run in here, not only
    the shipped ones): queued updates are launched sorted by address, so only updates that touch DISJOINT memory wait in
    the queue together -- one that writes what a queued one reads or writes, or reads what a queued one writes (weight
    decay after the step, a momentum buffer updated and then applied), first drains the queue; and so does every library
    call that is not part of the queue (``*=``, ``numpy()``, ``set``, any kernel: ``_C.lib()`` calls ``_C._ORDER_HOOK``)."""

    active: Optional['UpdateQueue'] = None

    def __init__(self):
        self._pending = []
        self._keep = []
        self.launches = 0        # kernels launched by run()
        self.updates = 0         # updates they stand for
        self.drains = 0          # times the queue was emptied early to keep program order

    def _enqueue(self, item: '_Pending') -> None:
        def overlap(a_lo, a_n, b_lo, b_n):
            return a_lo < b_lo + 4 * b_n and b_lo < a_lo + 4 * a_n
        for p in self._pending:
            if overlap(item.var, item.n, p.var, p.n) or overlap(item.var, item.n, p.grad, p.n) or overlap(item.grad, item.n, p.var, p.n):
                self.drains += 1
                self.run()
                break
        self._pending.append(item)
        _C._ORDER_HOOK = self._drain

    def _drain(self) -> None:
        if self._pending:
            self.drains += 1
            self.run()

    def axpy(self, var: 'DeviceArray', grad: 'DeviceArray', alpha: float) -> None:
        self._keep.append(grad)                     # a temporary (``lr * host_array``) must outlive the deferred launch
        self._enqueue(_Pending(var.ptr, grad.ptr, var.size, ('axpy', alpha), var._buf, grad._buf))

    def adam(self, var: 'DeviceArray', grad: 'DeviceArray', first_ptr: int, second_ptr: int, hyper: tuple, owner) -> None:
        """``hyper`` = (lr, beta1, beta2, epsilon, step); ``owner`` keeps the moment buffers alive until run()."""
        self._keep += [owner, grad]
        self._enqueue(_Pending(var.ptr, grad.ptr, var.size, ('adam',) + tuple(hyper), var._buf, grad._buf,
                               ((first_ptr, 8), (second_ptr, 8))))

    def run(self) -> None:
        _C._ORDER_HOOK = None                       # the launches below are the queue itself
        lib = _C.lib(ordered=False)
        pending, self._pending = sorted(self._pending, key=lambda u: (u.key[0], u.var)), []
        self.updates += len(pending)
        run: Optional[_Pending] = None
        for item in pending + [None]:
            if run is not None and item is not None and run.continued_by(item):
                run.n = (item.var - run.var) // 4 + item.n
                continue
            if run is not None and run.n:
                if run.key[0] == 'axpy':
                    _C.check(lib.npm_axpy(run.var, run.grad, run.key[1], run.n), 'npm_axpy')
                else:
                    lr, b1, b2, eps, step = run.key[1:]
                    _C.check(lib.npm_adam_step(run.var, run.grad, run.extra[0][0], run.extra[1][0], run.n, lr, b1, b2, eps,
                                               step), 'npm_adam_step')
                self.launches += 1
            run = item
        self._keep = []


class coalesced_updates:
    """``with coalesced_updates():`` -- see UpdateQueue.  Nests: an inner context joins the outer one."""

    def __enter__(self) -> UpdateQueue:
        self._mine = UpdateQueue.active is None and COALESCE_UPDATES
        if self._mine:
            UpdateQueue.active = UpdateQueue()
        return UpdateQueue.active

    def __exit__(self, exc_type, exc, tb) -> bool:
        if self._mine:
            queue, UpdateQueue.active = UpdateQueue.active, None
            _C._ORDER_HOOK = None
            if exc_type is None:
                queue.run()
        return False


def empty(shape: Sequence[int]) -> DeviceArray:
    return DeviceArray(shape)


def full(shape: Sequence[int], value: float) -> DeviceArray:
    out = DeviceArray(shape)
    if out.size:
        _C.check(_C.lib().npm_fill_f32(out.ptr, float(value), out.size), 'npm_fill_f32')
    return out


def zeros(shape: Sequence[int]) -> DeviceArray:
    return full(shape, 0.0)


def from_host(value) -> DeviceArray:
    host = np.ascontiguousarray(np.asarray(value, dtype=np.float32))
    out = DeviceArray(host.shape)
    if host.size:
        _C.check(_C.lib().npm_h2d(out.ptr, host.ctypes.data, host.nbytes), 'npm_h2d')
    return out


class ByteBuffer:
    """Raw device bytes (dropout masks)."""

    __slots__ = ('_buf', 'ptr', 'nbytes')

    def __init__(self, nbytes: int):
        self._buf = _Buffer(nbytes)
        self.ptr, self.nbytes = self._buf.ptr, int(nbytes)

    def numpy(self) -> np.ndarray:
        host = np.empty([self.nbytes], dtype=np.uint8)
        if self.nbytes:
            _C.check(_C.lib().npm_d2h(host.ctypes.data, self.ptr, self.nbytes), 'npm_d2h')
        return host

    def __deepcopy__(self, memo):
        out = ByteBuffer.__new__(ByteBuffer)
        out._buf = self._buf.__deepcopy__(memo)
        out.ptr, out.nbytes = out._buf.ptr, self.nbytes
        return out


def bytes_from_host(value: np.ndarray) -> ByteBuffer:
    host = np.ascontiguousarray(value).view(np.uint8)
    out = ByteBuffer(host.nbytes)
    if host.nbytes:
        _C.check(_C.lib().npm_h2d(out.ptr, host.ctypes.data, host.nbytes), 'npm_h2d')
    return out


def as_device(value) -> DeviceArray:
    """DeviceArray as is; ``Scaled`` materialised; anything else (np.ndarray, jax Array,
    nested lists) copied to the device as contiguous fp32."""
    if isinstance(value, DeviceArray):
        return value
    if isinstance(value, Scaled):
        return value.materialize()
    return from_host(value)


def synchronize() -> None:
    _C.check(_C.lib().npm_sync(), 'npm_sync')


def pool_stats() -> Tuple[int, int]:
    used, reserved = C.c_size_t(), C.c_size_t()
    _C.check(_C.lib().npm_pool_stats(C.byref(used), C.byref(reserved)))
    return used.value, reserved.value


def trim_pool() -> None:
    """Hand the cached (free) blocks of the pool back to the driver (between workloads of different shapes)."""
    import gc
    gc.collect()
    _C.check(_C.lib().npm_pool_trim(), 'npm_pool_trim')


class Event:
    """HIP event on the compute stream (bench.py times kernels with these)."""

    def __init__(self):
        self._h = C.c_void_p()
        _C.check(_C.lib().npm_event_create(C.byref(self._h)), 'npm_event_create')

    def record(self) -> 'Event':
        _C.check(_C.lib().npm_event_record(self._h), 'npm_event_record')
        return self

    def synchronize(self) -> None:
        _C.check(_C.lib().npm_event_sync(self._h), 'npm_event_sync')

    def elapsed_ms(self, end: 'Event') -> float:
        ms = C.c_float()
        _C.check(_C.lib().npm_event_elapsed_ms(self._h, end._h, C.byref(ms)), 'npm_event_elapsed_ms')
        return float(ms.value)

    def __del__(self):
        h, self._h = getattr(self, '_h', None), None
        if h and _C._LIB is not None:
            try:
                _C._LIB.npm_event_destroy(h)
            except Exception:
                pass


# ---- kernel wrappers -------------------------------------------------------------------------
FUSE_COLSUM = os.environ.get('NPM_FUSE_COLSUM', '1') != '0'      # A/B switches (see gemm, attentions.py)
FUSE_SOFTMAX_BWD = os.environ.get('NPM_FUSE_SOFTMAX_BWD', '1') != '0'
FUSE_BSUM = os.environ.get('NPM_FUSE_BSUM', '1') != '0'          # bias gradient inside the weight-gradient GEMM
PACK_QKV = os.environ.get('NPM_PACK_QKV', '1') != '0'
ATTN_CORE = os.environ.get('NPM_ATTN_CORE', '1') != '0'            # fused attention core (npm_mha_core_*) where it applies
# Whether the forward keeps the raw scores for the backward.  The fp32 matrix rate is the scarce resource at the large
# head sizes (a 32 x 32 score tile costs 64 MFMAs to recompute at D = 128, 16 loads to read back: forward + backward of the C4
# core 6.6 ms with saved scores against 7.5 ms recomputing); the score tensor does not shrink with the head size while the
# matrix work does, so at small head sizes writing and reading it is what the kernels wait for (H D = 1024, B 256, S 512,
# forward + backward: D 64 7.6 saved / 7.9 ms recomputed, D 32 8.8 / 8.7, D 16 13.4 / 10.4: profiles/r04_attn_modes.log).
# Default: keep them from head size 64 up.  NPM_ATTN_SAVE_SCORES=1 / 0 forces either mode everywhere (0 is the memory-lean
# mode: log-sum-exp only, 2.1 GB less at C4 / C5).
_save = os.environ.get('NPM_ATTN_SAVE_SCORES', '')
ATTN_SAVE_SCORES: Optional[bool] = None if _save == '' else _save != '0'
ATTN_SAVE_SCORES_FROM = 64


def attn_save_scores(head_size: int) -> bool:
    return head_size >= ATTN_SAVE_SCORES_FROM if ATTN_SAVE_SCORES is None else bool(ATTN_SAVE_SCORES)


# masked attention: let the fused kernels skip tiles without an allowed position (NPM_ATTN_TILE_SKIP=0: visit them all)
ATTN_TILE_SKIP = os.environ.get('NPM_ATTN_TILE_SKIP', '1') != '0'
# head size 128: the attention backward's row terms (dctx . ctx per query and head) come out of the epilogue of the GEMM that
# produces dctx (NPM_EPI_ROWDOT) instead of a pass over dctx and ctx in front of the attention kernel (NPM_ATTN_ROWDOT=0: that pass)
ATTN_ROWDOT = os.environ.get('NPM_ATTN_ROWDOT', '1') != '0'

class KernelTimer:
    """Brackets every kernel-wrapper call with HIP events on the compute stream and books its
    ALGORITHMIC work (flops for GEMMs, bytes for the HBM-bound kernels; DESIGN.md has the
    per-unit figures).  bench.py uses it over the timed region for the roofline object."""

    def __init__(self):
        self.records = []          # (name, flops, bytes, start, stop)

    def __enter__(self):
        global _TIMER
        self._prev, _TIMER = _TIMER, self
        return self

    def __exit__(self, *exc):
        global _TIMER
        _TIMER = self._prev
        return False

    def summary(self):
        synchronize()
        out = {}
        for name, flops, nbytes, start, stop in self.records:
            rec = out.setdefault(name, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
            rec['launches'] += 1
            rec['ms'] += start.elapsed_ms(stop)
            rec['flops'] += flops
            rec['bytes'] += nbytes
        return out


_TIMER: Optional[KernelTimer] = None


class _timed:
    __slots__ = ('name', 'flops', 'bytes', 'start')

    def __init__(self, name: str, flops: float = 0.0, nbytes: float = 0.0):
        self.name, self.flops, self.bytes = name, flops, nbytes

    def __enter__(self):
        self.start = Event().record() if _TIMER is not None else None

    def __exit__(self, *exc):
        if self.start is not None and _TIMER is not None:
            _TIMER.records.append((self.name, self.flops, self.bytes, self.start, Event().record()))
        return False


class Mat:
    """Operand descriptor for :func:`gemm`: base pointer, row pitch, two batch strides."""

    __slots__ = ('ptr', 'ld', 's0', 's1', '_keep')

    def __init__(self, array_or_ptr, ld: int, s0: int = 0, s1: int = 0):
        self._keep = array_or_ptr          # keeps a temporary alive until the launch is queued
        self.ptr = array_or_ptr.ptr if isinstance(array_or_ptr, DeviceArray) else int(array_or_ptr)
        self.ld, self.s0, self.s1 = int(ld), int(s0), int(s1)


def gemm(m: int, n: int, k: int, a: Mat, b: Mat, c: Mat, *, trans_a: bool = False, trans_b: bool = False,
         batch: Tuple[int, int] = (1, 1), alpha: float = 1.0, bias: Optional[DeviceArray] = None,
         residual: Optional[Mat] = None, relu_save: Optional[Mat] = None, relu_mask: Optional[Mat] = None,
         split_k: int = 0, colsum_out: Optional[DeviceArray] = None,
         softmax_bwd: Optional[Tuple[Mat, DeviceArray]] = None,
         bsum_out: Optional[DeviceArray] = None, asum_out: Optional[DeviceArray] = None,
         rowdot: Optional[Tuple[Mat, DeviceArray, float]] = None) -> None:
    """C = epilogue(alpha * op(A) @ op(B)); see include/npm_hip.h ``npm_sgemm``.
    ``rowdot=(X, out, scale)``: besides C = A @ B, out[n // 128, m] (zeros on entry) += scale * sum over each block of 128
    columns of C * X -- the attention backward's row term dctx . ctx per head of size 128, taken where dctx is produced.
    ``colsum_out`` ([batch1, n]) receives the column sums of the stored C (a bias gradient
    taken in the producing GEMM's epilogue instead of a separate pass over C).
    ``softmax_bwd=(P, delta)``: C = alpha * P * (A @ B - delta[row]) -- the softmax backward with its
    row term precomputed (:func:`attn_rowdot`), fused into the GEMM that produces dP.
    ``bsum_out`` ([n]) receives the column sums of B ([k, n], not transposed): the bias gradient that goes
    with a weight gradient x^T @ dy, taken from the dy tiles that GEMM stages anyway; ``asum_out`` ([m]) the
    column sums of a transposed A ([k, m]) for products written dproj^T @ x."""
    g = _C.npm_gemm()
    g.trans_a, g.trans_b = int(trans_a), int(trans_b)
    g.m, g.n, g.k = int(m), int(n), int(k)
    g.batch0, g.batch1 = int(batch[0]), int(batch[1])
    g.a, g.lda, g.stride_a0, g.stride_a1 = a.ptr, a.ld, a.s0, a.s1
    g.b, g.ldb, g.stride_b0, g.stride_b1 = b.ptr, b.ld, b.s0, b.s1
    g.c, g.ldc, g.stride_c0, g.stride_c1 = c.ptr, c.ld, c.s0, c.s1
    g.alpha = float(alpha)
    epi = 0
    if bias is not None:
        epi |= _C.EPI_BIAS
        g.bias = bias.ptr
    if residual is not None:
        epi |= _C.EPI_RESIDUAL
        g.residual, g.ldr = residual.ptr, residual.ld
    if relu_save is not None:
        epi |= _C.EPI_RELU_SAVE
        g.aux, g.ldaux = relu_save.ptr, relu_save.ld
    if relu_mask is not None:
        epi |= _C.EPI_RELU_MASK
        g.aux, g.ldaux = relu_mask.ptr, relu_mask.ld
    if softmax_bwd is not None:
        epi |= _C.EPI_SOFTMAX_BWD
        g.aux, g.ldaux = softmax_bwd[0].ptr, softmax_bwd[0].ld
        g.rowvec = softmax_bwd[1].ptr
    if rowdot is not None:
        epi |= _C.EPI_ROWDOT
        g.aux, g.ldaux = rowdot[0].ptr, rowdot[0].ld
        g.rowdot, g.rowdot_scale = rowdot[1].ptr, float(rowdot[2])
    g.epilogue = epi
    g.split_k = int(split_k)
    fuse = colsum_out is not None and FUSE_COLSUM
    g.colsum = colsum_out.ptr if fuse else None
    fuse_b = bsum_out is not None and FUSE_BSUM
    g.bsum = bsum_out.ptr if fuse_b else None
    fuse_a = asum_out is not None and FUSE_BSUM
    g.asum = asum_out.ptr if fuse_a else None
    layout = 'TN' if trans_a else ('NT' if trans_b else 'NN')
    nb = batch[0] * batch[1]
    unique = 4.0 * nb * (m * k + k * n + m * n * (1 + (residual is not None) + (relu_save is not None) +
                                                  (relu_mask is not None) + (softmax_bwd is not None) + (rowdot is not None)))
    with _timed('sgemm_' + layout, flops=2.0 * m * n * k * nb, nbytes=unique):
        _C.check(_C.lib().npm_sgemm(C.byref(g)), 'npm_sgemm')
    if bsum_out is not None and not fuse_b:      # A/B switch: separate pass over B
        assert batch == (1, 1) and not trans_b
        with _timed('colsum', nbytes=4.0 * k * n):
            _C.check(_C.lib().npm_colsum(b.ptr, bsum_out.ptr, k, n, b.ld), 'npm_colsum')
    if asum_out is not None and not fuse_a:
        assert batch == (1, 1) and trans_a
        with _timed('colsum', nbytes=4.0 * k * m):
            _C.check(_C.lib().npm_colsum(a.ptr, asum_out.ptr, k, m, a.ld), 'npm_colsum')
    if colsum_out is not None and not fuse:      # A/B switch: separate pass over the stored C
        assert c.ld == n * batch[1] and (batch[1] == 1 or c.s1 == n), 'unfused colsum needs row-contiguous head slices'
        colsum(DeviceArray([1], c._keep._buf, c.ptr), m * batch[0], n * batch[1], out=colsum_out)


def colsum(x: DeviceArray, rows: int, cols: int, out: Optional[DeviceArray] = None) -> DeviceArray:
    out = empty([cols]) if out is None else out
    with _timed('colsum', nbytes=4.0 * rows * cols):
        _C.check(_C.lib().npm_colsum(x.ptr, out.ptr, rows, cols, cols), 'npm_colsum')
    return out


def add(a: DeviceArray, b: DeviceArray, out: Optional[DeviceArray] = None) -> DeviceArray:
    out = empty(a.shape) if out is None else out
    with _timed('add', nbytes=12.0 * a.size):
        _C.check(_C.lib().npm_add(a.ptr, b.ptr, out.ptr, a.size), 'npm_add')
    return out


def add3(a: DeviceArray, b: DeviceArray, c: DeviceArray, out: Optional[DeviceArray] = None) -> DeviceArray:
    out = empty(a.shape) if out is None else out
    with _timed('add3', nbytes=16.0 * a.size):
        _C.check(_C.lib().npm_add3(a.ptr, b.ptr, c.ptr, out.ptr, a.size), 'npm_add3')
    return out


def relu_fwd(x: DeviceArray, out: Optional[DeviceArray] = None) -> DeviceArray:
    out = empty(x.shape) if out is None else out
    with _timed('relu_fwd', nbytes=8.0 * x.size):
        _C.check(_C.lib().npm_relu_fwd(x.ptr, out.ptr, x.size), 'npm_relu_fwd')
    return out


def relu_bwd(x_pre: DeviceArray, dy: DeviceArray, out: Optional[DeviceArray] = None) -> DeviceArray:
    out = empty(dy.shape) if out is None else out
    with _timed('relu_bwd', nbytes=12.0 * dy.size):
        _C.check(_C.lib().npm_relu_bwd(x_pre.ptr, dy.ptr, out.ptr, dy.size), 'npm_relu_bwd')
    return out


def relu_bwd_colsum(x_pre: DeviceArray, dy: DeviceArray, cols: int, colsum_out: DeviceArray) -> DeviceArray:
    """ReLU backward of a [rows, cols] matrix and the column sums of the result (the bias gradient of the
    layer in front of the ReLU) in one pass: conv.py:54-55, mlp.py:74 + mlp.py:34."""
    out = empty(dy.shape)
    rows = dy.size // cols
    assert rows * cols == dy.size and x_pre.size == dy.size and colsum_out.size == cols
    with _timed('relu_bwd_colsum', nbytes=12.0 * dy.size):
        _C.check(_C.lib().npm_relu_bwd_colsum(x_pre.ptr, dy.ptr, out.ptr, colsum_out.ptr, rows, cols),
                 'npm_relu_bwd_colsum')
    return out


def softmax_fwd(x: DeviceArray, scale: float = 1.0, out: Optional[DeviceArray] = None) -> DeviceArray:
    n = x.shape[-1] if x.ndim else 1
    rows = x.size // n if n else 0
    out = empty(x.shape) if out is None else out
    with _timed('softmax_fwd', nbytes=8.0 * x.size):
        _C.check(_C.lib().npm_softmax_fwd(x.ptr, out.ptr, rows, n, float(scale)), 'npm_softmax_fwd')
    return out


def softmax_bwd(y: DeviceArray, dy: DeviceArray, scale: float = 1.0, out: Optional[DeviceArray] = None) -> DeviceArray:
    n = y.shape[-1] if y.ndim else 1
    rows = y.size // n if n else 0
    out = empty(y.shape) if out is None else out
    with _timed('softmax_bwd', nbytes=12.0 * y.size):
        _C.check(_C.lib().npm_softmax_bwd(y.ptr, dy.ptr, out.ptr, rows, n, float(scale)), 'npm_softmax_bwd')
    return out


def attn_rowdot(a: DeviceArray, b: DeviceArray) -> DeviceArray:
    """out[b, h, s] = sum_d a[b, s, h, d] * b[b, s, h, d]  (inputs [B, S, H, D])."""
    bsz, seq, heads, dim = a.shape
    out = empty([bsz, heads, seq])
    with _timed('attn_rowdot', nbytes=8.0 * a.size):
        _C.check(_C.lib().npm_attn_rowdot(a.ptr, b.ptr, out.ptr, bsz, seq, heads, dim), 'npm_attn_rowdot')
    return out


def layernorm_dropout_supported(d: int) -> bool:
    """Row lengths npm_layernorm_dropout_fwd / _bwd take (include/npm_hip.h): the row-in-registers kernels."""
    return d % 4 == 0 and d <= 4096


def layernorm_fwd(x: DeviceArray, gamma: DeviceArray, beta: DeviceArray, eps: float, drop=None):
    """Returns (z, mean, rstd); mean/rstd have x's shape with the last axis reduced to 1.  ``drop = (mask bytes, keep_prob)``:
    the row is DropOut's output, formed on the way in (the dropped tensor is never stored)."""
    d = x.shape[-1]
    rows = x.size // d
    stat_shape = tuple(x.shape[:-1]) + (1,)
    z, mean, rstd = empty(x.shape), empty(stat_shape), empty(stat_shape)
    if drop is not None:
        with _timed('layernorm_fwd', nbytes=9.0 * x.size + 8.0 * rows):
            _C.check(_C.lib().npm_layernorm_dropout_fwd(x.ptr, drop[0].ptr, float(drop[1]), gamma.ptr, beta.ptr, float(eps), rows, d,
                                                        z.ptr, mean.ptr, rstd.ptr), 'npm_layernorm_dropout_fwd')
        return z, mean, rstd
    with _timed('layernorm_fwd', nbytes=8.0 * x.size + 8.0 * rows):
        _C.check(_C.lib().npm_layernorm_fwd(x.ptr, gamma.ptr, beta.ptr, float(eps), rows, d,
                                            z.ptr, mean.ptr, rstd.ptr), 'npm_layernorm_fwd')
    return z, mean, rstd


def layernorm_bwd(dz: DeviceArray, x: DeviceArray, mean: DeviceArray, rstd: DeviceArray, gamma: DeviceArray,
                  dgamma: DeviceArray, dbeta: DeviceArray, residual: Optional[DeviceArray] = None, drop=None) -> DeviceArray:
    """``drop = (mask bytes, keep_prob)``: ``x`` is the input of the DropOut in front of the norm; the result is the gradient
    with respect to THAT (DropOut.backward applied on the way out, before the residual)."""
    d = x.shape[-1]
    rows = x.size // d
    dx = empty(dz.shape)
    res = None if residual is None else residual.ptr
    if drop is not None:
        with _timed('layernorm_bwd', nbytes=(17.0 if residual is not None else 13.0) * x.size + 8.0 * rows):
            _C.check(_C.lib().npm_layernorm_dropout_bwd(dz.ptr, x.ptr, drop[0].ptr, float(drop[1]), mean.ptr, rstd.ptr, gamma.ptr,
                                                        res, rows, d, dx.ptr, dgamma.ptr, dbeta.ptr), 'npm_layernorm_dropout_bwd')
        return dx
    with _timed('layernorm_bwd', nbytes=(16.0 if residual is not None else 12.0) * x.size + 8.0 * rows):
        _C.check(_C.lib().npm_layernorm_bwd(dz.ptr, x.ptr, mean.ptr, rstd.ptr, gamma.ptr, res, rows, d,
                                            dx.ptr, dgamma.ptr, dbeta.ptr), 'npm_layernorm_bwd')
    return dx


# ---- fused attention core ----------------------------------------------------------------------------
def mha_core_supported(head_dim: int, value_dim: Optional[int] = None, *, any_math: bool = False) -> bool:
    """Whether the fused attention kernels take this head size.  They run the exact-fp32 MFMA only, so under a
    split-bf16 math mode the layers keep composing attention from ``gemm`` (which honours the mode) unless
    ``any_math`` is set (masked attention exists only in the fused kernels)."""
    if not ATTN_CORE or (value_dim is not None and value_dim != head_dim):
        return False
    if not any_math and _C.current_math() != 'f32':
        return False
    return bool(_C.lib().npm_mha_core_supported(int(head_dim)))


class AttnMask:
    """A boolean attention mask on the device: bytes plus (batch, head, query) strides; broadcast axes have
    stride 0.  ``np.where(mask, scaled, -inf)`` of reference layers/attentions.py:105-107."""

    def __init__(self, mask, b: int, h: int, sq: int, skv: int):
        self.dims = (int(b), int(h), int(sq), int(skv))
        host = np.asarray(mask).astype(bool)
        while host.ndim < 4:
            host = host[None]
        if host.ndim != 4 or any(have not in (1, want) for have, want in zip(host.shape, (b, h, sq, skv))):
            raise AssertionError(f'mask shape {np.shape(mask)} does not broadcast to {(b, h, sq, skv)}')
        if host.shape[3] != skv:
            host = np.broadcast_to(host, host.shape[:3] + (skv,))
        host = np.ascontiguousarray(host).astype(np.uint8)
        self.host = host.astype(bool)
        self.buf = bytes_from_host(host)
        nb, nh, nq, _ = host.shape
        self.strides = (0 if nb == 1 else nh * nq * skv, 0 if nh == 1 else nq * skv, 0 if nq == 1 else skv)
        # Tile summary (include/npm_hip.h npm_mha_mask_summary): one byte per (32 queries, 128 keys) and distinct mask plane,
        # made on the device from the bytes just uploaded; the fused kernels skip the tiles it marks empty.
        self.summary, self.summary_strides = None, (0, 0)
        if ATTN_TILE_SKIP and sq > 0:
            nqt, nkb = (sq + 31) // 32, (skv + 127) // 128
            self.summary_all_offset = nb * nh * nqt * nkb          # "every position allowed" bytes: the second half
            self.summary = ByteBuffer(2 * self.summary_all_offset)
            _C.check(_C.lib().npm_mha_mask_summary(self.buf.ptr, self.strides[0], self.strides[1], self.strides[2], nb, nh, sq, skv,
                                                   self.summary.ptr), 'npm_mha_mask_summary')
            self.summary_strides = (0 if nb == 1 else nh * nqt * nkb, 0 if nh == 1 else nqt * nkb)

    def full(self, b, h, sq, skv) -> np.ndarray:
        return np.broadcast_to(self.host, (b, h, sq, skv))


def _core_desc(q: Mat, k: Mat, v: Mat, ctx: Mat, lse: DeviceArray, dims, scale: float,
               mask: Optional[AttnMask], scores: Optional[DeviceArray]):
    b, h, sq, skv, d = (int(x) for x in dims)
    c = _C.npm_mha_core()
    c.batch, c.heads, c.seq_q, c.seq_kv, c.head_dim = b, h, sq, skv, d
    c.scale = float(scale)
    c.q, c.q_pitch, c.k, c.k_pitch, c.v, c.v_pitch = q.ptr, q.ld, k.ptr, k.ld, v.ptr, v.ld
    c.ctx, c.ctx_pitch, c.lse = ctx.ptr, ctx.ld, lse.ptr
    if mask is not None:
        c.mask = mask.buf.ptr
        c.mask_stride_b, c.mask_stride_h, c.mask_stride_q = mask.strides
        if mask.summary is not None:
            c.tile_summary = mask.summary.ptr
            c.summary_stride_b, c.summary_stride_h = mask.summary_strides
            c.summary_all_offset = mask.summary_all_offset
    if scores is not None:
        c.scores = scores.ptr
    return c


def mha_core_fwd(q: Mat, k: Mat, v: Mat, dims, scale: float, mask: Optional[AttnMask] = None,
                 save_scores: bool = False):
    """ctx[b, i, h, :] = softmax_j(scale q_i . k_j [masked]) v_j in one kernel (include/npm_hip.h npm_mha_core_fwd).
    ``q``/``k``/``v``: (array, row pitch) of [B, S, H, D] operands.  Returns (ctx [B, Sq, H, D], lse [B, H, Sq],
    scores or None)."""
    b, h, sq, skv, d = dims
    ctx, lse = empty([b, sq, h, d]), empty([b, h, sq])
    scores = empty([b, h, sq, skv]) if save_scores else None
    c = _core_desc(q, k, v, Mat(ctx, h * d), lse, dims, scale, mask, scores)
    nbytes = 4.0 * b * h * d * (2 * sq + 2 * skv) + (4.0 * b * h * sq * skv if save_scores else 0.0)
    with _timed('mha_core_fwd', flops=4.0 * b * h * sq * skv * d, nbytes=nbytes):
        _C.check(_C.lib().npm_mha_core_fwd(C.byref(c)), 'npm_mha_core_fwd')
    return ctx, lse, scores


def mha_core_bwd(q: Mat, k: Mat, v: Mat, ctx: DeviceArray, lse: DeviceArray, dctx: DeviceArray,
                 dq: Mat, dk: Mat, dv: Mat, dims, scale: float, mask: Optional[AttnMask] = None,
                 scores: Optional[DeviceArray] = None, neg_delta: Optional[Tuple[DeviceArray, int, int]] = None) -> None:
    """dq, dk, dv of the attention core from q, k, v, the forward's ctx and lse, and dctx (npm_mha_core_bwd).
    Algorithmic work: the four products dP, dV, dK, dQ (the recomputed q.k is the kernel's own business).
    ``neg_delta=(array, stride_b, stride_h)``: the row terms -scale * (dctx . ctx) already taken by the GEMM that produced dctx."""
    b, h, sq, skv, d = dims
    c = _core_desc(q, k, v, Mat(ctx, h * d), lse, dims, scale, mask, scores)
    c.dctx, c.dctx_pitch = dctx.ptr, h * d
    c.dq, c.dq_pitch, c.dk, c.dk_pitch, c.dv, c.dv_pitch = dq.ptr, dq.ld, dk.ptr, dk.ld, dv.ptr, dv.ld
    if neg_delta is not None:
        c.neg_delta, c.neg_delta_stride_b, c.neg_delta_stride_h = neg_delta[0].ptr, int(neg_delta[1]), int(neg_delta[2])
    nbytes = 4.0 * b * h * d * (4 * sq + 4 * skv) + (4.0 * b * h * sq * skv if scores is not None else 0.0)
    with _timed('mha_core_bwd', flops=8.0 * b * h * sq * skv * d, nbytes=nbytes):
        _C.check(_C.lib().npm_mha_core_bwd(C.byref(c)), 'npm_mha_core_bwd')
