"""Device-resident arrays: chain channel -> receiver -> DSP calls without crossing PCIe.

The reference's cupy twin converts back to numpy at every function boundary (``cp.asnumpy``,
optic/models/modelsGPU.py:271, 501-509; optic/dsp/coreGPU.py:72).  Here a field may stay in HBM
between calls:

    Ed = oa.to_device(E)                     # one upload
    Ed = oa.manakovSSF(Ed, paramCh)          # DeviceArray in -> DeviceArray out
    Sd = oa.pdmCoherentReceiver(Ed, Elo, paramFE, paramPD)
    Sd = oa.edc(Sd, paramEDC)
    S = oa.decimate(Sd, paramDec).get()      # one download

A DeviceArray is a typed, C-contiguous block of device memory (``ssf_device_malloc``); the C ABI
recognises device pointers wherever it takes array arguments (include/ssf.h), so the same entry
points serve both kinds of caller.  Functions return a DeviceArray when their main input is one."""
import ctypes as C
import math
import threading

import numpy as np

from . import _lib


# Freed blocks are kept (by device and size) and handed out again: hipMalloc / hipFree of tens of MiB cost
# hundreds of microseconds and hipFree synchronises the device.  Capped; release_pool() returns everything.
_POOL = {}
_POOL_BYTES = [0]
_POOL_CAP = 4 << 30
_POOL_LOCK = threading.Lock()


# Transfers between host and device memory made through DeviceArray.set / .get (tests count them: a chain of calls on
# DeviceArrays must not go through the host in between, tests/test_chain.py, tests/test_gpu_device_arrays.py)
_COUNTS = {"h2d": 0, "d2h": 0, "h2d_bytes": 0, "d2h_bytes": 0}


def transfer_counts():
    """Copy of the counters of DeviceArray.set (h2d) / .get (d2h) calls and bytes since the package was imported."""
    return dict(_COUNTS)


def release_pool():
    """Give the cached device blocks back to the driver."""
    lib = _lib.load()
    with _POOL_LOCK:
        for (dev, _), ptrs in _POOL.items():
            for p in ptrs:
                lib.ssf_device_free(dev, p)
        _POOL.clear()
        _POOL_BYTES[0] = 0


class DeviceArray:
    """C-contiguous ndarray-like block of HBM on one GPU.  ``get()`` / ``np.asarray`` download it."""

    def __init__(self, shape, dtype, device=None):
        from .models import _state
        self.shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.dtype = np.dtype(dtype)
        self.device = _state["device"] if device is None else int(device)
        self._ptr = C.c_void_p()
        self._alloc = max(self.nbytes, 1)
        with _POOL_LOCK:
            cached = _POOL.get((self.device, self._alloc))
            if cached:
                self._ptr = cached.pop()
                _POOL_BYTES[0] -= self._alloc
                return
        lib = _lib.load()
        rc = lib.ssf_device_malloc(self.device, self._alloc, C.byref(self._ptr))
        if rc == -3 and _POOL:                      # out of memory: drop the cache and retry once
            release_pool()
            rc = lib.ssf_device_malloc(self.device, self._alloc, C.byref(self._ptr))
        _lib.raise_for(lib, None, rc)

    # ---- ndarray-like surface
    @property
    def size(self):
        return math.prod(self.shape)                 # (np.prod costs microseconds per call: several per receiver-side call)

    @property
    def nbytes(self):
        return self.size * self.dtype.itemsize

    @property
    def ndim(self):
        return len(self.shape)

    def __len__(self):
        return self.shape[0]

    @property
    def ptr(self):
        return self._ptr

    def reshape(self, *shape):
        """Same memory, another C-contiguous shape (a view: keeps the owner alive)."""
        shape = shape[0] if len(shape) == 1 and isinstance(shape[0], (tuple, list)) else shape
        known = math.prod(int(s) for s in shape if s != -1)
        shape = tuple(self.size // known if s == -1 else int(s) for s in shape)
        if math.prod(shape) != self.size:
            raise ValueError(f"cannot reshape array of size {self.size} into shape {shape}")
        v = object.__new__(DeviceArray)
        v.shape, v.dtype, v.device, v._ptr, v._owner = shape, self.dtype, self.device, self._ptr, self
        return v

    # ---- transfers
    def set(self, x):
        x = np.ascontiguousarray(x, dtype=self.dtype)
        if x.shape != self.shape:
            raise ValueError(f"shape mismatch: {x.shape} vs {self.shape}")
        lib = _lib.load()
        _lib.raise_for(lib, None, lib.ssf_device_memcpy(self.device, self._ptr, x.ctypes.data_as(C.c_void_p), self.nbytes))
        _COUNTS["h2d"] += 1
        _COUNTS["h2d_bytes"] += self.nbytes
        return self

    def get(self):
        out = np.empty(self.shape, dtype=self.dtype)
        lib = _lib.load()
        _lib.raise_for(lib, None, lib.ssf_device_memcpy(self.device, out.ctypes.data_as(C.c_void_p), self._ptr, self.nbytes))
        _COUNTS["d2h"] += 1
        _COUNTS["d2h_bytes"] += self.nbytes
        return out

    def __array__(self, dtype=None, copy=None):
        a = self.get()
        return a if dtype is None else a.astype(dtype)

    def copy(self):
        out = DeviceArray(self.shape, self.dtype, self.device)
        lib = _lib.load()
        _lib.raise_for(lib, None, lib.ssf_device_memcpy(self.device, out._ptr, self._ptr, self.nbytes))
        return out

    def __del__(self):
        if getattr(self, "_owner", None) is None and getattr(self, "_ptr", None):
            try:
                with _POOL_LOCK:
                    keep = _POOL_BYTES[0] + self._alloc <= _POOL_CAP
                    if keep:
                        _POOL.setdefault((self.device, self._alloc), []).append(self._ptr)
                        _POOL_BYTES[0] += self._alloc
                if not keep:
                    _lib.load().ssf_device_free(self.device, self._ptr)
            except Exception:
                pass

    def __repr__(self):
        return f"DeviceArray(shape={self.shape}, dtype={self.dtype.name}, device={self.device})"


def to_device(x, device=None):
    """Upload a numpy array (made C-contiguous) and return the DeviceArray."""
    x = np.ascontiguousarray(x)
    return DeviceArray(x.shape, x.dtype, device).set(x)


def is_device(x):
    return isinstance(x, DeviceArray)


def arg(x, dtype):
    """(pointer, keepalive) of an array argument of the C ABI: numpy arrays are made contiguous in
    ``dtype``; a DeviceArray must already have it (converting would be a hidden device round trip)."""
    if isinstance(x, DeviceArray):
        from .models import _state
        if x.dtype != np.dtype(dtype):
            raise TypeError(f"device array has dtype {x.dtype.name}, this call needs {np.dtype(dtype).name}")
        if x.device != _state["device"]:
            raise ValueError(f"device array lives on GPU {x.device}, the selected device is {_state['device']} (set_device)")
        return x.ptr, x
    a = np.ascontiguousarray(x, dtype=dtype)
    return a.ctypes.data_as(C.c_void_p), a


def empty(like_device, shape, dtype, device=None):
    """Output buffer of the kind the caller works with."""
    return DeviceArray(shape, dtype, device) if like_device else np.empty(shape, dtype=dtype)


def out_ptr(x):
    return x.ptr if isinstance(x, DeviceArray) else x.ctypes.data_as(C.c_void_p)
