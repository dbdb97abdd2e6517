"""Device context and device-resident images (host side of include/stitching_amd.h).

`DeviceImage` stands where the reference hands numpy arrays / cv.UMat across the cv2
boundary (stitching/blender.py:41, stitching/warper.py:45-51): it exposes `.shape`,
`.dtype`, `__array__` (host copy on demand) and rectangular slicing
(stitching/cropper.py:150-151) while the pixels stay in HBM.
"""
import ctypes as C
import os
import threading

import numpy as np

from . import _lib
from .stitching_error import StitchingError

_DTYPES = {_lib.U8: np.uint8, _lib.S16: np.int16, _lib.F32: np.float32}
_ELEMS = {np.dtype(np.uint8): _lib.U8, np.dtype(np.int16): _lib.S16, np.dtype(np.float32): _lib.F32}


class Context:
    """One per (process, GPU): owns the HIP stream and the caching allocator."""

    def __init__(self, device=0):
        self._lib = _lib.lib()
        h = C.c_void_p()
        _lib.check(self._lib.stx_ctx_create(int(device), C.byref(h)))
        self.handle = h
        self.device = int(device)

    def sync(self):
        _lib.check(self._lib.stx_ctx_sync(self.handle))

    # ---- measurement hooks (bench.py)
    def prof_enable(self, on=True):
        _lib.check(self._lib.stx_prof_enable(self.handle, 1 if on else 0))

    def prof_reset(self):
        _lib.check(self._lib.stx_prof_reset(self.handle))

    def prof_results(self):
        n = C.c_int()
        _lib.check(self._lib.stx_prof_count(self.handle, C.byref(n)))
        out = []
        for i in range(n.value):
            name = C.create_string_buffer(64)
            calls, ms, by = C.c_int64(), C.c_double(), C.c_double()
            _lib.check(self._lib.stx_prof_get(self.handle, i, name, 64, C.byref(calls), C.byref(ms), C.byref(by)))
            out.append({"kernel": name.value.decode(), "calls": calls.value, "total_ms": ms.value,
                        "algo_bytes": by.value})
        return out

    def mark(self, slot):
        _lib.check(self._lib.stx_mark(self.handle, int(slot)))

    def elapsed_ms(self, a, b):
        ms = C.c_double()
        _lib.check(self._lib.stx_mark_elapsed_ms(self.handle, int(a), int(b), C.byref(ms)))
        return ms.value

    def close(self):
        if self.handle:
            self._lib.stx_ctx_destroy(self.handle)
            self.handle = None


_ctx_lock = threading.Lock()
_contexts = {}
_default_device = None


def device_count():
    n = C.c_int()
    _lib.check(_lib.lib().stx_device_count(C.byref(n)))
    return n.value


def set_default_device(device):
    global _default_device
    _default_device = int(device)


def get_context(device=None):
    """Process-wide context for `device` (default: STITCHING_AMD_DEVICE, else LOCAL_RANK, else 0)."""
    global _default_device
    if device is None:
        if _default_device is None:
            _default_device = int(os.environ.get("STITCHING_AMD_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        device = _default_device
    with _ctx_lock:
        ctx = _contexts.get(device)
        if ctx is None:
            ctx = _contexts[device] = Context(device)
        return ctx


class _PinnedBlock:
    """Owns one page-locked host allocation and exposes it through the array interface: numpy arrays made from it
    keep it alive as their .base."""

    def __init__(self, nbytes):
        self._lib = _lib.lib()
        p = C.c_void_p()
        _lib.check(self._lib.stx_host_alloc(int(nbytes), C.byref(p)))
        self.ptr = p
        self.__array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (p.value, False), "version": 3}

    def __del__(self):
        try:
            if self.ptr:
                self._lib.stx_host_free(self.ptr)
                self.ptr = None
        except Exception:
            pass


def pinned_empty(shape, dtype=np.uint8):
    """numpy array in page-locked host memory: decode frames into it (or np.copyto) and the upload in
    Warper.warp_image / DeviceImage.from_numpy runs at PCIe rate; DeviceImage.numpy(out=such an array) likewise."""
    dt = np.dtype(dtype)
    count = int(np.prod(shape))
    raw = np.asarray(_PinnedBlock(max(count * dt.itemsize, 1)))
    return raw[:count * dt.itemsize].view(dt).reshape(shape)


class DeviceImage:
    """A 2-D image in HBM (u8 / int16 / fp32, 1..4 interleaved channels)."""

    __array_priority__ = 100

    def __init__(self, ctx, handle):
        self.ctx = ctx
        self._h = handle
        info = (C.c_int64 * 6)()
        _lib.check(ctx._lib.stx_buf_info(handle, info))
        self.width, self.height, self.channels, self._elem = int(info[0]), int(info[1]), int(info[2]), int(info[3])
        self.stride_bytes = int(info[4])

    # ---- construction
    @classmethod
    def from_numpy(cls, arr, ctx=None, wait=True):
        """Upload.  wait=False: `arr` is page-locked (pinned_empty) and stays untouched until ctx.sync(); the copy is only
        queued (pageable arrays are copied synchronously either way)."""
        ctx = ctx or get_context()
        a = np.asarray(arr)
        if a.ndim == 2:
            h, w, c = a.shape[0], a.shape[1], 1
        elif a.ndim == 3:
            h, w, c = a.shape
        else:
            raise StitchingError(f"expected an HxW or HxWxC image, got shape {a.shape}")
        if a.dtype not in _ELEMS:
            raise StitchingError(f"unsupported image dtype {a.dtype}")
        if h == 0 or w == 0:
            raise StitchingError("empty image")
        if not a.flags["C_CONTIGUOUS"]:
            a = np.ascontiguousarray(a)
        out = C.c_void_p()
        if not wait and a is not arr and not np.shares_memory(a, arr):
            wait = True  # a temporary copy was made above: it must not be freed under the queued transfer
        fn = ctx._lib.stx_buf_from_host if wait else ctx._lib.stx_buf_from_host_async
        _lib.check(fn(ctx.handle, a.ctypes.data_as(C.c_void_p), a.strides[0], w, h, c, _ELEMS[a.dtype], C.byref(out)))
        return cls(ctx, out)

    # ---- numpy-like surface
    @property
    def shape(self):
        return (self.height, self.width) if self.channels == 1 else (self.height, self.width, self.channels)

    @property
    def dtype(self):
        return np.dtype(_DTYPES[self._elem])

    @property
    def ndim(self):
        return 2 if self.channels == 1 else 3

    @property
    def size(self):
        return self.height * self.width * self.channels

    def numpy(self, out=None, wait=True):
        """Host copy; `out`: a C-contiguous array of this shape / dtype to copy into (e.g. from pinned_empty).
        wait=False (needs a page-locked `out`): the read-back is only queued; `out` is valid after ctx.sync()."""
        if out is None:
            out = np.empty(self.shape, self.dtype)
            wait = True
        elif out.shape != self.shape or out.dtype != self.dtype or not out.flags.c_contiguous:
            raise StitchingError(f"out must be a C-contiguous {self.dtype} array of shape {self.shape}")
        fn = self.ctx._lib.stx_buf_to_host if wait else self.ctx._lib.stx_buf_to_host_async
        _lib.check(fn(self._h, out.ctypes.data_as(C.c_void_p), out.strides[0]))
        return out

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a if dtype is None else a.astype(dtype, copy=False)

    def __len__(self):
        return self.height

    def __getitem__(self, key):
        """Rectangular, unit-step slicing on (y, x) returns a device view (no copy)."""
        if not isinstance(key, tuple):
            key = (key,)
        if len(key) > 2 and not (len(key) == 3 and key[2] == slice(None)):
            return self.numpy()[key]
        ys = key[0] if len(key) > 0 else slice(None)
        xs = key[1] if len(key) > 1 else slice(None)
        if not isinstance(ys, slice) or not isinstance(xs, slice):
            return self.numpy()[key]
        y0, y1, ystep = ys.indices(self.height)
        x0, x1, xstep = xs.indices(self.width)
        if ystep != 1 or xstep != 1 or y1 <= y0 or x1 <= x0:
            return self.numpy()[key]
        out = C.c_void_p()
        _lib.check(self.ctx._lib.stx_buf_view(self._h, x0, y0, x1 - x0, y1 - y0, C.byref(out)))
        return DeviceImage(self.ctx, out)

    def device_ptr(self):
        p = C.c_void_p()
        _lib.check(self.ctx._lib.stx_buf_device_ptr(self._h, C.byref(p)))
        return p.value

    def free(self):
        if self._h is not None:
            if self.ctx.handle:  # a closed context has already returned all of its device memory
                self.ctx._lib.stx_buf_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def __repr__(self):
        return f"DeviceImage(shape={self.shape}, dtype={self.dtype}, device={self.ctx.device})"


def as_device(img, ctx=None, wait=True):
    """numpy array, cv.UMat or DeviceImage -> DeviceImage on `ctx` (uploads when needed; wait: see DeviceImage.from_numpy).
    A cv.UMat is what the reference's own classes hand on where cv2 produced one — the seam masks of `SeamFinder.resize`
    (stitching/seam_finder.py:37-43) reach `Blender.feed` that way when only Warper and Blender are switched — its array comes out of
    `.get()`."""
    if isinstance(img, DeviceImage):
        return img
    if not isinstance(img, np.ndarray) and hasattr(img, "get"):
        img = img.get()
    return DeviceImage.from_numpy(img, ctx, wait)

