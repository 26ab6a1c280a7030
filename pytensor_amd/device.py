"""Device arrays on top of the C-ABI pool (``pthip_alloc``/``pthip_free``).

A :class:`DeviceArray` is a NumPy-style strided view ``(buffer, offset, shape,
strides-in-elements, dtype)`` of a pooled HBM allocation; views (``DimShuffle``,
``Subtensor``, ``ExtractDiag``, ``Reshape`` of contiguous data) are host-side
descriptor arithmetic and launch no kernel, exactly as the reference's
``DimShuffle.c_code`` builds a strided view (pytensor/tensor/elemwise.py:186-256).
"""

from __future__ import annotations

import ctypes as C

import numpy as np

from pytensor_amd import ffi


class Buffer:
    """Owner of one pooled device allocation (returned to the pool on GC)."""

    __slots__ = ("ptr", "nbytes", "__weakref__")

    def __init__(self, nbytes: int):
        p = C.c_void_p()
        ffi.check(ffi.lib().pthip_alloc(max(int(nbytes), 1), C.byref(p)))
        self.ptr = p.value
        self.nbytes = int(nbytes)

    def __del__(self):
        try:
            if self.ptr:
                ffi.lib().pthip_free(self.ptr)
                self.ptr = 0
        except Exception:  # interpreter shutdown
            pass


def contiguous_strides(shape):
    st = []
    acc = 1
    for s in reversed(shape):
        st.append(acc)
        acc *= max(int(s), 1)
    return tuple(reversed(st))


class DeviceArray:
    __slots__ = ("buf", "offset", "shape", "strides", "dtype", "size")

    def __init__(self, buf, offset, shape, strides, dtype):
        self.buf = buf
        self.offset = int(offset)  # bytes
        self.shape = tuple(int(s) for s in shape)
        self.strides = tuple(int(s) for s in strides)  # elements
        self.dtype = np.dtype(dtype)
        n = 1
        for s in self.shape:
            n *= s
        self.size = n

    # -- construction -----------------------------------------------------------
    @classmethod
    def empty(cls, shape, dtype) -> "DeviceArray":
        dtype = np.dtype(dtype)
        shape = tuple(int(s) for s in shape)
        n = 1
        for s in shape:
            n *= s
        return cls(Buffer(n * dtype.itemsize), 0, shape, contiguous_strides(shape), dtype)

    @classmethod
    def from_host(cls, a: np.ndarray) -> "DeviceArray":
        a = np.asarray(a)
        if not a.flags.c_contiguous:
            a = np.ascontiguousarray(a)  # (never for 0-d: ascontiguousarray would make it 1-d)
        out = cls.empty(a.shape, a.dtype)
        if a.size:
            ffi.check(ffi.lib().pthip_h2d(out.ptr, a.ctypes.data, a.nbytes))
            # the copy from pageable memory is staged before return; `a` may die now
        return out

    # -- properties ---------------------------------------------------------------
    @property
    def ptr(self) -> int:
        return self.buf.ptr + self.offset

    @property
    def ndim(self) -> int:
        return len(self.shape)

    @property
    def itemsize(self) -> int:
        return self.dtype.itemsize

    @property
    def nbytes(self) -> int:
        return self.size * self.dtype.itemsize

    def is_contiguous(self) -> bool:
        if self.size <= 1:
            return True
        acc = 1
        for s, st in zip(reversed(self.shape), reversed(self.strides)):
            if s == 1:
                continue
            if st != acc:
                return False
            acc *= s
        return True

    def view(self, shape, strides, offset_elems=0) -> "DeviceArray":
        return DeviceArray(self.buf, self.offset + offset_elems * self.itemsize, shape, strides, self.dtype)

    # -- transfers ----------------------------------------------------------------
    def contiguous(self) -> "DeviceArray":
        if self.is_contiguous():
            return self
        out = DeviceArray.empty(self.shape, self.dtype)
        copy_into(out, self)
        return out

    def contiguous_copy(self) -> "DeviceArray":
        """A fresh contiguous copy (always a new buffer)."""
        out = DeviceArray.empty(self.shape, self.dtype)
        copy_into(out, self)
        return out

    def to_host(self, sync=True) -> np.ndarray:
        src = self.contiguous()
        out = np.empty(self.shape, dtype=self.dtype)
        if out.size:
            ffi.check(ffi.lib().pthip_d2h(out.ctypes.data, src.ptr, out.nbytes))
        if sync:
            ffi.check(ffi.lib().pthip_synchronize())
        return out

    def __repr__(self):
        return f"DeviceArray(shape={self.shape}, strides={self.strides}, dtype={self.dtype})"


_TILED_COPY_MIN = 1 << 14  # elements


def _i64arr(vals):
    return (C.c_int64 * max(len(vals), 1))(*[int(v) for v in vals])


def copy_into(dst: DeviceArray, src: DeviceArray):
    """dst[...] = broadcast(src) (same dtype); strided both sides."""
    assert dst.dtype == src.dtype, (dst.dtype, src.dtype)
    nd = dst.ndim
    sshape = (1,) * (nd - src.ndim) + src.shape
    sstr = (0,) * (nd - src.ndim) + src.strides
    sstr = tuple(0 if sshape[k] == 1 and dst.shape[k] != 1 else sstr[k] for k in range(nd))
    for k in range(nd):
        if sshape[k] != dst.shape[k] and sshape[k] != 1:
            raise ValueError(f"could not broadcast input array from shape {src.shape} into shape {dst.shape}")
    if dst.size == 0:
        return
    if dst.size >= _TILED_COPY_MIN and dst.is_contiguous() and not src.is_contiguous():
        # a large strided / transposed source into a contiguous destination: the tiled N-d Elemwise loop with the
        # identity as its scalar graph (pack loads along the source's contiguous axis, LDS transposes) instead
        # of the element-per-thread kernel with a 64-bit division per dimension per element
        from pytensor_amd.dispatch.elemwise import tiled_copy

        if tiled_copy(dst, DeviceArray(src.buf, src.offset, sshape, sstr, src.dtype)):
            return
    if nd > 6 and sum(1 for s_ in dst.shape if s_ != 1) > 6:
        # the strided-copy kernel walks at most 6 non-mergeable dims (a Tile over 7+ axes, reference
        # test tests/tensor/test_basic.py::TestTile): peel the leading axis on the host
        sv = DeviceArray(src.buf, src.offset, sshape, sstr, src.dtype)
        for i in range(dst.shape[0]):
            copy_into(dst.view(dst.shape[1:], dst.strides[1:], i * dst.strides[0]), sv.view(sshape[1:], sstr[1:], i * sstr[0]))
        return
    ffi.check(
        ffi.lib().pthip_copy_strided(
            dst.itemsize, nd, _i64arr(dst.shape), dst.ptr, _i64arr(dst.strides), src.ptr, _i64arr(sstr)
        )
    )
