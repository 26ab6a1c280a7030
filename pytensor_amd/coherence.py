"""Coherence between a resident (shared-variable) host array and its copy in HBM.

The reference backends read a shared variable's storage cell on every call, so an in-place edit
of a borrowed value (``w.get_value(borrow=True)[i] = v``; ``set_value(x, borrow=True)`` then
``x[idx] += d``; pytensor/compile/sharedvalue.py:97-130) is seen by the next call.  A
device-resident copy must notice such edits without re-reading gigabytes per call.  Modes
(``config.hip__resident`` / ``PTHIP_RESIDENT``):

``guard``  (default, sound)  arrays up to 64 KiB: a 64-bit hash of the whole content on every
           call.  Larger arrays: the whole pages inside the array are write-protected after the
           upload (``pthip_guard_protect``, csrc/guard.hip: the first CPU store faults, the handler
           marks the array dirty, unprotects and lets the store proceed) and the two partial
           pages at its ends are hashed on every call (<= 8 KiB).  A clean array costs one flag
           load.  Every CPU store is seen, whichever view or thread makes it.
``strict`` hash everything on every call (sound; ~0.1 ms per MB).
``sampled`` round-2 behaviour, opt-in: a fingerprint of 256 evenly spaced elements — a sparse
           edit between the sample points is NOT seen.
``trust``  identity only.
"""

from __future__ import annotations

import ctypes as C
import os

import numpy as np

from pytensor_amd import ffi

FULL_HASH_MAX = 64 << 10
_NSAMPLE = 256
_PAGE = os.sysconf("SC_PAGESIZE") if hasattr(os, "sysconf") else 4096
_sample_cache = {}

try:  # xxh3: ~10 GB/s; zlib.crc32 as the always-present fallback
    from xxhash import xxh3_64_intdigest as _hash64
except Exception:  # pragma: no cover
    import zlib

    def _hash64(buf):
        return zlib.crc32(buf)


_MODES = ("guard", "strict", "sampled", "trust")
_mode_override = None


def mode() -> str:
    if _mode_override is not None:
        return _mode_override
    # the config flag registered with mode="hip" (linker._add_config_flags; its default is the
    # PTHIP_RESIDENT environment variable), else — below the drop-in boundary, no PyTensor — the
    # variable itself
    import sys

    pt = sys.modules.get("pytensor")
    m = getattr(getattr(pt, "config", None), "hip__resident", None) if pt is not None else None
    if m is None:
        m = os.environ.get("PTHIP_RESIDENT")
    m = m or "guard"
    if m not in _MODES:
        raise ValueError(f"hip linker: resident coherence mode {m!r} is not one of {_MODES}")
    return m


def set_mode(m):
    """Process-wide override (tests); ``None`` returns to the flag / environment."""
    global _mode_override
    if m is not None and m not in _MODES:
        raise ValueError(m)
    _mode_override = m


def _span(a: np.ndarray):
    """[lo, hi) byte addresses touched by ``a`` (any strides)."""
    lo = hi = a.ctypes.data
    for n, s in zip(a.shape, a.strides):
        if s > 0:
            hi += (n - 1) * s
        else:
            lo += (n - 1) * s
    return lo, hi + a.itemsize


def _hash_range(lo, hi):
    return _hash64((C.c_char * (hi - lo)).from_address(lo)) if hi > lo else 0


def _hash_all(a):
    if a.size == 0:
        return 0
    c = a if a.flags.c_contiguous else np.ascontiguousarray(a)
    return _hash64(c.reshape(-1).view(np.uint8).data)


class _Hash:
    """content hash, recomputed per call (overlapped with the replay by the plan: ``late``)"""

    __slots__ = ("h",)
    late = True

    def __init__(self, a):
        self.h = _hash_all(a)

    def clean(self, a):
        return _hash_all(a) == self.h

    def release(self):
        pass


class _Sample:
    __slots__ = ("s",)
    late = True

    @staticmethod
    def _take(a):
        idx = _sample_cache.get(a.size)
        if idx is None:
            idx = _sample_cache[a.size] = np.linspace(0, a.size - 1, _NSAMPLE).astype(np.int64)
        return a.flat[idx].tobytes()

    def __init__(self, a):
        self.s = self._take(a)

    def clean(self, a):
        return self._take(a) == self.s

    def release(self):
        pass


class _Guard:
    """write-protected interior pages + hashed partial pages at the two ends"""

    __slots__ = ("slot", "flag", "_clean")
    late = False

    def __init__(self, lo, hi):
        plo = (lo + _PAGE - 1) & ~(_PAGE - 1)
        phi = hi & ~(_PAGE - 1)
        lib = ffi.lib()
        slot, flag = C.c_int(-1), C.c_void_p()
        ffi.check(lib.pthip_guard_protect(plo, phi - plo, C.byref(slot), C.byref(flag)))
        self.slot = slot.value
        self.flag = C.c_int.from_address(flag.value)
        # the partial pages at the two ends are not protected: the library keeps copies and compares them
        # (pthip_guard_clean: the dirty flag + two memcmp of < 1 page, one native call per evaluation)
        try:
            ffi.check(lib.pthip_guard_set_edges(self.slot, lo, plo - lo, phi, hi - phi))
        except Exception:
            self.release()
            raise
        self._clean = lib.pthip_guard_clean

    def clean(self, a=None):
        return self._clean(self.slot) == 1

    def release(self):
        if self.slot >= 0:
            try:
                ffi.lib().pthip_guard_release(self.slot)
            except Exception:  # pragma: no cover (interpreter shutdown)
                pass
            self.slot = -1

    def __del__(self):
        self.release()


class _ReadOnly:
    """An update-fed shared value that lives in one of the executor's PINNED result blocks (the array
    ``Function`` installs in the storage cell after a call with ``updates=``).  Pinned pages must not be
    write-protected: the device writes through its own mapping and a CPU-side ``mprotect`` makes the
    driver drop that mapping — a later device store then never completes (measured on the MI355X box,
    profiles/r3b_guard_probe.txt).  The array is handed out read-only instead: an in-place edit
    (``w.get_value(borrow=True)[i] = v``) raises NumPy's "assignment destination is read-only" — loud,
    where the reference would silently accept it — and re-enabling the flag marks the value dirty
    (conservative: the next call uploads it again).  ``get_value()`` (a copy) and ``set_value`` behave
    as in the reference."""

    __slots__ = ()
    late = False

    def clean(self, a):
        return not a.flags.writeable

    def release(self):
        pass


_pinned = []  # [lo, hi) address ranges of live pinned blocks (plan._PinnedBlock)


def register_pinned(ptr: int, nbytes: int):
    _pinned.append((int(ptr), int(ptr) + int(nbytes)))


def unregister_pinned(ptr: int):
    for k, (lo, _) in enumerate(_pinned):
        if lo == int(ptr):
            del _pinned[k]
            return


def is_pinned(lo: int, hi: int) -> bool:
    return any(lo < phi and plo < hi for plo, phi in _pinned)


def watch_update_fed(a: np.ndarray):
    """Token for the host mirror of an update-fed resident: pinned + large -> read-only hand-out."""
    if mode() != "trust" and a.nbytes > FULL_HASH_MAX and a.size and is_pinned(*_span(a)):
        try:
            a.flags.writeable = False
            return _ReadOnly()
        except ValueError:  # pragma: no cover
            pass
    return watch(a)


def watch(a: np.ndarray):
    """Start watching ``a`` (just uploaded).  Returns a token with ``clean(a)``, ``release()`` and
    ``late`` (True: the check reads the array and is worth overlapping with device work), or
    ``None`` when nothing is checked (mode ``trust``)."""
    m = mode()
    if m == "trust":
        return None
    if a.nbytes <= FULL_HASH_MAX or m == "strict":
        return _Hash(a)
    if m == "sampled":
        return _Sample(a)
    lo, hi = _span(a)
    if is_pinned(lo, hi):
        return _Hash(a)  # (never mprotect pinned pages: see _ReadOnly)
    if not a.flags.writeable or hi - lo < 4 * _PAGE:
        # (a read-only mapping must not be made writable by the handler; nothing else can store
        # through this array object, but another view might: hash it)
        return _Hash(a)
    try:
        return _Guard(lo, hi)
    except ffi.HipError:
        return _Hash(a)  # (out of slots, or the kernel refused the protection: still sound)


def clean(token, a) -> bool:
    return token is None or token.clean(a)


def release(token):
    if token is not None:
        token.release()
