"""CPU: the native call path of a frozen plan (pytensor_amd/_fastplan.so, csrc/fastplan.c) against stub
``replay`` / ``guard_clean`` entry points: what it accepts, what it hands back to the Python path (``None``
before anything is launched), fresh result arrays, ScalarType outputs, the device status word."""
import ctypes as C

import numpy as np
import pytest

fp = pytest.importorskip("pytensor_amd._fastplan")

REPLAY = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int)
GCLEAN = C.CFUNCTYPE(C.c_int, C.c_int)


class Rig:
    def __init__(self, status=0, rc=0):
        self.inblk = np.zeros(64, dtype=np.uint8)  # the "pinned staging block": a (3,) f64 at 0 and a 0-d f64 at 32
        self.outblk = np.zeros(8, dtype=np.float64)  # results: (2, 2) f64 at 0, a 0-d f64 at 32; status word at 56
        self.status = self.outblk.view(np.int32)[14:15]
        self.done = np.zeros(1, dtype=np.int32)
        self.clean = {7: 1}
        self.replays = 0
        self.res_obj = np.arange(10.0)
        self.gen = C.c_uint64(5)  # the executable's resident generation counter

        def replay(desc, host_out, done, sync):
            self.replays += 1
            x = self.inblk[:24].view(np.float64)
            s = self.inblk[32:40].view(np.float64)[0]
            self.outblk[:4] = [x.sum(), x[0], x[1], x[2] * s]
            self.outblk[4] = s * 2
            self.status[0] = status
            C.cast(done, C.POINTER(C.c_int))[0] = 1
            assert sync == 2
            return rc

        self._replay = REPLAY(replay)
        self._gclean = GCLEAN(lambda slot: self.clean.get(slot, 0))
        addr = lambda f: C.cast(f, C.c_void_p).value
        staged = [(0, self.inblk.ctypes.data, np.dtype("float64"), (3,)), (2, self.inblk.ctypes.data + 32, np.dtype("float64"), ())]
        res = [(1, self.res_obj, 7)]
        outs = [(self.outblk.ctypes.data, None, np.dtype("float64"), (2, 2), 0),
                (self.outblk.ctypes.data + 32, None, np.dtype("float64"), (), 1),
                (None, np.array([5, 6], dtype=np.int64), np.dtype("int64"), (2,), 0)]
        self.plan = fp.FastPlan(3, staged, res, outs, addr(self._replay), addr(self._gclean), 0, self.done.ctypes.data, self.status.ctypes.data, 2, C.addressof(self.gen), self.gen.value)


def test_fast_call_returns_fresh_arrays_and_scalars():
    r = Rig()
    x, s = np.array([1.0, 2.0, 3.0]), np.asarray(4.0)
    a, b, c = r.plan((x, r.res_obj, s))
    np.testing.assert_array_equal(a, [[6.0, 1.0], [2.0, 12.0]])
    assert isinstance(b, np.float64) and b == 8.0  # a ScalarType output: a NumPy scalar, not a 0-d array
    np.testing.assert_array_equal(c, [5, 6])
    a2, _, c2 = r.plan((x * 2, r.res_obj, s))
    assert a2 is not a and c2 is not c and a[0, 0] == 6.0 and a2[0, 0] == 12.0  # the first result is not overwritten
    c2[0] = 99
    assert r.plan((x, r.res_obj, s))[2][0] == 5  # constants are copied out
    assert r.plan.stats() == {"calls": 3, "misses": 0} and r.replays == 3


@pytest.mark.parametrize("what", ["shape", "dtype", "strided", "list", "other resident", "dirty resident", "arity", "not a tuple"])
def test_everything_else_goes_back_to_python_before_anything_is_launched(what):
    r = Rig()
    x, s, w = np.array([1.0, 2.0, 3.0]), np.asarray(4.0), r.res_obj
    args = {
        "shape": (np.zeros(4), w, s),
        "dtype": (x.astype("float32"), w, s),
        "strided": (np.zeros(6)[::2], w, s),
        "list": ([1.0, 2.0, 3.0], w, s),
        "other resident": (x, w.copy(), s),
        "dirty resident": (x, w, s),
        "arity": (x, w),
        "not a tuple": [x, w, s],
    }[what]
    if what == "dirty resident":
        r.clean[7] = 0
    before = r.inblk.copy()
    assert r.plan(args) is None
    assert r.replays == 0 and (r.inblk == before).all()  # nothing staged, nothing launched
    assert r.plan.stats()["misses"] == 1


def test_status_word_and_hip_error_come_back_as_ints():
    r = Rig(status=2)
    assert r.plan((np.zeros(3), r.res_obj, np.asarray(1.0))) == 2
    r = Rig(rc=1)
    assert r.plan((np.zeros(3), r.res_obj, np.asarray(1.0))) == -1


def test_the_resident_object_is_kept_alive():
    import gc
    import weakref

    r = Rig()
    ref = weakref.ref(r.res_obj)
    plan = r.plan
    r.res_obj = None
    del r.plan
    gc.collect()
    assert ref() is not None  # the plan compares against this address: it must not be recycled
    del plan
    gc.collect()
    assert ref() is None


def test_resident_generation_bump_hands_the_call_back():
    """``invalidate_resident`` / a re-upload by another plan or an eager call bumps the executable's
    generation counter; the native path must then decline (the Python path re-validates and re-uploads) even though
    the object is the captured one and its guard slot reads clean."""
    r = Rig()
    x, s = np.array([1.0, 2.0, 3.0]), np.asarray(4.0)
    assert r.plan((x, r.res_obj, s)) is not None and r.replays == 1
    r.gen.value += 1
    assert r.plan((x, r.res_obj, s)) is None and r.replays == 1  # declined BEFORE anything was launched
    assert r.plan.stats()["misses"] == 1
    r.gen.value -= 1
    assert r.plan((x, r.res_obj, s)) is not None and r.replays == 2
