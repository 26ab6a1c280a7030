"""Host side of resident coherence (pytensor_amd/coherence.py + csrc/guard.hip): no GPU needed —
page protection and the SIGSEGV handler are plain host code inside libpthip.so."""
import ctypes as C
import subprocess
import sys
import threading

import numpy as np
import pytest

from pytensor_amd import coherence, ffi


@pytest.fixture(autouse=True)
def _default_mode():
    coherence.set_mode("guard")
    yield
    coherence.set_mode(None)


def _slots():
    u, a = C.c_int(), C.c_int()
    ffi.check(ffi.lib().pthip_guard_stats(C.byref(u), C.byref(a)))
    return u.value, a.value


def test_small_arrays_are_hashed_whole():
    a = np.arange(1000.0)
    t = coherence.watch(a)
    assert type(t).__name__ == "_Hash" and t.late
    assert t.clean(a)
    a[517] += 1
    assert not t.clean(a)


@pytest.mark.parametrize("where", ["first", "interior", "last", "old_view", "thread", "strided_view"])
def test_guard_sees_every_single_element_store(where):
    a = np.random.default_rng(0).normal(size=1 << 18)  # 2 MB
    old = a[1000:2000]
    t = coherence.watch(a)
    assert type(t).__name__ == "_Guard" and not t.late
    assert a.sum() == a.sum() and t.clean(a), "reads must neither fault nor dirty"
    used, active = _slots()
    assert (used, active) == (1, 1)
    if where == "first":
        a[0] = 1.0
    elif where == "interior":
        a[131_071] = 1.0
    elif where == "last":
        a[-1] = 1.0
    elif where == "old_view":
        old[5] = 1.0
    elif where == "thread":
        th = threading.Thread(target=lambda: a.__setitem__(200_000, 1.0))
        th.start()
        th.join()
    else:
        a[::4097][3] = 1.0
    assert not t.clean(a)
    t.release()
    assert _slots() == (0, 0)
    a[5] = 2.0  # writable again, no handler involved


def test_rewatch_after_dirty_and_overlapping_watches():
    a = np.zeros(1 << 17)
    t1, t2 = coherence.watch(a), coherence.watch(a[10:])
    a[70_000] = 1.0
    assert not t1.clean(a) and not t2.clean(a)
    t1.release()
    t2.release()
    t3 = coherence.watch(a)
    assert t3.clean(a)
    a[70_001] = 1.0
    assert not t3.clean(a)
    t3.release()
    assert _slots() == (0, 0)


def test_non_contiguous_span_and_readonly_fall_back_soundly():
    base = np.zeros((600, 600))
    v = base[:, 5:300]  # strided: the guard covers the span, stores next to it are conservative
    t = coherence.watch(v)
    assert t.clean(v)
    base[300, 7] = 1.0
    assert not t.clean(v)
    t.release()
    r = np.zeros(1 << 16)
    r.flags.writeable = False
    assert type(coherence.watch(r)).__name__ == "_Hash"


def test_modes():
    a = np.zeros(1 << 16)
    for m, name in [("strict", "_Hash"), ("sampled", "_Sample"), ("trust", "NoneType")]:
        coherence.set_mode(m)
        assert type(coherence.watch(a)).__name__ == name
    with pytest.raises(ValueError):
        coherence.set_mode("bogus")


def test_token_released_with_its_owner():
    a = np.zeros(1 << 16)
    t = coherence.watch(a)
    assert _slots()[0] == 1
    del t
    assert _slots() == (0, 0)


def test_a_genuine_segfault_still_kills_the_process():
    code = ("import numpy as np, ctypes; from pytensor_amd import coherence; a = np.zeros(1 << 16); "
            "t = coherence.watch(a); ctypes.string_at(8)")
    r = subprocess.run([sys.executable, "-X", "faulthandler", "-c", code], capture_output=True, cwd=__import__("os").path.dirname(__import__("os").path.dirname(__file__)))
    assert r.returncode == -11, (r.returncode, r.stderr[-300:])
    assert b"Segmentation fault" in r.stderr  # faulthandler (installed before us) still gets its turn


def test_faulthandler_enabled_after_us_does_not_break_tracking():
    import faulthandler

    a = np.zeros(1 << 16)
    was = faulthandler.is_enabled()
    faulthandler.disable()
    t = coherence.watch(a)
    faulthandler.enable()  # replaces our handler ...
    t.release()
    t = coherence.watch(a)  # ... and the next protect goes back in front of it
    a[30_000] = 1.0
    assert not t.clean(a)
    t.release()
    if not was:
        faulthandler.disable()


def test_pinned_ranges_are_never_guarded_and_fed_values_are_handed_out_read_only():
    """Pinned result blocks must not be mprotect-ed (a device store into them would never complete):
    an array inside a registered pinned range is hashed; as the host mirror of an update-fed
    resident it is handed out read-only, and re-enabling writes marks it dirty."""
    a = np.zeros(1 << 16)
    lo = a.ctypes.data
    coherence.register_pinned(lo, a.nbytes)
    try:
        assert type(coherence.watch(a)).__name__ == "_Hash"
        assert _slots() == (0, 0)
        t = coherence.watch_update_fed(a)
        assert type(t).__name__ == "_ReadOnly" and not a.flags.writeable and t.clean(a)
        with pytest.raises(ValueError, match="read-only"):
            a[5] = 1.0
        a.flags.writeable = True  # the user insists: conservative dirty
        assert not t.clean(a)
    finally:
        coherence.unregister_pinned(lo)
    assert type(coherence.watch(a)).__name__ == "_Guard"
    small = np.zeros(100)
    assert type(coherence.watch_update_fed(small)).__name__ == "_Hash"


def test_concurrent_writers_from_many_threads_all_land_and_dirty_once():
    """several threads store into DIFFERENT pages of one guarded array at the same time: the handler runs on whichever
    thread faults first (the others fault on pages already unprotected by then, or re-enter the handler and find the slot
    dirty) — every store must land and the array must read dirty afterwards"""
    a = np.zeros(1 << 19)  # 4 MB = 1024 pages
    t = coherence.watch(a)
    assert type(t).__name__ == "_Guard"
    start = threading.Barrier(8)

    def writer(k):
        start.wait()
        for j in range(64):
            a[(k * 64 + j) * 512 + 7] = k + 1.0  # one element in each of 64 pages of this thread's own range

    ths = [threading.Thread(target=writer, args=(k,)) for k in range(8)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not t.clean(a)
    for k in range(8):
        assert all(a[(k * 64 + j) * 512 + 7] == k + 1.0 for j in range(64))
    assert float(a.sum()) == sum((k + 1.0) * 64 for k in range(8))
    t.release()
    assert _slots() == (0, 0)


@pytest.mark.skipif(not hasattr(__import__("os"), "fork"), reason="needs fork")
def test_a_forked_child_can_write_a_guarded_array_and_the_parent_stays_clean():
    """``os.fork`` (multiprocessing's default start method on Linux: PyMC's chains): the child inherits the protection and
    the handler; its stores into the (copy-on-write) pages must not kill it, and the PARENT's array and its clean state
    are untouched — the child's pages are its own"""
    import os

    a = np.arange(1 << 18, dtype="float64")
    t = coherence.watch(a)
    assert type(t).__name__ == "_Guard" and t.clean(a)
    r, w = os.pipe()
    pid = os.fork()
    if pid == 0:  # child: write, report what it read back, leave without running pytest's teardown
        code = 1
        try:
            a[123_456] = -5.0
            ok = a[123_456] == -5.0 and a[123_455] == 123_455.0 and not t.clean(a)
            os.write(w, b"ok" if ok else b"no")
            code = 0
        finally:
            os._exit(code)
    os.close(w)
    _, status = os.waitpid(pid, 0)
    msg = os.read(r, 2)
    os.close(r)
    assert os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0, f"the child died: status {status}"
    assert msg == b"ok"
    assert a[123_456] == 123_456.0 and t.clean(a), "the parent's copy is untouched and still clean"
    a[0] = 1.0  # and the parent's own guard still works after the fork
    assert not t.clean(a)
    t.release()
