"""GPU: the index / layout / signal / FFT ops of dispatch/misc.py and dispatch/fft.py, and lazy IfElse,
on shapes and corner cases the golden cases do not reach, against the oracle's restatement of the
reference's ``perform`` (NumPy / SciPy calls)."""
import numpy as np
import pytest

import np_graph
from pytensor_amd.ir import Graph

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from pytensor_amd import ffi

    if ffi.device_count() <= 0:
        pytest.fail("no HIP device visible: GPU tests must run on the MI355X box")
    ffi.init(0)
    return ffi


def one_node(op, params, in_specs, out_specs, consts=()):
    g = Graph(name=f"one_{op}")
    ins = [g.new_var(dt, (None,) * nd) for dt, nd in in_specs]
    cs = [g.new_var(str(np.asarray(c).dtype), np.asarray(c).shape, const=np.asarray(c)) for c in consts]
    outs = [g.new_var(dt, (None,) * nd) for dt, nd in out_specs]
    g.add_node(op, params, [*ins, *cs], outs)
    g.inputs, g.outputs = ins, outs
    return g


def check(g, *vals, rtol=0.0, atol=0.0):
    from pytensor_amd.executor import HipExecutable

    got = HipExecutable(g)(*vals)
    want = np_graph.run_graph(g, list(vals))
    assert len(got) == len(want)
    for k, (a, b) in enumerate(zip(got, want)):
        b = np.asarray(b)
        assert a.shape == b.shape and a.dtype == b.dtype, (k, a.shape, b.shape, a.dtype, b.dtype)
        if b.dtype.kind in "biu" or (rtol == 0.0 and atol == 0.0):
            assert np.array_equal(a, b, equal_nan=b.dtype.kind == "f"), k
        else:
            np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, err_msg=str(k))
    return got


def test_searchsorted_dtypes_nan_and_empty(hip):
    rng = np.random.default_rng(0)
    x = np.sort(rng.normal(size=1000))
    x[-3:] = np.nan  # NaNs sort last; a NaN key goes to the first NaN (left) or the end (right)
    v = np.concatenate([rng.normal(size=500), [np.nan, -np.inf, np.inf, x[10], x[10]]])
    for side in ("left", "right"):
        check(one_node("SearchsortedOp", {"side": side}, [("float64", 1), ("float64", 1)], [("int64", 1)]), x, v)
        check(one_node("SearchsortedOp", {"side": side}, [("float32", 1), ("float64", 2)], [("int64", 2)]), x.astype("float32")[:900], v[:504].reshape(4, 126))
        check(one_node("SearchsortedOp", {"side": side}, [("int64", 1), ("int8", 1)], [("int64", 1)]),
              np.sort(rng.integers(-50, 50, size=64)), rng.integers(-60, 60, size=33).astype("int8"))
        check(one_node("SearchsortedOp", {"side": side}, [("float64", 1), ("float64", 1)], [("int64", 1)]), np.zeros(0), v[:5])
    perm = rng.permutation(997)
    xs = rng.normal(size=997)
    check(one_node("SearchsortedOp", {"side": "left"}, [("float64", 1), ("float64", 1), ("int32", 1)], [("int64", 1)]),
          xs, rng.normal(size=40), np.argsort(xs).astype("int32"))


def test_unique_repeat_ravel(hip):
    rng = np.random.default_rng(1)
    with_nan = np.round(rng.normal(size=60), 1)
    with_nan[[3, 17, 40]] = np.nan  # np.unique(equal_nan=True): one NaN in the result, counted three times
    for x in (rng.integers(0, 50, size=5000), np.round(rng.normal(size=(30, 40)), 1), np.array([3.0]), np.zeros(0), rng.integers(0, 3, size=7).astype("int8"), with_nan):
        dt, nd = str(x.dtype), x.ndim
        check(one_node("Unique", {"return_index": True, "return_inverse": True, "return_counts": True, "axis": None},
                       [(dt, nd)], [(dt, 1), ("int64", 1), ("int64", 1), ("int64", 1)]), x)
        check(one_node("Unique", {"return_index": False, "return_inverse": False, "return_counts": False, "axis": None}, [(dt, nd)], [(dt, 1)]), x)
    # unique slices along an axis: lexicographic order, first occurrences, 1-d inverse
    rows = rng.integers(0, 3, size=(200, 3))
    cube = rng.integers(0, 2, size=(4, 30, 2)).astype("float64")
    for x, axis in ((rows, 0), (rows.T.copy(), 1), (cube, 1), (cube, 0), (np.round(rng.normal(size=(50, 2)), 0), 0), (np.zeros((0, 3)), 0), (np.ones((5, 1)), 0)):
        dt, nd = str(x.dtype), x.ndim
        check(one_node("Unique", {"return_index": True, "return_inverse": True, "return_counts": True, "axis": axis},
                       [(dt, nd)], [(dt, nd), ("int64", 1), ("int64", 1), ("int64", 1)]), x)
        check(one_node("Unique", {"return_index": False, "return_inverse": False, "return_counts": False, "axis": axis}, [(dt, nd)], [(dt, nd)]), x)
    m = rng.normal(size=(4, 5, 3))
    check(one_node("Repeat", {"axis": 1}, [("float64", 3), ("int64", 1)], [("float64", 3)]), m, np.array([0, 3, 1, 0, 2]))
    check(one_node("Repeat", {"axis": 2}, [("float64", 3), ("int64", 1)], [("float64", 3)]), m, np.array([2, 2, 2]))
    check(one_node("Repeat", {"axis": 0}, [("float64", 3), ("int64", 1)], [("float64", 3)]), m, np.zeros(4, dtype="int64"))
    with pytest.raises(ValueError, match="negative"):
        check(one_node("Repeat", {"axis": 0}, [("float64", 3), ("int64", 1)], [("float64", 3)]), m, np.array([1, -1, 1, 1]))
    dims = np.array([7, 1, 11, 5])
    flat = rng.integers(0, 385, size=(6, 50))
    for order in "CF":
        comps = check(one_node("UnravelIndex", {"order": order}, [("int64", 2)], [("int64", 2)] * 4, consts=[dims]), flat)
        g = one_node("RavelMultiIndex", {"mode": "raise", "order": order}, [("int64", 2)] * 4, [("int64", 2)], consts=[dims])
        (back,) = check(g, *comps)
        assert np.array_equal(back, flat)  # round trip
    with pytest.raises(ValueError, match="invalid entry"):
        check(one_node("RavelMultiIndex", {"mode": "raise", "order": "C"}, [("int64", 1)] * 2, [("int64", 1)], consts=[np.array([3, 4])]),
              np.array([0, 3]), np.array([1, 1]))


def test_fill_diagonal_bartlett_lu_conv(hip):
    rng = np.random.default_rng(2)
    for shape in ((5, 5), (3, 8), (8, 3), (1, 1)):
        a = rng.normal(size=shape)
        check(one_node("FillDiagonal", {}, [("float64", 2), ("float64", 0)], [("float64", 2)]), a, np.asarray(7.5))
        for off in (-9, -2, 0, 1, 4, 9):
            check(one_node("FillDiagonalOffset", {}, [("float64", 2), ("float64", 0)], [("float64", 2)], consts=[np.asarray(off)]), a, np.asarray(-1.0))
    check(one_node("FillDiagonal", {}, [("float32", 4), ("float64", 0)], [("float32", 4)]), rng.normal(size=(3, 3, 3, 3)).astype("float32"), np.asarray(2.0))
    for M in (0, 1, 2, 7, 1000):
        check(one_node("Bartlett", {}, [], [("float64", 1)], consts=[np.asarray(M)]))
    for n in (2, 17, 64, 130):  # (n = 1: scipy.linalg.lu itself returns int64 indices there, against LU.make_node's int32)
        A = rng.normal(size=(n, n))
        for params, specs in (({"permute_l": False, "p_indices": False}, [("float64", 2)] * 3), ({"permute_l": True, "p_indices": False}, [("float64", 2)] * 2),
                              ({"permute_l": False, "p_indices": True}, [("int32", 1), ("float64", 2), ("float64", 2)])):
            check(one_node("LU", params, [("float64", 2)], specs), A, rtol=1e-10, atol=1e-12)
    for na, nb in ((1, 1), (5, 1), (1, 5), (100, 7), (7, 100), (4096, 33)):
        a, b = rng.normal(size=na), rng.normal(size=nb)
        for full in (True, False):
            check(one_node("Convolve1d", {}, [("float64", 1), ("float64", 1)], [("float64", 1)], consts=[np.asarray(full)]), a, b, rtol=1e-12, atol=1e-13)
    ai, bi = rng.integers(-9, 9, size=40), rng.integers(-9, 9, size=6).astype("int16")
    check(one_node("Convolve1d", {}, [("int64", 1), ("int16", 1)], [("int64", 1)], consts=[np.asarray(True)]), ai, bi)
    for sa, sb in (((1, 1), (1, 1)), ((20, 30), (3, 5)), ((3, 5), (20, 30)), ((7, 7), (7, 7))):
        a, b = rng.normal(size=sa), rng.normal(size=sb)
        for full in (True, False):
            check(one_node("Convolve2d", {}, [("float64", 2), ("float64", 2)], [("float64", 2)], consts=[np.asarray(full)]), a, b, rtol=1e-12, atol=1e-13)


def test_choose_and_permute(hip):
    rng = np.random.default_rng(3)
    ch = rng.normal(size=(5, 4, 7))
    a = rng.integers(-9, 14, size=(4, 7))
    for mode in ("wrap", "clip"):
        check(one_node("Choose", {"mode": mode}, [("int64", 2), ("float64", 3)], [("float64", 2)]), a, ch)
    check(one_node("Choose", {"mode": "raise"}, [("int32", 1), ("float64", 3)], [("float64", 2)]), rng.integers(0, 5, size=7).astype("int32"), ch)
    with pytest.raises(ValueError, match="invalid entry in choice array"):
        check(one_node("Choose", {"mode": "raise"}, [("int64", 2), ("float64", 3)], [("float64", 2)]), a, ch)
    x = rng.normal(size=(3, 4, 9))
    p = np.stack([np.stack([rng.permutation(9) for _ in range(4)]) for _ in range(3)])
    for inverse in (False, True):
        check(one_node("PermuteRowElements", {"inverse": inverse}, [("float64", 3), ("int64", 3)], [("float64", 3)]), x, p)
        check(one_node("PermuteRowElements", {"inverse": inverse}, [("float64", 3), ("int64", 2)], [("float64", 3)]), x, p[0])
        check(one_node("PermuteRowElements", {"inverse": inverse}, [("float64", 1), ("int64", 3)], [("float64", 3)]), x[0, 0], p)


@pytest.mark.parametrize("n", [1, 2, 3, 16, 17, 250, 1024])
def test_rfft_irfft_lengths(hip, n):
    rng = np.random.default_rng(n)
    x = rng.normal(size=(5, n))
    tol = 1e-13 * max(n, 8) * np.abs(x).max() * np.sqrt(n)
    (X,) = check(one_node("RFFTOp", {}, [("float64", 2)], [("float64", 3)], consts=[np.array([n])]), x, rtol=0, atol=tol)
    (back,) = check(one_node("IRFFTOp", {}, [("float64", 3)], [("float64", 2)], consts=[np.array([n])]), X, rtol=0, atol=tol * n)
    np.testing.assert_allclose(back / n, x, atol=tol)  # the round trip
    if n >= 16:
        # s pads and truncates; a transform over two axes; float32
        check(one_node("RFFTOp", {}, [("float64", 2)], [("float64", 3)], consts=[np.array([n + 5])]), x, rtol=0, atol=tol)
        check(one_node("RFFTOp", {}, [("float64", 2)], [("float64", 3)], consts=[np.array([n - 3])]), x, rtol=0, atol=tol)
        x3 = rng.normal(size=(2, 6, n))
        (X3,) = check(one_node("RFFTOp", {}, [("float64", 3)], [("float64", 4)], consts=[np.array([6, n])]), x3, rtol=0, atol=10 * tol)
        check(one_node("IRFFTOp", {}, [("float64", 4)], [("float64", 3)], consts=[np.array([6, n])]), X3, rtol=0, atol=100 * tol * n)
        xf = x.astype("float32")
        check(one_node("RFFTOp", {}, [("float32", 2)], [("float32", 3)], consts=[np.array([n])]), xf, rtol=0, atol=2e-6 * n * np.abs(x).max())


def test_ifelse_runs_only_the_branch_taken(hip):
    """a Solve with mismatched shapes sits in the branch that is not taken: it must not run"""
    from pytensor_amd.executor import HipExecutable

    g = Graph(name="lazy")
    c = g.new_var("bool", (), name="c")
    A, b, b_bad = g.new_var("float64", (None, None)), g.new_var("float64", (None,)), g.new_var("float64", (None,))
    good, bad, out = (g.new_var("float64", (None,)) for _ in range(3))
    sp = {"assume_a": "gen", "lower": False, "b_ndim": 1}
    g.add_node("Solve", sp, [A, b], [good])
    g.add_node("Solve", sp, [A, b_bad], [bad])
    g.add_node("IfElse", {"n_outs": 1}, [c, good, bad], [out])
    g.inputs, g.outputs = [c, A, b, b_bad], [out]
    rng = np.random.default_rng(4)
    Av, bv = rng.normal(size=(6, 6)), rng.normal(size=6)
    exe = HipExecutable(g, auto_freeze=True)
    for _ in range(3):
        (x,) = exe(np.asarray(True), Av, bv, np.zeros(4))
        np.testing.assert_allclose(Av @ x, bv, atol=1e-12)
    with pytest.raises(ValueError, match="incompatible shapes"):
        exe(np.asarray(False), Av, bv, np.zeros(4))
    assert exe.stats["captures"] == 0  # the condition is read on the host: such graphs stay eager


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_scatter_add_many_bins_is_exact_and_reproducible(hip, dtype):
    """``out[idx] += y`` with more bins than the fixed-order kernel takes (> 256; AdvancedIncSubtensor1,
    pytensor/tensor/subtensor.py: np.add.at semantics) accumulates 128-bit fixed-point integers with integer atomics
    (csrc/index.hip scatter_add_exact_kernel): every bin is the EXACT sum of its addends rounded once — compared
    here with math.fsum — so two calls give the same bits, whatever order the hardware served the adds in.  Duplicated
    indices by the thousand, negative indices, mixed signs and magnitudes over 2^40, zeros, -0.0, subnormals, and bins
    that receive NaN / +inf / -inf / both infinities."""
    import ctypes as C
    import math

    from pytensor_amd.device import DeviceArray

    lib = hip.lib()
    rng = np.random.default_rng(31)
    n, bins = 200_000, 5000
    idx = rng.integers(0, bins - 8, size=n)
    idx[::7] -= bins  # negative indices wrap
    y = rng.normal(size=n) * np.exp2(rng.integers(-20, 20, size=n))
    y[::11] = 0.0
    y[5::13] = -0.0
    if dtype == "float64":
        y[3::1001] = 5e-324 * rng.integers(1, 1000, size=y[3::1001].shape)  # subnormals (far below the window: they may be dropped)
    y = y.astype(dtype)
    # the special bins bins-8 .. bins-1
    sp_idx = np.array([bins - 8, bins - 7, bins - 6, bins - 6, bins - 5, bins - 5, bins - 4], dtype="int64")
    sp_y = np.array([np.nan, np.inf, -np.inf, 1.0, np.inf, -np.inf, 3.5], dtype=dtype)
    idx = np.concatenate([idx, sp_idx])
    y = np.concatenate([y, sp_y])
    base = rng.normal(size=bins).astype(dtype)
    d_idx, d_y = DeviceArray.from_host(idx), DeviceArray.from_host(y)

    def run():
        out = DeviceArray.from_host(base.copy())
        ws_bytes = lib.pthip_scatter_rows_workspace(len(idx), bins, 1)
        ws = DeviceArray.empty((ws_bytes,), "uint8")
        hip.check(lib.pthip_scatter_rows(hip.np_dtype_code(dtype), 1, len(idx), 1, out.ptr, bins, d_idx.ptr, d_y.ptr, 1, ws.ptr, ws_bytes))
        return out.to_host()

    a, b = run(), run()
    np.testing.assert_array_equal(a, b)
    st = C.c_int(-1)
    hip.check(lib.pthip_check_status(C.byref(st)))
    assert st.value == 0
    # exact sums: the window keeps 43 bits below the largest addend's last bit, so everything here but the subnormals
    # is inside it; the subnormals are 1e-300 of a rounding step
    want = np.empty(bins, dtype="float64")
    pos = np.where(idx < 0, idx + bins, idx)
    order = np.argsort(pos, kind="stable")
    bounds = np.searchsorted(pos[order], np.arange(bins + 1))
    y64 = y.astype("float64")
    for k in range(bins):
        vals = y64[order[bounds[k]:bounds[k + 1]]]
        vals = vals[np.abs(vals) > 1e-300] if np.isfinite(vals).all() else vals
        want[k] = math.fsum(vals) if np.isfinite(vals).all() else np.sum(vals)
    with np.errstate(invalid="ignore"):
        want = (base.astype("float64") + want).astype(dtype)
    np.testing.assert_array_equal(a[: bins - 8], want[: bins - 8])
    assert np.isnan(a[bins - 8]) and a[bins - 7] == np.inf and a[bins - 6] == -np.inf and np.isnan(a[bins - 5])
    assert a[bins - 4] == want[bins - 4]

    # rows of a matrix (inner = 3: bins = rows x 3), few addends per bin (the table in memory, not in LDS) and many
    for rows, n2 in ((40_000, 30_000), (700, 60_000)):
        idx2 = rng.integers(-rows, rows, size=n2)
        y2 = (rng.normal(size=(n2, 3)) * np.exp2(rng.integers(-10, 10, size=(n2, 1)))).astype(dtype)
        base2 = rng.normal(size=(rows, 3)).astype(dtype)
        d_i2, d_y2 = DeviceArray.from_host(idx2), DeviceArray.from_host(y2)
        got = []
        for _ in range(2):
            out = DeviceArray.from_host(base2.copy())
            ws_bytes = lib.pthip_scatter_rows_workspace(n2, rows, 3)
            ws = DeviceArray.empty((ws_bytes,), "uint8")
            hip.check(lib.pthip_scatter_rows(hip.np_dtype_code(dtype), 1, n2, 3, out.ptr, rows, d_i2.ptr, d_y2.ptr, 3, ws.ptr, ws_bytes))
            got.append(out.to_host())
        np.testing.assert_array_equal(got[0], got[1])
        pos2 = np.where(idx2 < 0, idx2 + rows, idx2)
        exact = np.zeros((rows, 3))
        order2 = np.argsort(pos2, kind="stable")
        b2 = np.searchsorted(pos2[order2], np.arange(rows + 1))
        y2d = y2.astype("float64")
        for k in range(rows):
            sel = order2[b2[k]:b2[k + 1]]
            if len(sel):
                exact[k] = [math.fsum(y2d[sel, c]) for c in range(3)]
        np.testing.assert_array_equal(got[0], (base2.astype("float64") + exact).astype(dtype))
