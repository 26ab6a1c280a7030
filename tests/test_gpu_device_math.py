"""GPU: the generated kernels' own fp64 log1p and log (codegen.PRELUDE pt_log1p / pt_log: the classical 2^k (1 + f) reduction in
~60 / ~45 VALU instructions instead of the device library's ~125 / ~90) against long-double log1pl / logl, in ulps, over every magnitude and both
signs, the k = 0 / k = 1 switch points, and the special values; softplus and the shared sigmoid / softplus pair that call it.
The reference's Log1p / Softplus c_code is libm (scalar/basic.py:3042, scalar/math.py:1224): < 1 ulp — the bar here is 2.5 ulp
for log1p (2.1 measured on the host emulation, tools note in the prelude) and 4 ulp for softplus (exp's ulp on top)."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pte():
    import e2e_util

    pytensor = e2e_util.activate()
    if not e2e_util.have_gpu():
        pytest.fail("no HIP device visible: GPU tests must run on the MI355X box")
    return pytensor


def _ulps(got, want_ld):
    want = want_ld.astype(np.float64)
    ulp = np.abs(np.nextafter(want, np.inf) - want)
    return np.abs(got.astype(np.longdouble) - want_ld).astype(np.float64) / np.maximum(ulp, 5e-324)


def _points(seed=3):
    rng = np.random.default_rng(seed)
    n = 300_000
    r = rng.random
    parts = [r(n), -r(n) * 0.999999, np.exp((r(n) - 0.5) * 80.0), -np.exp(-r(n) * 40.0), np.ldexp(r(n) + 0.5, rng.integers(-1000, 1000, n)),
             (r(n) - 0.5) * 1.2, 0.41421356 + (r(n) - 0.5) * 1e-6, -0.29289321 + (r(n) - 0.5) * 1e-6, np.exp(-r(n) * 700.0)]
    return np.concatenate(parts)


def test_log1p_within_2p5_ulp_of_long_double(pte):
    import pytensor.tensor as pt

    x = pt.dvector("x")
    f = pte.function([x], pt.log1p(x), mode="hip")
    xs = _points()
    got = f(xs)
    worst = float(_ulps(got, np.log1p(xs.astype(np.longdouble))).max())
    assert worst <= 2.5, worst
    sp = np.array([0.0, -0.0, 1e-320, -1e-320, 4.9e-324, 1e-300, np.inf, -1.0, -1.5, -np.inf, np.nan, 1.79e308, -0.9999999999999999, 2.0**-54, -(2.0**-54)])
    got = f(sp)
    with np.errstate(all="ignore"):
        want = np.log1p(sp)
    np.testing.assert_array_equal(np.isnan(got), np.isnan(want))
    np.testing.assert_array_equal(np.signbit(got[~np.isnan(want)]), np.signbit(want[~np.isnan(want)]))
    fin = np.isfinite(want)
    np.testing.assert_array_equal(got[~fin & ~np.isnan(want)], want[~fin & ~np.isnan(want)])
    assert float(_ulps(got[fin], np.log1p(sp[fin].astype(np.longdouble))).max()) <= 2.5


def test_log_within_1p5_ulp_of_long_double(pte):
    import pytensor.tensor as pt

    x = pt.dvector("x")
    f = pte.function([x], pt.log(x), mode="hip")
    rng = np.random.default_rng(11)
    n = 300_000
    r = rng.random
    xs = np.concatenate([r(n) * 4.0, np.exp((r(n) - 0.5) * 1400.0), 1.0 + (r(n) - 0.5) * 0.6, np.ldexp(r(n) + 0.5, rng.integers(-1074, 1024, n)),
                         1.41421356 + (r(n) - 0.5) * 1e-6, 0.70710678 + (r(n) - 0.5) * 1e-6])
    xs = xs[(xs > 0) & np.isfinite(xs)]
    assert float(_ulps(f(xs), np.log(xs.astype(np.longdouble))).max()) <= 1.5
    sp = np.array([0.0, -0.0, -1.0, np.inf, -np.inf, np.nan, 4.9e-324, 1e-310, 1.0, 1.79e308])
    got = f(sp)
    with np.errstate(all="ignore"):
        want = np.log(sp)
    np.testing.assert_array_equal(np.isnan(got), np.isnan(want))
    np.testing.assert_array_equal(got[np.isinf(want)], want[np.isinf(want)])
    fin = np.isfinite(want)
    assert float(_ulps(got[fin], np.log(sp[fin].astype(np.longdouble))).max()) <= 1.5
    assert got[8] == 0.0 and not np.signbit(got[8])


def test_softplus_and_the_shared_pair(pte):
    import pytensor.tensor as pt

    x = pt.dvector("x")
    f1 = pte.function([x], pt.softplus(x), mode="hip")
    f2 = pte.function([x], [pt.sigmoid(x), pt.softplus(x)], mode="hip")  # one Composite holding both: pt_sig_sp
    rng = np.random.default_rng(5)
    xs = np.concatenate([rng.standard_normal(400_000) * 8.0, rng.uniform(-745.0, 745.0, 200_000), rng.standard_normal(100_000) * 1e-3,
                         np.array([0.0, -0.0, 18.0, -37.0, 33.3, 40.0, -800.0, 800.0, np.inf, -np.inf])])
    xl = xs.astype(np.longdouble)
    with np.errstate(all="ignore"):
        want_sp = np.where(xl > 0, xl, 0) + np.log1p(np.exp(-np.abs(xl)))
        want_sg = np.where(xl >= 0, 1 / (1 + np.exp(-np.abs(xl))), np.exp(-np.abs(xl)) / (1 + np.exp(-np.abs(xl))))
    fin = np.isfinite(xs)
    for got in (f1(xs), f2(xs)[1]):
        assert float(_ulps(got[fin], want_sp[fin]).max()) <= 4.0
        assert got[-2] == np.inf and got[-1] == 0.0
    sg = f2(xs)[0]
    # a value below the normal range has fewer bits: ulps of the smallest normal there
    assert float((np.abs(sg.astype(np.longdouble) - want_sg).astype(np.float64)[fin] / np.maximum(np.abs(np.nextafter(want_sg.astype(np.float64), np.inf) - want_sg.astype(np.float64)), 2.0**-1074)[fin]).max()) <= 4.0
    nan = f2(np.array([np.nan]))
    assert np.isnan(nan[0][0]) and np.isnan(nan[1][0])


def test_pow_is_exact_where_the_result_is_representable(pte):
    """the device library's pow is ~1.3 ulp everywhere, exact nowhere in particular: pow(3, 1) = 2.9999999999999996 and an
    integer power, computed like Pow.c_code as (T)pow((double)x, (double)y), truncated 19**3 to 6858.  codegen's pt_pow."""
    import pytensor.tensor as pt

    x, y = pt.dvector("x"), pt.dvector("y")
    f = pte.function([x, y], pt.pow(x, y), mode="hip")
    xs, ys = (a.ravel() for a in np.meshgrid(np.arange(-20, 21, dtype=np.float64), np.arange(0, 16, dtype=np.float64)))
    want = np.array([float(int(a) ** int(b)) for a, b in zip(xs, ys)])
    got = f(xs, ys)
    small = np.abs(want) < 2.0**53
    np.testing.assert_array_equal(got[small], want[small])
    assert float(np.max(np.abs(got - want) / np.maximum(np.abs(np.nextafter(want, np.inf) - want), 5e-324))) <= 2.0
    neg = f(np.array([2.0, 4.0, -2.0, 10.0, 3.0]), np.array([-3.0, -1.0, -3.0, -2.0, -1.0]))
    np.testing.assert_array_equal(neg, np.array([0.125, 0.25, -0.125, 1.0 / 100.0, 1.0 / 3.0]))
    xi, yi = pt.lvector("xi"), pt.lvector("yi")
    gi = pte.function([xi, yi], pt.pow(xi, yi), mode="hip")(xs.astype(np.int64), ys.astype(np.int64))
    wi = [int(a) ** int(b) for a, b in zip(xs, ys)]
    assert all(int(g) == w for g, w in zip(gi, wi) if abs(w) < 2**53)
    a32, b32 = pt.fvector("a"), pt.fvector("b")
    g32 = pte.function([a32, b32], pt.pow(a32, b32), mode="hip")(xs.astype(np.float32), ys.astype(np.float32))
    np.testing.assert_array_equal(g32, want.astype(np.float32))
    rng = np.random.default_rng(0)
    xr, yr = rng.uniform(0.1, 10, 100_000), rng.uniform(-5, 5, 100_000)
    assert float(_ulps(f(xr, yr), np.power(xr.astype(np.longdouble), yr.astype(np.longdouble))).max()) <= 2.0
    np.testing.assert_array_equal(f(xr, np.ones_like(xr)), xr)
    np.testing.assert_array_equal(f(xr, np.full_like(xr, 2.0)), xr * xr)
    sp = f(np.array([0.0, -0.0, np.nan, np.inf, -np.inf, 5.0, np.nan]), np.array([0.0, -1.0, 0.0, 2.0, 3.0, np.nan, 1.0]))
    with np.errstate(all="ignore"):
        np.testing.assert_array_equal(sp, np.power(np.array([0.0, -0.0, np.nan, np.inf, -np.inf, 5.0, np.nan]), np.array([0.0, -1.0, 0.0, 2.0, 3.0, np.nan, 1.0])))
