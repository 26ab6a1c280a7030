"""TEST INFRASTRUCTURE ONLY — scalar special functions as the reference's *C* backend computes them.

The reference's ``impl`` of these ops calls SciPy, its ``c_code`` calls small C files shipped with
the package; the golden vectors hold the C linker's values (the default runtime), so the oracle
restates the C algorithms (IEEE doubles, libm through ``math``):

* ``gammap`` / ``gammaq`` — regularised incomplete gamma P and Q: power series below ``x < k+1``,
  Lentz continued fraction above, both scaled by ``exp(k log x - x - logGamma(k))`` where logGamma
  is a 9-term Lanczos sum with a table for (half-)integers
  (pytensor/scalar/c_code/gamma.c: tables 62-80, logGamma 83-106, _series 143-155, _cfrac 172-189,
  GammaP 207-218, GammaQ 222-233; called by GammaInc/GammaIncC.c_code, scalar/math.py:648-655, 695-702).
* ``betainc`` — regularised incomplete beta (Cephes incbet): power series, two continued
  fractions, the symmetry flip (pytensor/scalar/c_code/incbet.c: BetaInc 33-90, incbcf 96-178,
  incbd 184-268, pseries 274-311; called by BetaInc.c_code, scalar/math.py:1371-1381).
* ``trigamma`` — AS 121 with 10-digit constants (TriGamma.c_support_code, scalar/math.py:518-567).
"""
import math

import numpy as np

EPS = 2.2204460492503131e-16
TINY = EPS * EPS * EPS
MAXFACT = 170
MAXITER = 1024
LN_BASE = 2.71828182845904523536028747135
SQRT_PI = 1.77245385090551602729816748334
LN_PI = 1.14472988584940017414342735135
LN_SQRT_2PI = 0.918938533204672741780329736406


def gamma_tables():
    """log(i!) and log(Gamma(i + 1/2)), built by the same running products (gamma.c:62-80);
    the last half-integer slot is never written there and stays 0."""
    logfs = [0.0] * (MAXFACT + 1)
    loghs = [0.0] * (MAXFACT + 1)
    x = 1.0
    for i in range(2, MAXFACT + 1):
        x *= i
        logfs[i] = math.log(x)
    x = SQRT_PI
    loghs[0] = 0.5 * LN_PI
    for i in range(1, MAXFACT):
        x *= i - 0.5
        loghs[i] = math.log(x)
    return logfs, loghs


_LOGFS, _LOGHS = gamma_tables()
LANCZOS = (
    0.99999999999980993227684700473478,
    676.520368121885098567009190444019,
    -1259.13921672240287047156078755283,
    771.3234287776530788486528258894,
    -176.61502916214059906584551354,
    12.507343278686904814458936853,
    -0.13857109526572011689554707,
    9.984369578019570859563e-6,
    1.50563273514931155834e-7,
)


def log_gamma(n):
    if n <= 0:
        return math.nan
    if n < MAXFACT + 1 + 4 * EPS:
        if abs(n - math.floor(n)) < 4 * EPS:
            return _LOGFS[max(int(math.floor(n)) - 1, 0)]
        if abs(2 * n - math.floor(2 * n)) < 4 * EPS:
            return _LOGHS[int(math.floor(n))]
    s = LANCZOS[0]
    for j in range(1, 9):
        s += LANCZOS[j] / (n + j)
    return (n + 0.5) * math.log((n + 7.5) / LN_BASE) + (LN_SQRT_2PI + math.log(s / n) - 7.0)


def _series(n, x):
    t = 1.0 / n
    s = t
    for _ in range(MAXITER):
        n += 1.0
        t *= x / n
        s += t
        if abs(t) < abs(s) * EPS:
            break
    return s


def _cfrac(n, x):
    b = x + 1.0 - n
    c = 1.0 / TINY
    d = 1.0 / b
    f = d
    for i in range(1, MAXITER):
        a = i * (n - i)
        b += 2.0
        d = a * d + b
        if abs(d) < TINY:
            d = TINY
        c = b + a / c
        if abs(c) < TINY:
            c = TINY
        d = 1.0 / d
        e = d * c
        f *= e
        if abs(e - 1.0) < EPS:
            break
    return f


def _scale(n, x):
    try:
        return math.exp(n * math.log(x) - x - log_gamma(n))
    except OverflowError:
        return math.inf


def gammap(n, x):
    if math.isnan(n) or math.isnan(x):
        return math.nan
    if n <= 0 or x < 0:
        return math.nan
    if x <= 0:
        return 0.0
    if math.isinf(n):
        return math.nan if math.isinf(x) else 0.0
    if math.isinf(x):
        return 1.0
    if x < n + 1:
        return _series(n, x) * _scale(n, x)
    return 1.0 - _cfrac(n, x) * _scale(n, x)


def gammaq(n, x):
    if math.isnan(n) or math.isnan(x):
        return math.nan
    if n <= 0 or x < 0:
        return math.nan
    if x <= 0:
        return 1.0
    if math.isinf(n):
        return math.nan if math.isinf(x) else 1.0
    if math.isinf(x):
        return 0.0
    if x < n + 1:
        return 1.0 - _series(n, x) * _scale(n, x)
    return _cfrac(n, x) * _scale(n, x)


# ---- incomplete beta ----
MINLOG = -7.451332191019412076235e2
MAXLOG = 7.09782712893383996732e2
MAXGAM = 171.624376956302725
BEPS = 1.11022302462515654042e-16
BIG = 4.503599627370496e15
BIGINV = 2.22044604925031308085e-16


def _cf(x_or_z, k, steps):
    """the shared three-term recurrence of the two continued fractions; ``k`` holds k1..k8 and
    ``steps`` their increments"""
    pkm2, qkm2, pkm1, qkm1 = 0.0, 1.0, 1.0, 1.0
    ans = r = 1.0
    thresh = 3.0 * BEPS
    k = list(k)
    for _ in range(300):
        xk = -(x_or_z * k[0] * k[1]) / (k[2] * k[3])
        pk = pkm1 + pkm2 * xk
        qk = qkm1 + qkm2 * xk
        pkm2, pkm1, qkm2, qkm1 = pkm1, pk, qkm1, qk
        xk = (x_or_z * k[4] * k[5]) / (k[6] * k[7])
        pk = pkm1 + pkm2 * xk
        qk = qkm1 + qkm2 * xk
        pkm2, pkm1, qkm2, qkm1 = pkm1, pk, qkm1, qk
        if qk != 0.0:
            r = pk / qk
        if r != 0.0:
            t = abs((ans - r) / r)
            ans = r
        else:
            t = 1.0
        if t < thresh:
            break
        for j in range(8):
            k[j] += steps[j]
        if abs(qk) + abs(pk) > BIG:
            pkm2 *= BIGINV
            pkm1 *= BIGINV
            qkm2 *= BIGINV
            qkm1 *= BIGINV
        if abs(qk) < BIGINV or abs(pk) < BIGINV:
            pkm2 *= BIG
            pkm1 *= BIG
            qkm2 *= BIG
            qkm1 *= BIG
    return ans


def _incbcf(a, b, x):
    return _cf(x, (a, a + b, a, a + 1.0, 1.0, b - 1.0, a + 1.0, a + 2.0), (1.0, 1.0, 2.0, 2.0, 1.0, -1.0, 2.0, 2.0))


def _incbd(a, b, x):
    z = x / (1.0 - x)
    return _cf(z, (a, b - 1.0, a, a + 1.0, 1.0, a + b, a + 1.0, a + 2.0), (1.0, -1.0, 2.0, 2.0, 1.0, 1.0, 2.0, 2.0))


def _pseries(a, b, x):
    ai = 1.0 / a
    u = (1.0 - b) * x
    v = u / (a + 1.0)
    t1 = v
    t = u
    n = 2.0
    s = 0.0
    z = BEPS * ai
    while abs(v) > z:
        u = (n - b) * x / n
        t *= u
        v = t / (a + n)
        s += v
        n += 1.0
    s += t1
    s += ai
    u = a * math.log(x)
    if (a + b) < MAXGAM and abs(u) < MAXLOG:
        t = math.gamma(a + b) / (math.gamma(a) * math.gamma(b))
        s = s * t * math.pow(x, a)
    else:
        t = math.lgamma(a + b) - math.lgamma(a) - math.lgamma(b) + u + math.log(s)
        s = 0.0 if t < MINLOG else math.exp(t)
    return s


def betainc(a, b, x, _depth=0):
    if math.isnan(a) or math.isnan(b) or math.isnan(x):
        return math.nan
    if a <= 0.0 or b <= 0.0 or x < 0.0 or 1.0 < x:
        return math.nan
    if x == 0.0:
        return 0.0
    if x == 1.0:
        return 1.0
    if b * x <= 1.0 and x <= 0.95:
        return _pseries(a, b, x)
    xc = 1.0 - x
    if x > a / (a + b) and _depth == 0:
        t = betainc(b, a, xc, 1)
        return 1.0 - BEPS if t <= BEPS else 1.0 - t
    y = x * (a + b - 2.0) - (a - 1.0)
    w = _incbcf(a, b, x) if y < 0.0 else _incbd(a, b, x) / xc
    y = a * math.log(x)
    t = b * math.log(xc)
    if (a + b) < MAXGAM and abs(y) < MAXLOG and abs(t) < MAXLOG:
        t = math.pow(xc, b)
        t *= math.pow(x, a)
        t /= a
        t *= w
        t *= math.gamma(a + b) / (math.gamma(a) * math.gamma(b))
        return t
    y += t + math.lgamma(a + b) - math.lgamma(a) - math.lgamma(b)
    y += math.log(w / a)
    return 0.0 if y < MINLOG else math.exp(y)


def trigamma(x):
    if math.isnan(x):
        return math.nan
    a, b = 0.0001, 5.0
    b2, b4, b6, b8 = 0.1666666667, -0.03333333333, 0.02380952381, -0.03333333333
    if x <= 0:
        return 0.0
    if x <= a:
        return 1.0 / x / x
    value = 0.0
    z = x
    while z < b:
        value += 1.0 / z / z
        z += 1.0
    y = 1.0 / z / z
    value += 0.5 * y + (1.0 + y * (b2 + y * (b4 + y * (b6 + y * b8)))) / z
    return value


def _lift(fn, nin):
    uf = np.frompyfunc(lambda *a: fn(*(float(v) for v in a)), nin, 1)

    def call(*args):
        # (the C code computes in double whatever the storage type; the caller casts the result)
        return np.asarray(uf(*[np.asarray(a, dtype=np.float64) for a in args]), dtype=np.float64)

    return call


GammaInc = _lift(gammap, 2)
GammaIncC = _lift(gammaq, 2)
BetaInc = _lift(betainc, 3)
TriGamma = _lift(trigamma, 1)
