"""TEST INFRASTRUCTURE ONLY — CPU restatement of the device samplers (pytensor_amd/csrc/random.hip).

The reference draws from ``numpy.random.Generator`` methods (RandomVariable.perform,
pytensor/tensor/random/op.py; rng_fn of the classes in random/basic.py); those samplers are
sequential (ziggurat / rejection loops on one stream) and cannot be matched by a parallel device
sampler, so for this row the oracle restates the *device* algorithm, and is
pinned two ways: ``philox_block`` against ``numpy.random.Philox().random_raw`` and the uniform
draws against ``Generator(Philox).random`` bit for bit; every sampler's output against the
distribution the reference draws from (Kolmogorov–Smirnov / moment tests in tests/test_random.py).
Plain Python integers and floats; small sizes only.
"""
import math

import numpy as np

MASK = (1 << 64) - 1
M0, M1 = 0xD2E7470EE14C6C93, 0xCA5A826395121157
W0, W1 = 0x9E3779B97F4A7C15, 0xBB67AE8584CAA73B


def philox_block(counter: int, k0: int, k1: int):
    """Philox4x64-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3",
    SC'11) of a 256-bit counter under a 128-bit key, the variant numpy.random.Philox implements."""
    c = [(counter >> (64 * j)) & MASK for j in range(4)]
    for r in range(10):
        if r:
            k0 = (k0 + W0) & MASK
            k1 = (k1 + W1) & MASK
        p0 = M0 * c[0]
        p1 = M1 * c[2]
        c = [(p1 >> 64) ^ c[1] ^ k0, p1 & MASK, (p0 >> 64) ^ c[3] ^ k1, p0 & MASK]
    return c


def generator_state(gen):
    st = gen.bit_generator.state
    assert st["bit_generator"] == "Philox", "the oracle takes the Philox generators the boundary produces"
    key = [int(w) for w in st["state"]["key"]]
    ctr = 0
    for j, w in enumerate(st["state"]["counter"]):
        ctr |= int(w) << (64 * j)
    return key, ctr


def make_generator(key, counter: int):
    words = np.array([(counter >> (64 * j)) & MASK for j in range(4)], dtype=np.uint64)
    return np.random.Generator(np.random.Philox(key=np.array(key, dtype=np.uint64), counter=words))


class Stream:
    """blocks of element ``i``: counter + 1 + i under key (k0 ^ W0*substream, k1 + attempt)"""

    def __init__(self, key, counter):
        self.key, self.counter = key, counter

    def block(self, i, sub=0, attempt=0):
        ctr = (self.counter + 1 + i) & ((1 << 256) - 1)
        return philox_block(ctr, self.key[0] ^ ((W0 * sub) & MASK), (self.key[1] + attempt) & MASK)


def u53(w):
    return (w >> 11) * (1.0 / 9007199254740992.0)


def uopen(w):
    return ((w >> 12) + 0.5) * (1.0 / 4503599627370496.0)


def box_muller(w0, w1):
    return math.sqrt(-2.0 * math.log(uopen(w0))) * math.cos(6.283185307179586 * uopen(w1))


def gamma_mt(s: Stream, i, sub, shape):
    if not shape > 0.0:
        return 0.0 if shape == 0.0 else math.nan
    w = s.block(i, sub, 0)
    boost = 1.0
    if shape < 1.0:
        boost = math.pow(uopen(w[3]), 1.0 / shape)
        shape += 1.0
    d = shape - 1.0 / 3.0
    c = 1.0 / math.sqrt(9.0 * d)
    for attempt in range(64):
        if attempt:
            w = s.block(i, sub, attempt)
        z = box_muller(w[0], w[1])
        v = 1.0 + c * z
        if v <= 0.0:
            continue
        v = v * v * v
        u = uopen(w[2])
        if math.log(u) < 0.5 * z * z + d - d * v + d * math.log(v):
            return d * v * boost
    return d * boost


def poisson_draw(s: Stream, i, lam, sub=0):
    if not lam >= 0.0:
        return math.nan
    if lam == 0.0:
        return 0.0
    if lam < 10.0:
        L = math.exp(-lam)
        p = 1.0
        k = 0
        for attempt in range(64):
            w = s.block(i, sub, attempt)
            for j in range(4):
                p *= uopen(w[j])
                if p <= L:
                    return float(k)
                k += 1
        return float(k)
    slam, loglam = math.sqrt(lam), math.log(lam)
    b = 0.931 + 2.53 * slam
    al = -0.059 + 0.02483 * b
    invalpha = 1.1239 + 1.1328 / (b - 3.4)
    vr = 0.9277 - 3.6224 / (b - 2.0)
    for attempt in range(256):
        w = s.block(i, sub, attempt)
        U = uopen(w[0]) - 0.5
        V = uopen(w[1])
        us = 0.5 - abs(U)
        k = math.floor((2.0 * al / us + b) * U + lam + 0.43)
        if us >= 0.07 and V <= vr:
            return float(k)
        if k < 0 or (us < 0.013 and V > us):
            continue
        if math.log(V) + math.log(invalpha) - math.log(al / (us * us) + b) <= -lam + k * loglam - math.lgamma(k + 1.0):
            return float(k)
    return float(math.floor(lam))


def binomial_draw(s: Stream, i, n, p, sub=0):
    if not (0.0 <= p <= 1.0) or not n >= 0.0:
        return math.nan
    n = math.floor(n)
    flip = p > 0.5
    q = 1.0 - p if flip else p
    if q == 0.0 or n == 0.0:
        return n if flip else 0.0
    x = -1.0
    if n * q < 10.0:
        qn = math.exp(n * math.log1p(-q))
        odds = q / (1.0 - q)
        bound = min(n, n * q + 10.0 * math.sqrt(n * q * (1.0 - q) + 1.0))
        for attempt in range(64):
            if x >= 0.0:
                break
            w = s.block(i, sub, attempt)
            for j in range(4):
                if x >= 0.0:
                    break
                u, px, k, ok = uopen(w[j]), qn, 0.0, True
                while u > px:
                    k += 1.0
                    if k > bound:
                        ok = False
                        break
                    u -= px
                    px *= (n - k + 1.0) * odds / k
                if ok:
                    x = k
        if x < 0.0:
            x = math.floor(n * q)
    else:
        spq = math.sqrt(n * q * (1.0 - q))
        b = 1.15 + 2.53 * spq
        al = -0.0873 + 0.0248 * b + 0.01 * q
        c = n * q + 0.5
        vr = 0.92 - 4.2 / b
        alpha = (2.83 + 5.1 / b) * spq
        lpq = math.log(q / (1.0 - q))
        m = math.floor((n + 1.0) * q)
        h = math.lgamma(m + 1.0) + math.lgamma(n - m + 1.0)
        for attempt in range(256):
            w = s.block(i, sub, attempt)
            u = uopen(w[0]) - 0.5
            v = uopen(w[1])
            us = 0.5 - abs(u)
            k = math.floor((2.0 * al / us + b) * u + c)
            if k < 0.0 or k > n:
                continue
            if us >= 0.07 and v <= vr:
                x = k
                break
            v = math.log(v * alpha / (al / (us * us) + b))
            if v <= h - math.lgamma(k + 1.0) - math.lgamma(n - k + 1.0) + (k - m) * lpq:
                x = k
                break
        if x < 0.0:
            x = m
    return n - x if flip else x


def vonmises_draw(s: Stream, i, mu, kappa):
    PI = 3.141592653589793
    if not kappa >= 0.0:
        return math.nan
    w = s.block(i, 0, 0)
    if kappa < 1e-8:
        return PI * (2.0 * uopen(w[0]) - 1.0)
    if kappa > 1e6:
        res = mu + math.sqrt(1.0 / kappa) * box_muller(w[0], w[1])
    else:
        sh = 0.5 / kappa
        r = sh + math.sqrt(1.0 + sh * sh)
        W, sign = 1.0, w[2]
        for attempt in range(256):
            if attempt:
                w = s.block(i, 0, attempt)
            Z = math.cos(PI * uopen(w[0]))
            W = (1.0 + r * Z) / (r + Z)
            Y, V = kappa * (r - W), uopen(w[1])
            sign = w[2]
            if Y * (2.0 - Y) - V >= 0.0 or math.log(Y / V) + 1.0 - Y >= 0.0:
                break
        W = min(1.0, max(-1.0, W))
        res = math.acos(W)
        if sign >> 63:
            res = -res
        res += mu
    m = math.fmod(abs(res) + PI, 2.0 * PI) - PI
    return -m if res < 0.0 else m


def hypergeometric_draw(s: Stream, i, good, bad, sample):
    good, bad, sample = math.floor(good), math.floor(bad), math.floor(sample)
    if not (good >= 0 and bad >= 0 and sample >= 0) or sample > good + bad:
        return math.nan
    good, bad, sample = float(good), float(bad), float(sample)
    lo, hi = max(0.0, sample - bad), min(sample, good)
    if lo == hi:
        return lo
    u = uopen(s.block(i, 0, 0)[0])
    m = math.floor((sample + 1.0) * (good + 1.0) / (good + bad + 2.0))
    m = min(hi, max(lo, float(m)))
    lg = math.lgamma
    lpm = (lg(good + 1.0) - lg(m + 1.0) - lg(good - m + 1.0) + lg(bad + 1.0) - lg(sample - m + 1.0)
           - lg(bad - sample + m + 1.0) - lg(good + bad + 1.0) + lg(sample + 1.0) + lg(good + bad - sample + 1.0))
    pm = math.exp(lpm)
    u -= pm
    if u <= 0.0:
        return m
    kd = ku = m
    pd = pu = pm
    while kd > lo or ku < hi:
        if kd > lo:
            pd *= kd * (bad - sample + kd) / ((good - kd + 1.0) * (sample - kd + 1.0))
            kd -= 1.0
            u -= pd
            if u <= 0.0:
                return kd
        if ku < hi:
            pu *= (good - ku) * (sample - ku) / ((ku + 1.0) * (bad - sample + ku + 1.0))
            ku += 1.0
            u -= pu
            if u <= 0.0:
                return ku
    return m


def multinomial_row(s: Stream, i, n, row):
    k = len(row)
    out = [0] * k
    remaining, mass = float(n), 1.0
    for j in range(k - 1):
        pj = float(row[j])
        x = 0.0
        if remaining > 0.0:
            q = min(1.0, max(0.0, pj / mass)) if mass > 0.0 else 1.0
            x = binomial_draw(s, i, remaining, q, j)
        out[j] = int(x)
        remaining -= x
        mass -= pj
    out[k - 1] = int(remaining)
    return out


def _element(name, s: Stream, i, p):
    """one draw of a per-element distribution; ``p`` are this element's (float) parameters"""
    if name in ("normal", "halfnormal", "lognormal"):
        w = s.block(i)
        z = box_muller(w[0], w[1])
        if name == "halfnormal":
            z = abs(z)
        r = p[0] + p[1] * z
        return math.exp(r) if name == "lognormal" else r
    if name == "gamma":
        return gamma_mt(s, i, 0, p[0]) * p[1]
    if name == "beta":
        x, y = gamma_mt(s, i, 0, p[0]), gamma_mt(s, i, 1, p[1])
        return x / (x + y)
    if name == "invgamma":
        return p[1] / gamma_mt(s, i, 0, p[0])
    if name == "t":
        g = gamma_mt(s, i, 0, 0.5 * p[0])
        w = s.block(i, 1, 0)
        return p[1] + p[2] * (math.sqrt(0.5 * p[0]) * box_muller(w[0], w[1]) / math.sqrt(g))
    if name == "poisson":
        return poisson_draw(s, i, p[0])
    if name == "binomial":
        return binomial_draw(s, i, p[0], p[1])
    if name == "gengamma":
        return p[2] * math.pow(gamma_mt(s, i, 0, p[0] / p[1]), 1.0 / p[1])
    if name == "beta_binomial":
        x, y = gamma_mt(s, i, 0, p[1]), gamma_mt(s, i, 1, p[2])
        return binomial_draw(s, i, p[0], x / (x + y), 2)
    if name == "vonmises":
        return vonmises_draw(s, i, p[0], p[1])
    if name == "hypergeometric":
        return hypergeometric_draw(s, i, p[0], p[1], p[2])
    if name == "wald":
        w = s.block(i)
        mu, lam = p
        z = box_muller(w[0], w[1])
        y, d = mu * z * z, 0.5 * mu / lam
        x = mu + d * (y - math.sqrt(4.0 * lam * y + y * y))
        return x if uopen(w[2]) <= mu / (mu + x) else mu * mu / x
    if name == "negative_binomial":
        return poisson_draw(s, i, gamma_mt(s, i, 0, p[0]) * ((1.0 - p[1]) / p[1]), 1)
    w = s.block(i)
    u = uopen(w[0])
    if name == "exponential":
        return -p[0] * math.log(u)
    if name == "truncexpon":
        return p[1] + p[2] * -math.log1p(u * math.expm1(-p[0]))
    if name == "laplace":
        e = -math.log(u)
        return p[0] + p[1] * (e if (w[1] >> 63) else -e)
    if name == "logistic":
        return p[0] + p[1] * math.log(u / (1.0 - u))
    if name == "cauchy":
        return p[0] + p[1] * math.tan(3.141592653589793 * (u - 0.5))
    if name == "halfcauchy":
        return p[0] + p[1] * math.tan(1.5707963267948966 * u)
    if name == "gumbel":
        return p[0] - p[1] * math.log(-math.log(u))
    if name == "weibull":
        return math.pow(-math.log(u), 1.0 / p[0])
    if name == "pareto":
        return p[1] * math.exp(-math.log(u) / p[0])
    if name == "triangular":
        l, m, h = p
        fc = (m - l) / (h - l)
        return l + math.sqrt(u * (h - l) * (m - l)) if u < fc else h - math.sqrt((1.0 - u) * (h - l) * (h - m))
    if name == "bernoulli":
        return 1.0 if u < p[0] else 0.0
    if name == "geometric":
        r = 1.0 if p[0] >= 1.0 else math.ceil(math.log(u) / math.log1p(-p[0]))
        return max(r, 1.0)
    raise NotImplementedError(name)


def random_order(s: Stream, n, weights=None):
    """(stable argsort of n random keys, blocks consumed): uniform keys (Generator.random's numbers)
    or Exp(1) / weight keys"""
    if weights is None:
        keys = np.empty(n)
        for b in range((n + 3) // 4):
            w = s.block(b)
            for j in range(4):
                if 4 * b + j < n:
                    keys[4 * b + j] = u53(w[j])
        return np.argsort(keys, kind="stable"), keys, (n + 3) // 4
    e = np.array([-1.0 * math.log(uopen(s.block(i)[0])) for i in range(n)])
    with np.errstate(divide="ignore"):
        keys = e / np.asarray(weights, dtype=np.float64)
    return np.argsort(keys, kind="stable"), keys, n


def draw(name, gen, size, params, dtype, ndims_params=None, method="cholesky"):
    """(advanced generator, draws) — what the ``RandomVariable`` node of the hip linker returns"""
    key, ctr = generator_state(gen)
    s = Stream(key, ctr)
    dtype = np.dtype(dtype)
    if name == "permutation":
        x = np.asarray(params[0])
        core = 1 if ndims_params is None else int(ndims_params[0])
        if x.ndim != core:
            raise NotImplementedError("permutation with batch dimensions")
        n = int(x) if core == 0 else x.shape[0]
        order, _, blocks = random_order(s, n)
        return make_generator(key, ctr + blocks), (order if core == 0 else x[order]).astype(dtype)
    if name == "choice_without_replacement":
        a, *rest, core_shape = params
        a = np.asarray(a)
        pr = np.asarray(rest[0]) if rest else None
        core = a.ndim if ndims_params is None else int(ndims_params[0])
        if a.ndim != core:
            raise NotImplementedError("choice without replacement with batch dimensions")
        core_shape = tuple(int(v) for v in np.asarray(core_shape).ravel())
        take = int(np.prod(core_shape)) if core_shape else 1
        n = int(a) if core == 0 else a.shape[0]
        if take > n:
            raise ValueError("Cannot take a larger sample than population when replace is False")
        if pr is not None and pr.shape[0] != n:
            raise ValueError("a and p must have same size")
        order, keys, blocks = random_order(s, n, pr)
        if pr is not None and take and not np.isfinite(keys[order[take - 1]]):
            raise ValueError("Fewer non-zero entries in p than size")
        head = order[:take]
        out = head if core == 0 else a[head]
        return make_generator(key, ctr + blocks), out.reshape(*core_shape, *a.shape[1:]).astype(dtype)
    if name == "categorical":
        (pr,) = params
        pr = np.asarray(pr)
        batch = pr.shape[:-1]
        shape = tuple(batch) if size is None else tuple(size)
        prb = np.broadcast_to(pr, (*shape, pr.shape[-1])).reshape(-1, pr.shape[-1])
        out = np.empty(len(prb), dtype=np.int64)
        for i, row in enumerate(prb):
            u = uopen(s.block(i)[0])
            acc, pick = 0.0, len(row) - 1
            for j, pj in enumerate(row):
                acc += float(pj)
                if u < acc:
                    pick = j
                    break
            out[i] = pick
        return make_generator(key, ctr + len(prb)), out.reshape(shape).astype(dtype)
    if name == "multinomial":
        nn, pr = (np.asarray(v) for v in params)
        batch = np.broadcast_shapes(nn.shape, pr.shape[:-1])
        shape = tuple(batch) if size is None else tuple(int(v) for v in size)
        k = pr.shape[-1]
        nb = np.broadcast_to(nn, shape).reshape(-1)
        prb = np.broadcast_to(pr, (*shape, k)).reshape(-1, k)
        out = np.array([multinomial_row(s, i, int(nb[i]), prb[i]) for i in range(len(prb))], dtype=np.int64).reshape(*shape, k)
        return make_generator(key, ctr + len(prb)), out.astype(dtype)
    if name == "dirichlet":
        al = np.asarray(params[0], dtype=np.float64)
        shape = (*(al.shape[:-1] if size is None else tuple(size)), al.shape[-1])
        flat = np.broadcast_to(al, shape).reshape(-1)
        g = np.array([gamma_mt(s, i, 0, float(a)) for i, a in enumerate(flat)]).reshape(shape)
        return make_generator(key, ctr + g.size), (g / g.sum(axis=-1, keepdims=True)).astype(dtype)
    if name == "multivariate_normal":
        mean, cov = (np.asarray(p, dtype=np.float64) for p in params)
        k = mean.shape[-1]
        lead = tuple(mean.shape[:-1]) if size is None else tuple(int(v) for v in size)
        rows = int(np.prod(lead)) if lead else 1
        mean = np.broadcast_to(mean, (*lead, k)).reshape(rows, k)
        z = np.array([box_muller(*s.block(i)[:2]) for i in range(rows * k)]).reshape(rows, k)
        if method == "cholesky":
            A = np.linalg.cholesky(cov)
        elif method == "svd":  # (the signs of the columns are LAPACK's choice: only the distribution is pinned)
            A, sv, _ = np.linalg.svd(cov)
            A = A * np.sqrt(sv)[None, :]
        else:
            w, A = np.linalg.eigh(cov)
            A = A * np.sqrt(w)[None, :]
        out = z @ A.T + mean
        return make_generator(key, ctr + rows * k), out.reshape(*lead, k).astype(dtype)
    params = [np.asarray(p) for p in params]
    bshape = np.broadcast_shapes(*[p.shape for p in params]) if params else ()
    shape = tuple(bshape) if size is None else tuple(int(v) for v in size)
    n = int(np.prod(shape)) if shape else 1
    flat = [np.broadcast_to(p, shape).reshape(-1) for p in params]
    if name == "uniform":
        u = np.empty(n)
        for b in range((n + 3) // 4):
            w = s.block(b)
            for j in range(4):
                if 4 * b + j < n:
                    u[4 * b + j] = u53(w[j])
        lo, hi = flat[0].astype(np.float64), flat[1].astype(np.float64)
        out = lo + (hi - lo) * u
        return make_generator(key, ctr + (n + 3) // 4), out.reshape(shape).astype(dtype)
    if name == "integers":
        out = np.empty(n, dtype=np.int64)
        for i in range(n):
            lo, hi = int(flat[0][i]), int(flat[1][i])
            rng_ = (hi - lo) & MASK
            out[i] = lo + ((s.block(i)[0] * rng_) >> 64)
        return make_generator(key, ctr + n), out.reshape(shape).astype(dtype)
    out = np.empty(n, dtype=np.float64)
    for i in range(n):
        out[i] = _element(name, s, i, [float(f[i]) for f in flat])
    return make_generator(key, ctr + n), out.reshape(shape).astype(dtype)
