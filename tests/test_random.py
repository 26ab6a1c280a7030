"""RandomVariable draws: the reference's numbers come from NumPy's sequential
samplers and cannot be reproduced by a parallel device sampler; what is pinned instead:

* CPU: the Philox restatement against ``numpy.random.Philox`` bit for bit; uniform draws against
  ``Generator(Philox).random``; every sampler against the distribution the reference draws from
  (scipy.stats, Kolmogorov–Smirnov / exact pmf chi-square);
* GPU: the device kernels against that restatement (bit-exact for integer draws and uniforms,
  rtol 1e-12 elsewhere), through the C-ABI and through the executor.
"""
import numpy as np
import pytest
import scipy.stats as st

import np_graph
import philox_ref
from pytensor_amd.ir import Graph

KEY = [0x0123456789ABCDEF, 0xFEDCBA9876543210]


def gen(counter=0, key=KEY):
    return philox_ref.make_generator(key, counter)


def rv_graph(name, dtype, size, param_specs):
    """(rng, *params) -> (next_rng, draws)"""
    g = Graph(name=f"rv_{name}")
    r = g.new_var("object", (), kind="rng", name="rng")
    sz = g.new_var("int64", (len(size),), const=np.asarray(size, dtype="int64")) if size is not None else g.new_var("object", (), kind="none")
    ps = [g.new_var(dt, (None,) * nd) for dt, nd in param_specs]
    r2 = g.new_var("object", (), kind="rng")
    out_nd = len(size) if size is not None else max([nd for _, nd in param_specs] + [0]) - (1 if name == "categorical" else 0)
    out = g.new_var(dtype, (None,) * out_nd)
    g.add_node("RandomVariable", {"name": name, "dtype": dtype, "size_is_none": size is None}, [r, sz, *ps], [r2, out])
    g.inputs, g.outputs = [r, *ps], [r2, out]
    return g


# name -> (parameter values, scipy frozen distribution, discrete?)
CASES = {
    "uniform": ([-1.5, 2.25], st.uniform(-1.5, 3.75), False),
    "normal": ([0.5, 2.0], st.norm(0.5, 2.0), False),
    "halfnormal": ([1.0, 0.5], st.halfnorm(1.0, 0.5), False),
    "lognormal": ([0.2, 0.7], st.lognorm(0.7, scale=np.exp(0.2)), False),
    "exponential": ([1.7], st.expon(scale=1.7), False),
    "laplace": ([-0.3, 1.2], st.laplace(-0.3, 1.2), False),
    "logistic": ([0.4, 0.8], st.logistic(0.4, 0.8), False),
    "cauchy": ([0.1, 0.6], st.cauchy(0.1, 0.6), False),
    "halfcauchy": ([0.0, 1.5], st.halfcauchy(0.0, 1.5), False),
    "gumbel": ([0.3, 1.1], st.gumbel_r(0.3, 1.1), False),
    "weibull": ([1.8], st.weibull_min(1.8), False),
    "pareto": ([2.5, 1.5], st.pareto(2.5, scale=1.5), False),
    "triangular": ([-1.0, 0.5, 2.0], st.triang(c=0.5, loc=-1.0, scale=3.0), False),
    "gamma": ([2.7, 1.3], st.gamma(2.7, scale=1.3), False),
    "gamma_small": ([0.4, 2.0], st.gamma(0.4, scale=2.0), False),
    "beta": ([0.7, 2.2], st.beta(0.7, 2.2), False),
    "invgamma": ([3.0, 2.0], st.invgamma(3.0, scale=2.0), False),
    "t": ([5.0, 0.5, 1.5], st.t(5.0, 0.5, 1.5), False),
    "bernoulli": ([0.3], st.bernoulli(0.3), True),
    "geometric": ([0.25], st.geom(0.25), True),
    "poisson": ([3.6], st.poisson(3.6), True),
    "poisson_large": ([47.5], st.poisson(47.5), True),
    "integers": ([-3, 9], st.randint(-3, 9), True),
    "binomial": ([12, 0.3], st.binom(12, 0.3), True),
    "binomial_large": ([400, 0.35], st.binom(400, 0.35), True),
    "binomial_flip": ([60, 0.85], st.binom(60, 0.85), True),
    "negative_binomial": ([5.0, 0.4], st.nbinom(5.0, 0.4), True),
}


def _name(case):
    return "negative_binomial" if case.startswith("negative_binomial") else case.split("_")[0]


def _params(case):
    vals = CASES[case][0]
    dt = "int64" if _name(case) == "integers" else "float64"
    return [np.asarray(v, dtype=dt) for v in vals]


def _dtype(case):
    return "int64" if CASES[case][2] else "float64"


def check_distribution(case, x):
    vals, dist, discrete = CASES[case]
    x = np.asarray(x).ravel()
    if discrete:
        lo, hi = int(x.min()), int(x.max())
        ks = np.arange(lo, hi + 1)
        obs = np.array([(x == k).sum() for k in ks], dtype=float)
        exp = dist.pmf(ks) * len(x)
        keep = exp >= 5
        chi2 = ((obs[keep] - exp[keep]) ** 2 / exp[keep]).sum()
        pval = st.chi2.sf(chi2, max(int(keep.sum()) - 1, 1))
        assert pval > 1e-4, (case, chi2, pval)
        assert abs(obs[~keep].sum() - exp[~keep].sum()) < 6 * np.sqrt(exp[~keep].sum() + 1) + 5
    else:
        pval = st.kstest(x, dist.cdf).pvalue
        assert pval > 1e-4, (case, pval)


def test_philox_block_is_numpy_philox():
    for counter in (0, 5, (1 << 64) - 2, (1 << 130) + 77):
        g = gen(counter)
        raw = g.bit_generator.random_raw(8)
        mine = philox_ref.philox_block(counter + 1, *KEY) + philox_ref.philox_block(counter + 2, *KEY)
        assert [int(v) for v in raw] == mine


def test_uniform_draws_are_generator_random():
    n = 1001
    g2, x = philox_ref.draw("uniform", gen(3), (n,), [np.asarray(0.0), np.asarray(1.0)], "float64")
    assert np.array_equal(x, gen(3).random(n))
    assert philox_ref.generator_state(g2) == (KEY, 3 + (n + 3) // 4)
    # the advanced generator continues where a fresh numpy generator at that counter would
    _, y = philox_ref.draw("uniform", g2, (8,), [np.asarray(-1.0), np.asarray(1.0)], "float64")
    assert np.array_equal(y, -1.0 + 2.0 * gen(3 + (n + 3) // 4).random(8))


@pytest.mark.parametrize("case", sorted(CASES))
def test_sampler_matches_the_reference_distribution(case):
    n = 6000
    g = rv_graph(_name(case), _dtype(case), (n,), [(str(p.dtype), 0) for p in _params(case)])
    g2, x = np_graph.run_graph(g, [gen(11), *_params(case)])
    assert x.shape == (n,) and str(x.dtype) == _dtype(case)
    assert philox_ref.generator_state(g2)[1] == 11 + ((n + 3) // 4 if case == "uniform" else n)
    check_distribution(case, x)


def test_categorical_and_broadcast_parameters():
    p = np.array([[0.1, 0.2, 0.7], [0.6, 0.3, 0.1]])
    g = rv_graph("categorical", "int64", (4000, 2), [("float64", 2)])
    g2, x = np_graph.run_graph(g, [gen(), p])
    assert x.shape == (4000, 2)
    for col in range(2):
        freq = np.bincount(x[:, col], minlength=3) / 4000
        assert np.abs(freq - p[col]).max() < 0.03
    # size=None: batch shape from the broadcast parameters
    g = rv_graph("normal", "float64", None, [("float64", 1), ("float64", 2)])
    _, y = np_graph.run_graph(g, [gen(), np.array([0.0, 100.0, -100.0]), np.full((500, 1), 0.5)])
    assert y.shape == (500, 3) and np.abs(y.mean(axis=0) - [0.0, 100.0, -100.0]).max() < 0.1


def _mv_cases():
    rng = np.random.default_rng(3)
    B = rng.normal(size=(4, 4))
    return np.array([0.5, 2.0, 1.0, 4.0]), np.array([1.0, -2.0, 0.5, 3.0]), B @ B.T + np.eye(4)


def test_dirichlet_and_multivariate_normal_moments():
    alpha, mean, cov = _mv_cases()
    g = rv_graph("dirichlet", "float64", (5000,), [("float64", 1)])
    g.vars[g.outputs[1]].shape = (None, None)
    g2, x = np_graph.run_graph(g, [gen(2), alpha])
    assert x.shape == (5000, 4) and np.allclose(x.sum(axis=1), 1.0)
    assert np.abs(x.mean(axis=0) - alpha / alpha.sum()).max() < 0.01
    a0 = alpha.sum()
    assert np.abs(x.var(axis=0) - alpha * (a0 - alpha) / (a0 * a0 * (a0 + 1))).max() < 0.003
    assert philox_ref.generator_state(g2)[1] == 2 + 20000
    g = rv_graph("multivariate_normal", "float64", (6000,), [("float64", 1), ("float64", 2)])
    g.vars[g.outputs[1]].shape = (None, None)
    _, y = np_graph.run_graph(g, [gen(2), mean, cov])
    assert y.shape == (6000, 4)
    assert np.abs(y.mean(axis=0) - mean).max() < 0.12 and np.abs(np.cov(y.T) - cov).max() < 0.35


def test_successive_nodes_draw_disjoint_blocks():
    _, a = philox_ref.draw("normal", gen(0), (64,), [np.asarray(0.0), np.asarray(1.0)], "float64")
    g1, _ = philox_ref.draw("normal", gen(0), (32,), [np.asarray(0.0), np.asarray(1.0)], "float64")
    _, b = philox_ref.draw("normal", g1, (32,), [np.asarray(0.0), np.asarray(1.0)], "float64")
    assert np.array_equal(a[32:], b)  # element i always owns block counter+1+i


def test_boundary_rekeys_other_bit_generators():
    from pytensor_amd.rng import RngState

    s1 = RngState.from_generator(np.random.default_rng(7))
    s2 = RngState.from_generator(np.random.default_rng(7))
    s3 = RngState.from_generator(np.random.default_rng(8))
    assert s1.key == s2.key and s1.counter == 0 and s1.key != s3.key
    ph = gen(41)
    s = RngState.from_generator(ph)
    assert (list(s.key), s.counter) == (KEY, 41)
    assert philox_ref.generator_state(s.advanced(3).to_generator()) == (KEY, 44)


# ---------------------------------------------------------------------------- GPU


@pytest.fixture(scope="module")
def hip():
    from pytensor_amd import ffi

    ffi.init(0)
    return ffi


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(CASES))
def test_device_sampler_is_the_restatement(hip, case):
    from pytensor_amd.executor import HipExecutable

    n = 4099
    g = rv_graph(_name(case), _dtype(case), (n,), [(str(p.dtype), 0) for p in _params(case)])
    want_g, want = np_graph.run_graph(g, [gen(11), *_params(case)])
    exe = HipExecutable(g, auto_freeze=True)
    for _ in range(2):  # the second call must not replay captured counters
        got_g, got = exe(gen(11), *_params(case))
        assert philox_ref.generator_state(got_g) == philox_ref.generator_state(want_g)
        assert got.shape == want.shape and got.dtype == want.dtype
        if CASES[case][2] or case == "uniform":
            assert np.array_equal(got, want)
        else:
            np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12 * np.abs(want).max())
    if case == "uniform":
        assert np.array_equal(got, -1.5 + 3.75 * gen(11).random(n))
    # the advanced generator feeds the next call: different numbers, same distribution
    _, nxt = exe(got_g, *_params(case))
    assert not np.array_equal(nxt, got)
    check_distribution(case, nxt)


@pytest.mark.gpu
def test_device_categorical_float32_and_broadcast(hip):
    from pytensor_amd.executor import HipExecutable

    p = np.array([[0.1, 0.2, 0.7], [0.6, 0.3, 0.1]])
    g = rv_graph("categorical", "int64", (1000, 2), [("float64", 2)])
    want = np_graph.run_graph(g, [gen(5), p])
    got = HipExecutable(g)(gen(5), p)
    assert np.array_equal(got[1], want[1]) and philox_ref.generator_state(got[0]) == philox_ref.generator_state(want[0])
    g = rv_graph("normal", "float32", None, [("float32", 1), ("float32", 2)])
    ins = [np.array([0.0, 100.0, -100.0], dtype="float32"), np.full((257, 1), 0.5, dtype="float32")]
    want = np_graph.run_graph(g, [gen(5), *ins])
    got = HipExecutable(g)(gen(5), *ins)
    assert got[1].dtype == np.float32 and got[1].shape == (257, 3)
    np.testing.assert_allclose(got[1], want[1], rtol=1e-6, atol=1e-5)


@pytest.mark.gpu
def test_device_dirichlet_and_multivariate_normal(hip):
    from pytensor_amd.executor import HipExecutable

    alpha, mean, cov = _mv_cases()
    for name, ins, size in (("dirichlet", [alpha], (1000,)), ("multivariate_normal", [mean, cov], (1500,)), ("multivariate_normal", [mean, cov], None)):
        g = rv_graph(name, "float64", size, [("float64", a.ndim) for a in ins])
        want = np_graph.run_graph(g, [gen(4), *ins])
        got = HipExecutable(g)(gen(4), *ins)
        assert philox_ref.generator_state(got[0]) == philox_ref.generator_state(want[0])
        np.testing.assert_allclose(got[1], want[1], rtol=1e-11, atol=1e-11)


@pytest.mark.gpu
def test_device_large_uniform_and_normal_moments(hip):
    from pytensor_amd.executor import HipExecutable

    n = 1 << 22
    g = rv_graph("uniform", "float64", (n,), [("float64", 0), ("float64", 0)])
    _, u = HipExecutable(g)(gen(1 << 70), np.asarray(0.0), np.asarray(1.0))
    assert np.array_equal(u[:4096], gen(1 << 70).random(4096)) and np.array_equal(u[-3:], gen(1 << 70).random(n)[-3:])
    g = rv_graph("normal", "float64", (n,), [("float64", 0), ("float64", 0)])
    _, z = HipExecutable(g)(gen(9), np.asarray(0.0), np.asarray(1.0))
    assert abs(z.mean()) < 5 / np.sqrt(n) and abs(z.var() - 1) < 8 * np.sqrt(2 / n)
    assert abs(st.skew(z)) < 0.01 and abs(st.kurtosis(z)) < 0.02
