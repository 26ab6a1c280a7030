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
    out_nd = len(size) if size is not None else max([nd for _, nd in param_specs] + [0]) - (1 if name in ("categorical", "multinomial") else 0)
    out_nd += name == "multinomial"
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
    "wald": ([1.5, 2.5], st.invgauss(1.5 / 2.5, scale=2.5), False),
    "truncexpon": ([2.5, -0.5, 1.5], st.truncexpon(2.5, loc=-0.5, scale=1.5), False),
    "gengamma": ([3.0, 1.5, 0.8], st.gengamma(2.0, 1.5, scale=0.8), False),
    "beta_binomial": ([25, 2.0, 3.5], st.betabinom(25, 2.0, 3.5), True),
    "beta_binomial_large": ([300, 4.0, 1.5], st.betabinom(300, 4.0, 1.5), True),
    "vonmises": ([0.7, 2.5], st.vonmises(2.5, loc=0.7), False),
    "vonmises_wrapped": ([3.0, 0.6], None, False),
    "hypergeometric": ([30, 45, 20], st.hypergeom(75, 30, 20), True),
    "hypergeometric_large": ([4000, 9000, 2500], st.hypergeom(13000, 4000, 2500), True),
}
# von Mises on the circle: the draw lies in [-pi, pi] (wrapped around mu), scipy's cdf does not wrap
_VM = st.vonmises(0.6)
CASES["vonmises_wrapped"] = ([3.0, 0.6], None, False)
_MULTIWORD = ("negative_binomial", "beta_binomial")


def _name(case):
    for nm in _MULTIWORD:
        if case.startswith(nm):
            return nm
    return case.split("_")[0]


def _params(case):
    vals = CASES[case][0]
    dt = "int64" if _name(case) in ("integers", "hypergeometric") else "float64"
    out = [np.asarray(v, dtype=dt) for v in vals]
    if _name(case) == "beta_binomial":
        out[0] = np.asarray(vals[0], dtype="int64")
    return out


def _dtype(case):
    return "int64" if CASES[case][2] else "float64"


def check_distribution(case, x):
    vals, dist, discrete = CASES[case]
    x = np.asarray(x).ravel()
    if case == "vonmises_wrapped":
        assert x.min() >= -np.pi and x.max() <= np.pi
        y = np.mod(x - vals[0] + np.pi, 2 * np.pi) - np.pi   # angle relative to mu, in [-pi, pi)
        assert st.kstest(y, _VM.cdf).pvalue > 1e-4
        return
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


_MULTI_P = np.array([0.05, 0.4, 0.25, 0.3])


def check_multinomial(x, n):
    """counts sum to n; every marginal is Binomial(n, p_j) (exact pmf chi-square)"""
    assert x.shape[-1] == 4 and (x >= 0).all() and (x.sum(axis=-1) == n).all()
    for j, pj in enumerate(_MULTI_P):
        col = x[..., j].ravel()
        ks = np.arange(col.min(), col.max() + 1)
        obs = np.array([(col == k).sum() for k in ks], dtype=float)
        exp = st.binom(n, pj).pmf(ks) * len(col)
        keep = exp >= 5
        chi2 = ((obs[keep] - exp[keep]) ** 2 / exp[keep]).sum()
        assert st.chi2.sf(chi2, max(int(keep.sum()) - 1, 1)) > 1e-4, (j, chi2)
    # and the pairwise covariance -n p_i p_j
    flat = x.reshape(-1, 4).astype(float)
    c = np.cov(flat.T)
    want = -n * np.outer(_MULTI_P, _MULTI_P) + np.diag(n * _MULTI_P)
    assert np.abs(c - want).max() < 0.08 * n * 0.25 + 0.3


def test_multinomial_counts():
    g = rv_graph("multinomial", "int64", (5000,), [("int64", 0), ("float64", 1)])
    g2, x = np_graph.run_graph(g, [gen(2), np.asarray(40), _MULTI_P])
    assert x.shape == (5000, 4) and x.dtype == np.int64 and philox_ref.generator_state(g2)[1] == 2 + 5000
    check_multinomial(x, 40)
    # size=None: n broadcast against the batch dimensions of p; a zero-probability category stays empty
    g = rv_graph("multinomial", "int64", None, [("int64", 1), ("float64", 2)])
    p2 = np.array([[0.5, 0.0, 0.5], [0.2, 0.3, 0.5]])
    _, y = np_graph.run_graph(g, [gen(2), np.array([7, 900]), p2])
    assert y.shape == (2, 3) and list(y.sum(axis=-1)) == [7, 900] and y[0, 1] == 0


def order_graph(name, dtype, specs, ndims_params, out_nd=None):
    """permutation / choice_without_replacement node: (rng, *params) -> (next_rng, draws); size=None"""
    g = Graph(name=f"rv_{name}")
    r = g.new_var("object", (), kind="rng", name="rng")
    sz = g.new_var("object", (), kind="none")
    ps = [g.new_var(dt, (None,) * nd) for dt, nd in specs]
    r2 = g.new_var("object", (), kind="rng")
    if out_nd is None:
        out_nd = max(specs[0][1], 1)
    out = g.new_var(dtype, (None,) * out_nd)
    g.add_node("RandomVariable", {"name": name, "dtype": dtype, "size_is_none": True, "ndims_params": ndims_params}, [r, sz, *ps], [r2, out])
    g.inputs, g.outputs = [r, *ps], [r2, out]
    return g


def test_permutation_and_choice_without_replacement():
    # permutation of arange(n): a permutation; position of every element uniform
    g = order_graph("permutation", "int64", [("int64", 0)], [0])
    first = []
    gcur = gen(0)
    for _ in range(1200):
        gcur, x = np_graph.run_graph(g, [gcur, np.asarray(6)])
        assert sorted(x) == list(range(6))
        first.append(x[0] * 6 + x[5])
    obs = np.bincount(first, minlength=36).astype(float)
    obs = obs[[i * 6 + j for i in range(6) for j in range(6) if i != j]]
    assert st.chisquare(obs).pvalue > 1e-4
    assert philox_ref.generator_state(gcur)[1] == 1200 * 2
    # rows of a matrix are moved whole
    g = order_graph("permutation", "float64", [("float64", 2)], [2])
    m = np.arange(21.0).reshape(7, 3)
    _, y = np_graph.run_graph(g, [gen(3), m])
    assert y.shape == (7, 3) and sorted(y[:, 0]) == list(m[:, 0]) and np.array_equal(y[:, 1], y[:, 0] + 1)
    # weighted choice without replacement: ordered pairs follow successive sampling, p_i p_j / (1 - p_i)
    pr = np.array([0.5, 0.3, 0.15, 0.05])
    g = order_graph("choice_without_replacement", "int64", [("int64", 0), ("float64", 1), ("int64", 1)], [0, 1, 1])
    gcur, pairs = gen(1), []
    for _ in range(4000):
        gcur, x = np_graph.run_graph(g, [gcur, np.asarray(4), pr, np.array([2])])
        assert x.shape == (2,) and x[0] != x[1]
        pairs.append(x[0] * 4 + x[1])
    keep = [i * 4 + j for i in range(4) for j in range(4) if i != j]
    exp = np.array([pr[i] * pr[j] / (1 - pr[i]) for i in range(4) for j in range(4) if i != j]) * 4000
    assert st.chisquare(np.bincount(pairs, minlength=16)[keep], exp).pvalue > 1e-4
    # unweighted: elements of an array, 2-d core shape; errors as Generator.choice raises them
    g = order_graph("choice_without_replacement", "float64", [("float64", 1), ("int64", 1)], [1, 1], out_nd=2)
    vals = np.linspace(0.0, 1.0, 12)
    _, z = np_graph.run_graph(g, [gen(8), vals, np.array([2, 3])])
    assert z.shape == (2, 3) and len(set(z.ravel())) == 6 and set(z.ravel()) <= set(vals)
    with pytest.raises(ValueError, match="larger sample than population"):
        np_graph.run_graph(g, [gen(8), vals, np.array([13])])
    g = order_graph("choice_without_replacement", "int64", [("int64", 0), ("float64", 1), ("int64", 1)], [0, 1, 1])
    with pytest.raises(ValueError, match="Fewer non-zero entries in p than size"):
        np_graph.run_graph(g, [gen(8), np.asarray(3), np.array([0.5, 0.5, 0.0]), np.array([3])])


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
def test_device_multinomial(hip):
    from pytensor_amd.executor import HipExecutable

    g = rv_graph("multinomial", "int64", (3000,), [("int64", 0), ("float64", 1)])
    want = np_graph.run_graph(g, [gen(6), np.asarray(40), _MULTI_P])
    exe = HipExecutable(g, auto_freeze=True)
    for _ in range(2):
        got = exe(gen(6), np.asarray(40), _MULTI_P)
        assert np.array_equal(got[1], want[1]) and philox_ref.generator_state(got[0]) == philox_ref.generator_state(want[0])
    _, nxt = exe(got[0], np.asarray(40), _MULTI_P)
    check_multinomial(nxt, 40)
    # batched n and float32 p rows, size=None; large n takes the BTRS branch per category
    g = rv_graph("multinomial", "int64", None, [("int64", 1), ("float32", 2)])
    ins = [np.array([7, 900, 100000]), np.array([[0.5, 0.0, 0.5], [0.2, 0.3, 0.5], [0.25, 0.7, 0.05]], dtype="float32")]
    want = np_graph.run_graph(g, [gen(6), *ins])
    got = HipExecutable(g)(gen(6), *ins)
    assert got[1].shape == (3, 3) and np.array_equal(got[1], want[1])


@pytest.mark.gpu
def test_device_permutation_and_choice_without_replacement(hip):
    from pytensor_amd.executor import HipExecutable

    g = order_graph("permutation", "int64", [("int64", 0)], [0])
    want = np_graph.run_graph(g, [gen(5), np.asarray(5000)])
    exe = HipExecutable(g, auto_freeze=True)
    for _ in range(3):
        got = exe(gen(5), np.asarray(5000))
        assert np.array_equal(got[1], want[1]) and philox_ref.generator_state(got[0]) == philox_ref.generator_state(want[0])
    assert sorted(got[1]) == list(range(5000))
    g = order_graph("permutation", "float32", [("float32", 2)], [2])
    m = np.arange(3000, dtype="float32").reshape(1000, 3)
    want = np_graph.run_graph(g, [gen(5), m])
    got = HipExecutable(g)(gen(5), m)
    assert got[1].dtype == np.float32 and np.array_equal(got[1], want[1])
    pr = np.random.default_rng(0).dirichlet(np.ones(300))
    pr[::7] = 0.0
    pr /= pr.sum()
    g = order_graph("choice_without_replacement", "int64", [("int64", 0), ("float64", 1), ("int64", 1)], [0, 1, 1], out_nd=2)
    ins = [np.asarray(300), pr, np.array([5, 20])]
    want = np_graph.run_graph(g, [gen(9), *ins])
    got = HipExecutable(g)(gen(9), *ins)
    assert got[1].shape == (5, 20) and np.array_equal(got[1], want[1]) and (pr[got[1]] > 0).all()
    assert philox_ref.generator_state(got[0]) == philox_ref.generator_state(want[0])
    g = order_graph("choice_without_replacement", "float64", [("float64", 2), ("int64", 1)], [2, 1])
    a = np.arange(80.0).reshape(40, 2)
    want = np_graph.run_graph(g, [gen(9), a, np.array([7])])
    got = HipExecutable(g)(gen(9), a, np.array([7]))
    assert got[1].shape == (7, 2) and np.array_equal(got[1], want[1])
    with pytest.raises(ValueError, match="larger sample than population"):
        HipExecutable(g)(gen(9), a, np.array([41]))
    g = order_graph("choice_without_replacement", "int64", [("int64", 0), ("float64", 1), ("int64", 1)], [0, 1, 1])
    with pytest.raises(ValueError, match="Fewer non-zero entries in p than size"):
        HipExecutable(g)(gen(8), np.asarray(3), np.array([0.5, 0.5, 0.0]), np.array([3]))


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


def test_multivariate_normal_with_a_mean_per_draw():
    """MvNormalRV.rng_fn 915-916: without ``size`` the batch shape is the mean's leading shape"""
    _, mean, cov = _mv_cases()
    means = mean[None, :] + np.arange(3000)[:, None] * 0.0 + np.array([0.0, 10.0, -10.0])[np.arange(3000) % 3][:, None]
    g = rv_graph("multivariate_normal", "float64", None, [("float64", 2), ("float64", 2)])
    g2, x = np_graph.run_graph(g, [gen(4), means, cov])
    assert x.shape == (3000, 4) and philox_ref.generator_state(g2)[1] == 4 + 3000 * 4
    for k, off in enumerate((0.0, 10.0, -10.0)):
        assert np.abs(x[k::3].mean(axis=0) - (mean + off)).max() < 6 * np.sqrt(np.diag(cov).max() / 1000)
    assert np.abs(np.cov((x - means).T) - cov).max() < 0.12 * np.abs(cov).max()


@pytest.mark.gpu
def test_device_multivariate_normal_with_a_mean_per_draw(hip):
    from pytensor_amd.executor import HipExecutable

    _, mean, cov = _mv_cases()
    means = mean[None, :] + np.random.default_rng(1).normal(size=(257, 1))
    g = rv_graph("multivariate_normal", "float64", None, [("float64", 2), ("float64", 2)])
    want = np_graph.run_graph(g, [gen(4), means, cov])
    got = HipExecutable(g)(gen(4), means, cov)
    assert philox_ref.generator_state(got[0]) == philox_ref.generator_state(want[0])
    np.testing.assert_allclose(got[1], want[1], rtol=1e-11, atol=1e-11)
    # size given, a (1, k) mean broadcast against it
    g = rv_graph("multivariate_normal", "float64", (5, 7), [("float64", 2), ("float64", 2)])
    want = np_graph.run_graph(g, [gen(4), mean[None, :], cov])
    got = HipExecutable(g)(gen(4), mean[None, :], cov)
    assert got[1].shape == (5, 7, 4)
    np.testing.assert_allclose(got[1], want[1], rtol=1e-11, atol=1e-11)


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["svd", "eigh"])
def test_device_multivariate_normal_svd_and_eigh_factors(hip, method):
    """MvNormalRV(method=...) (random/basic.py:918-926): A = U sqrt(s) or V sqrt(w) instead of the Cholesky
    factor — the column signs are the decomposition's own, so the draws are pinned through their moments"""
    from pytensor_amd.executor import HipExecutable

    _, mean, cov = _mv_cases()
    g = rv_graph("multivariate_normal", "float64", (40000,), [("float64", 1), ("float64", 2)])
    g.nodes[0].params["method"] = method
    _, x = HipExecutable(g)(gen(4), mean, cov)
    assert x.shape == (40000, 4)
    assert np.abs(x.mean(axis=0) - mean).max() < 5 * np.sqrt(np.diag(cov).max() / 40000)
    assert np.abs(np.cov(x.T) - cov).max() < 0.05 * np.abs(cov).max()
    _, y = np_graph.run_graph(g, [gen(4), mean, cov])  # the restatement with NumPy's factor: same law
    assert np.abs(np.cov(y.T) - cov).max() < 0.05 * np.abs(cov).max()


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
