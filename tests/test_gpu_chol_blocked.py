"""GPU: Cholesky beyond one CU's LDS (n > 141 fp64 / 200 fp32) — the blocked multi-workgroup
factorisation (csrc/linalg.hip ``chol_blocked``: LDS-resident diagonal blocks, panel solves over
many workgroups, trailing updates on the MFMA GEMM) through the C-ABI ``pthip_potrf``.

Reference: ``Cholesky.perform`` (pytensor/tensor/linalg/decomposition/cholesky.py:48-83: LAPACK
``potrf``, ``clean=True`` zeroes the other triangle, ``info != 0`` => all-NaN).  Bound: the backward
error of a Cholesky factorisation is c·n·eps·|L||L^T| (Higham, Accuracy and Stability, Thm 10.3);
two correct factorisations of a well-conditioned matrix differ entry-wise by about cond·n·eps —
asserted as ``|L - L_lapack| <= C·n·eps·cond(S)·max|L|`` with C stated, and the residual
``|L L^T - S| <= C·n·eps·(|L||L^T|)`` entry-wise, which does not depend on the condition number."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from pytensor_amd import ffi

    if ffi.device_count() <= 0:
        pytest.fail("no HIP device visible: GPU tests must run on the MI355X box")
    ffi.init(0)
    return ffi


def _spd(n, dtype, seed):
    rng = np.random.default_rng(seed)
    A = rng.normal(size=(n, n + 5))
    return (A @ A.T / n + np.eye(n)).astype(dtype)


def _potrf(hip, S, lower):
    from pytensor_amd.device import DeviceArray

    n = S.shape[-1]
    batch = S.shape[0] if S.ndim == 3 else 1
    dS = DeviceArray.from_host(np.ascontiguousarray(S))
    L = DeviceArray.empty(S.shape, S.dtype)
    hip.check(hip.lib().pthip_potrf(hip.np_dtype_code(S.dtype), int(lower), batch, n, dS.ptr, L.ptr))
    return L.to_host()


@pytest.mark.parametrize("dtype,n", [("float64", 142), ("float64", 192), ("float64", 257), ("float64", 512), ("float64", 1000),
                                     ("float64", 2048), ("float32", 201), ("float32", 640), ("float32", 1500)])
@pytest.mark.parametrize("lower", [True, False])
def test_blocked_cholesky_matches_lapack(hip, dtype, n, lower):
    import scipy.linalg

    S = _spd(n, dtype, n)
    got = _potrf(hip, S, lower)
    want = scipy.linalg.cholesky(S, lower=lower)
    eps = np.finfo(dtype).eps
    # the other triangle is exactly zero (potrf clean=True)
    other = np.triu(got, 1) if lower else np.tril(got, -1)
    assert not other.any()
    L = got if lower else got.T
    Lw = want if lower else want.T
    cond = np.linalg.cond(S.astype("float64"))
    C = 4.0
    assert np.max(np.abs(L - Lw)) <= C * n * eps * cond * np.max(np.abs(Lw)), (np.max(np.abs(L - Lw)), cond)
    L64 = L.astype("float64")
    resid = np.abs(L64 @ L64.T - S.astype("float64"))
    bound = C * n * eps * (np.abs(L64) @ np.abs(L64).T)
    assert (resid <= bound).all(), float(np.max(resid / bound))
    # deterministic: a second call gives the same bits
    np.testing.assert_array_equal(got, _potrf(hip, S, lower))


@pytest.mark.parametrize("n,bad", [(300, 0), (300, 150), (300, 299), (777, 640), (512, 0), (512, 200), (512, 511)])
def test_blocked_cholesky_failure_is_all_nan(hip, n, bad):
    """A non-positive pivot in ANY diagonal block — first, middle, last panel — NaN-fills the whole
    result (cholesky.py:78-80), not just what was computed after it."""
    S = _spd(n, "float64", 7 * n)
    S[bad, bad] = -3.0
    for lower in (True, False):
        assert np.isnan(_potrf(hip, S, lower)).all()
    S[bad, bad] = np.nan
    assert np.isnan(_potrf(hip, S, True)).all()


@pytest.mark.parametrize("n", [260, 512])  # (512: whole tiles — the lower factor is formed straight in the output)
def test_blocked_cholesky_batch_and_upper_reads_only_its_triangle(hip, n):
    S = np.stack([_spd(n, "float64", 11), _spd(n, "float64", 12)])
    junk = S.copy()
    iu = np.triu_indices(n, 1)
    junk[0][iu] = 1e300  # lower factorisation must not read the strict upper triangle
    import scipy.linalg

    got = _potrf(hip, junk, True)
    for b in range(2):
        np.testing.assert_allclose(got[b], scipy.linalg.cholesky(S[b], lower=True), rtol=1e-10, atol=1e-12)
    junk = S.copy()
    il = np.tril_indices(n, -1)
    junk[1][il] = -1e300
    got = _potrf(hip, junk, False)
    for b in range(2):
        np.testing.assert_allclose(got[b], scipy.linalg.cholesky(S[b], lower=False), rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("n", [512, 2048])
def test_gp_marginal_likelihood_large_through_the_graph(hip, n):
    """An ordinary PyMC-shaped graph at sizes the LDS kernels cannot hold: the GP marginal
    log-likelihood and its gradient pieces — Cholesky(n), two triangular solves with a vector —
    as a lowered graph against the NumPy oracle."""
    import json
    import os

    import np_graph
    from pytensor_amd.executor import HipExecutable
    from pytensor_amd.ir import Graph

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = json.load(open(os.path.join(root, "tests", "golden", "gp_marginal_likelihood.json")))
    g = Graph.from_dict(d)
    z = np.load(os.path.join(root, "tests", "golden", "gp_marginal_likelihood.npz"))
    small = [z[f"in{k}"] for k in range(len(g.inputs))]
    # same graph, n points: inputs regenerated at that size with the fixture's own recipe
    rng = np.random.default_rng(n)
    ins = []
    for a in small:
        if a.ndim >= 1 and a.shape[0] == small[0].shape[0] and a.size > 1:
            ins.append(rng.normal(size=(n, *a.shape[1:])).astype(a.dtype))
        else:
            ins.append(a)
    want = np_graph.run_graph(g, ins)
    # the kernel matrix the graph factorises (input of its Cholesky node), from the oracle: every output is a
    # function of K^-1 y, K^-1 and log det K, whose forward errors scale with cond_2(K) (Higham, Thms 10.3/10.4:
    # |x - x^| <= c n eps cond(K) |x| for a Cholesky solve).  Bound: C eps cond(K) max|want| with C = 2, a
    # statistical (sqrt-free) constant — measured 0.18 (n = 512, cond 2.5e3) and 0.44 (n = 2048, cond 1.0e4) of
    # eps cond(K), profiles/r4_gp_bench.txt — floored at north_star's 1e-12
    import copy

    dd = copy.deepcopy(d)
    chol = next(nd for nd in dd["nodes"] if nd["op"] == "Cholesky")
    dd["outputs"] = [chol["inputs"][0]]
    ev = np.linalg.eigvalsh(np_graph.run_graph(Graph.from_dict(dd), ins)[0])
    cond = float(ev[-1] / ev[0])
    tol = max(1e-12, 2.0 * np.finfo("float64").eps * cond)
    exe = HipExecutable(g)
    got = exe(*ins)
    for k, (a, b) in enumerate(zip(got, want)):
        scale = max(1.0, float(np.max(np.abs(b))))
        err = float(np.max(np.abs(a - b)))
        assert err <= tol * scale, f"gp n={n} out{k}: err {err:.3e} > {tol:.3e} * {scale:.3e} (cond(K) = {cond:.3e})"
    # the shape asserts of the graph are host arithmetic (hostsplit.py): the graph freezes, replays are bit-identical
    plan = exe.freeze(*ins)
    try:
        for a, b in zip(plan(*ins), got):
            np.testing.assert_array_equal(a, b)
    finally:
        plan.close()


def test_persistent_kernels_replay_inside_a_captured_graph(hip):
    """Cholesky(1000) -> L^-1 b -> L^-T y as a lowered graph: the task-graph factorisation and the row-block
    solves (memsets of their flag / box arrays included) are captured into a hipGraph plan; replays
    reproduce the eager call bit for bit."""
    import scipy.linalg

    from pytensor_amd.executor import HipExecutable
    from pytensor_amd.ir import Graph

    g = Graph(name="chol_solves")
    S = g.new_var("float64", (None, None), name="S")
    b = g.new_var("float64", (None,), name="b")
    L = g.new_var("float64", (None, None))
    y = g.new_var("float64", (None,))
    Lt = g.new_var("float64", (None, None))
    x = g.new_var("float64", (None,))
    g.add_node("Cholesky", {"lower": True, "on_error": "nan"}, [S], [L])
    g.add_node("SolveTriangular", {"lower": True, "unit_diagonal": False, "b_ndim": 1}, [L, b], [y])
    g.add_node("DimShuffle", {"new_order": [1, 0]}, [L], [Lt])
    g.add_node("SolveTriangular", {"lower": False, "unit_diagonal": False, "b_ndim": 1}, [Lt, y], [x])
    g.inputs, g.outputs = [S, b], [L, x]
    n = 1000
    Sv = _spd(n, "float64", 3)
    bv = np.random.default_rng(4).normal(size=n)
    exe = HipExecutable(g)
    first = exe(Sv, bv)
    np.testing.assert_allclose(first[1], scipy.linalg.cho_solve((np.linalg.cholesky(Sv), True), bv), rtol=1e-9, atol=1e-12)
    plan = exe.freeze(Sv, bv)  # (raises if anything in the launch sequence cannot be captured)
    try:
        for _ in range(4):
            again = plan(Sv, bv)
            for a, c in zip(again, first):
                np.testing.assert_array_equal(a, c)
    finally:
        plan.close()


def test_launch_per_step_form_still_matches(hip):
    """``PTHIP_CHOL=steps`` (the launch-per-step form kept as the A/B reference of the task-graph
    kernel) is read once per process: checked in a child process against LAPACK."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import numpy as np, scipy.linalg, sys\n"
        f"sys.path.insert(0, {root!r})\n"
        "from pytensor_amd import ffi\n"
        "from pytensor_amd.device import DeviceArray\n"
        "ffi.init(0)\n"
        "for n in (300, 1000):\n"
        "    rng = np.random.default_rng(n); A = rng.normal(size=(n, n + 5)); S = A @ A.T / n + np.eye(n)\n"
        "    d = DeviceArray.from_host(S); L = DeviceArray.empty(S.shape, S.dtype)\n"
        "    ffi.check(ffi.lib().pthip_potrf(ffi.np_dtype_code(S.dtype), 1, 1, n, d.ptr, L.ptr))\n"
        "    np.testing.assert_allclose(L.to_host(), scipy.linalg.cholesky(S, lower=True), rtol=1e-10, atol=1e-12)\n"
        "print('ok')\n"
    )
    env = dict(os.environ, PTHIP_CHOL="steps")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]
