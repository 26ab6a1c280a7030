"""GPU, end to end: ``pytensor.function(..., mode="hip")`` against the reference's C linker
(``mode="CVM"``) in the same process, on the MI355X box.

This is the drop-in boundary itself: ``FunctionMaker`` → rewrites → ``HipLinker`` →
``JITLinker`` thunk (link/basic.py:670-684) → ``Function.__call__`` (compile/executor.py:651-744)
with its real storage cells, shared variables and update feedback.  Every function is called at
least three times (eager → hipGraph capture → replay).  Helper shaped after
tests/link/pytorch/test_basic.py:41-87 of the reference.
"""
import numpy as np
import pytest

import e2e_util as E
from e2e_util import assert_close, compare_hip_and_cvm, hip_executable

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pt():
    pytensor = E.activate()
    if not E.have_gpu():
        pytest.fail("no HIP device visible: GPU tests must run on the MI355X box")
    import pytensor.tensor as ptt

    return pytensor, ptt


def test_native_library_is_what_runs(pt):
    """The thunk of a ``mode="hip"`` function is a HipExecutable bound to libpthip.so."""
    pytensor, ptt = pt
    from pytensor_amd import ffi

    x = ptt.dvector("x")
    f = pytensor.function([x], ptt.exp(x).sum(), mode="hip")
    exe = hip_executable(f)
    assert ffi.lib()._name.endswith("pytensor_amd/libpthip.so")
    r = f(np.linspace(0, 1, 1000))
    assert isinstance(r, np.ndarray) and r.shape == () and r.dtype == np.float64
    assert exe.graph.nodes, "lowered graph is empty"


def test_c1_gauss_sum_and_grad(pt):
    pytensor, ptt = pt
    x, mu = ptt.dvector("x"), ptt.dscalar("mu")
    y = ptt.exp(-0.5 * (x - mu) ** 2).sum()
    rng = np.random.default_rng(1)
    # y is a sum of 1e5 positive terms: no cancellation, plain rtol
    compare_hip_and_cvm([x, mu], [y, pytensor.grad(y, x)], [rng.normal(size=100_000), 0.3], must_freeze=True)


def test_elemwise_broadcast_and_mixed_dtypes(pt):
    pytensor, ptt = pt
    a = ptt.tensor("a", shape=(None, 1, None), dtype="float64")
    b = ptt.tensor("b", shape=(1, None, None), dtype="float32")
    i = ptt.lvector("i")
    rng = np.random.default_rng(2)
    outs = [ptt.tanh(a) * b + ptt.sqrt(ptt.abs(a)), (a > b) & (b < 0.5), ptt.switch(a > 0, i, -i) // 3, ptt.cast(i, "int8") * 2,
            ptt.log1p(ptt.exp(b))]
    compare_hip_and_cvm([a, b, i], outs, [rng.normal(size=(7, 1, 5)), rng.normal(size=(1, 6, 5)).astype("float32"),
                                          rng.integers(-50, 50, size=5)])


def test_runtime_broadcast_error_is_a_valueerror(pt):
    pytensor, ptt = pt
    x, y = ptt.dmatrix("x"), ptt.dmatrix("y")
    f = pytensor.function([x, y], x + y, mode="hip")
    f(np.ones((3, 4)), np.ones((3, 4)))
    with pytest.raises(ValueError):  # elemwise.py:825-840
        f(np.ones((3, 4)), np.ones((1, 4)))
    with pytest.raises(TypeError):  # TensorType.filter (type.py:162): wrong ndim never reaches the thunk
        f(np.ones(3), np.ones((3, 4)))


@pytest.mark.parametrize("axis", [None, 0, 1, (0, 2), -1])
def test_careduce_axes(pt, axis):
    pytensor, ptt = pt
    x = ptt.dtensor3("x")
    xi = ptt.ltensor3("xi")
    rng = np.random.default_rng(3)
    xv = rng.normal(size=(9, 17, 33))
    # sums of N(0,1) values cancel: |error| is relative to sum|x_i|, not to |sum x_i|
    compare_hip_and_cvm([x], [x.sum(axis=axis)], [xv], atol=1e-12 * np.abs(xv).sum() / 10)
    compare_hip_and_cvm([x, xi], [x.max(axis=axis), x.min(axis=axis), ptt.abs(x).prod(axis=axis) if axis != None else x.max(),
                                  xi.sum(axis=axis), (xi > 0).all(axis=axis), (xi > 90).any(axis=axis)],
                        [xv, rng.integers(-100, 100, size=(9, 17, 33))])


def test_softmax_family(pt):
    pytensor, ptt = pt
    from pytensor.tensor.special import log_softmax, softmax

    x = ptt.dmatrix("x")
    rng = np.random.default_rng(4)
    xv = rng.normal(size=(64, 300)) * 3
    compare_hip_and_cvm([x], [softmax(x, axis=-1), log_softmax(x, axis=-1), ptt.logsumexp(x, axis=1), softmax(x, axis=0)], [xv],
                        atol=1e-15)  # log_softmax entries near 0 (max element of a peaked row)


def test_blas_family(pt):
    pytensor, ptt = pt
    A, B, Cm = ptt.dmatrix("A"), ptt.dmatrix("B"), ptt.dmatrix("C")
    v, w = ptt.dvector("v"), ptt.dvector("w")
    rng = np.random.default_rng(5)
    Av, Bv, Cv = rng.normal(size=(130, 70)), rng.normal(size=(70, 90)), rng.normal(size=(130, 90))
    vv, wv = rng.normal(size=70), rng.normal(size=130)
    outs = [ptt.dot(A, B), 0.5 * ptt.dot(A, B) + 2.0 * Cm, ptt.dot(A, v), wv.mean() * w + 3.0 * ptt.dot(A, v), ptt.dot(w, A),
            ptt.dot(A.T, Cm), ptt.outer(w, v) + A]
    # dot products of N(0,1) vectors cancel: bound by K * eps * |a||b| ~ 70 * 1e-16 * 10
    f, _ = compare_hip_and_cvm([A, B, Cm, v, w], outs, [Av, Bv, Cv, vv, wv], atol=2e-13)
    ops = [n.op for n in hip_executable(f).source_graph.nodes]
    assert any(o in ops for o in ("Dot22", "Gemm")) and "Gemv" in ops, ops


def test_blas_mixed_dtypes(pt):
    """Dot.make_node upcasts mixed operands (dispatch/blas.py): f32 @ f64, int64 @ f64."""
    pytensor, ptt = pt
    X32, Xi = ptt.fmatrix("X32"), ptt.lmatrix("Xi")
    w = ptt.dvector("w")
    W = ptt.dmatrix("W")
    rng = np.random.default_rng(6)
    compare_hip_and_cvm([X32, Xi, w, W], [ptt.dot(X32, w), ptt.dot(Xi, w), ptt.dot(X32, W), ptt.dot(Xi, W)],
                        [rng.normal(size=(40, 30)).astype("float32"), rng.integers(-5, 5, size=(40, 30)), rng.normal(size=30),
                         rng.normal(size=(30, 20))], atol=1e-13)


def test_batched_dot_f32(pt):
    pytensor, ptt = pt
    X, Y = ptt.ftensor3("X"), ptt.ftensor3("Y")
    rng = np.random.default_rng(7)
    f, _ = compare_hip_and_cvm([X, Y], [ptt.matmul(X, Y)], [rng.normal(size=(6, 40, 24)).astype("float32"),
                                                          rng.normal(size=(6, 24, 56)).astype("float32")], atol=2e-5)
    assert "BatchedDot" in [n.op for n in hip_executable(f).source_graph.nodes]


def test_cholesky_solves_and_nan_on_failure(pt):
    pytensor, ptt = pt
    S, b = ptt.dmatrix("S"), ptt.dvector("b")
    L = ptt.linalg.cholesky(S)
    rng = np.random.default_rng(8)
    A = rng.normal(size=(48, 48))
    Sv = A @ A.T + 48 * np.eye(48)
    outs = [L, ptt.linalg.solve_triangular(L, b, lower=True), ptt.linalg.solve(S, b, assume_a="pos"), ptt.log(ptt.diagonal(L)).sum()]
    f, _ = compare_hip_and_cvm([S, b], outs, [Sv, rng.normal(size=48)], rtol=1e-11, atol=1e-13)
    Sv[20, 20] = -1.0  # not positive definite: NaN, not an exception (cholesky.py:78-80)
    r = f(Sv, np.ones(48))
    assert np.isnan(r[0]).all()


def _hier(ptt, pytensor, N, K, G, seed=0):
    rng = np.random.default_rng(seed)
    y = pytensor.shared(rng.normal(size=N), name="y")
    X = pytensor.shared(rng.normal(size=(N, K)), name="X")
    gidx = pytensor.shared(rng.integers(0, G, size=N), name="gidx")
    A = rng.normal(size=(K, K))
    Sigma = pytensor.shared(A @ A.T + K * np.eye(K), name="Sigma")
    mu_g, log_tau, log_sigma = ptt.dscalar("mu_g"), ptt.dscalar("log_tau"), ptt.dscalar("log_sigma")
    z, beta = ptt.dvector("z"), ptt.dvector("beta")
    tau, sigma = ptt.exp(log_tau), ptt.exp(log_sigma)
    a = mu_g + tau * z
    L = ptt.linalg.cholesky(Sigma)
    alpha = ptt.linalg.solve_triangular(L, beta, lower=True)
    logp_beta = -0.5 * (alpha**2).sum() - ptt.log(ptt.diagonal(L)).sum() - 0.5 * K * np.log(2 * np.pi)
    eta = a[gidx] + ptt.dot(X, beta)
    r = (y - eta) / sigma
    logp_y = (-0.5 * r**2 - log_sigma - 0.5 * np.log(2 * np.pi)).sum()
    logp_z = (-0.5 * z**2 - 0.5 * np.log(2 * np.pi)).sum()
    logp = logp_y + logp_z + logp_beta - 0.5 * (mu_g**2 + log_tau**2 + log_sigma**2)
    params = [mu_g, log_tau, z, beta, log_sigma]
    vals = [np.float64(0.1), np.float64(-0.2), rng.normal(size=G), 0.1 * rng.normal(size=K), np.float64(0.3)]
    return params, [logp, *pytensor.grad(logp, params)], vals, dict(y=y, X=X, gidx=gidx, Sigma=Sigma)


def test_c4_hierarchical_logp_grad_with_shared_data(pt):
    """BASELINE configs[3] through the real Function: data as shared
    variables (device resident), parameters per call; `set_value` between calls re-uploads."""
    pytensor, ptt = pt
    N, K, G = 20_000, 32, 16
    params, outs, vals, sh = _hier(ptt, pytensor, N, K, G)
    # logp ~ -1e5 is a sum of same-sign terms; the gradients are sums of N mixed-sign terms:
    # error relative to sum|terms| ~ N * 1 -> atol 1e-12 * N * 0.1
    f, r0 = compare_hip_and_cvm(params, outs, vals, calls=4, atol=1e-12 * N * 0.1, must_freeze=True)
    exe = hip_executable(f)
    assert exe.resident, "shared variables should be device resident"
    # new data through set_value (compile/sharedvalue.py:97-130): both backends see it
    rng = np.random.default_rng(99)
    f_ref = pytensor.function(params, outs, mode=E.reference_mode())
    sh["y"].set_value(rng.normal(size=N))
    want = f_ref(*vals)
    for c in range(3):
        got = f(*vals)
        for k, (a, b) in enumerate(zip(got, want)):
            assert_close(a, b, f"after set_value: output {k}, call {c}", atol=1e-12 * N * 0.1)
    assert abs(float(got[0]) - float(r0[0])) > 1.0, "set_value had no effect"
    # other parameter values, same signature: replay path
    vals2 = [v * 1.5 for v in vals]
    want = f_ref(*vals2)
    for k, (a, b) in enumerate(zip(f(*vals2), want)):
        assert_close(a, b, f"other parameters: output {k}", atol=1e-12 * N * 0.1)
    assert exe._auto_plan is not None, "the call after a parameter change must stay on the replay path"


def test_borrowed_shared_value_mutated_in_place(pt):
    """`get_value(borrow=True)[...] = v` and `set_value(x, borrow=True)` + in-place edits
    (compile/sharedvalue.py:97-130): the reference backends read the storage cell's memory on
    every call, so the new contents must be seen (VERDICT r1 missing-5)."""
    pytensor, ptt = pt
    rng = np.random.default_rng(10)
    d = pytensor.shared(rng.normal(size=50_000), name="d", borrow=True)
    s = ptt.dscalar("s")
    out = [(d * s).sum(), ptt.exp(-d * d).sum()]
    f = pytensor.function([s], out, mode="hip")
    f_ref = pytensor.function([s], out, mode=E.reference_mode())
    for step in range(3):
        for c in range(3):
            for k, (a, b) in enumerate(zip(f(1.5), f_ref(1.5))):
                assert_close(a, b, f"step {step} call {c} output {k}", atol=1e-10)
        buf = d.get_value(borrow=True)
        buf[...] = rng.normal(size=buf.shape)  # bulk overwrite in place
    # an array handed over with borrow=True and edited afterwards
    mine = rng.normal(size=50_000)
    d.set_value(mine, borrow=True)
    f(1.5)
    f(1.5)
    mine *= 2.0
    for k, (a, b) in enumerate(zip(f(1.5), f_ref(1.5))):
        assert_close(a, b, f"borrowed set_value then in-place edit: output {k}", atol=1e-10)


def test_single_element_edit_of_a_large_borrowed_value_is_seen(pt):
    """The edit round 2's sampled fingerprint missed (VERDICT r2 missing-4 / ADVICE r2 medium): ONE
    element of a 4 MB borrowed array, between the sample points, through a view created before the
    upload, from another thread, and through ``set_value(x, borrow=True); x[i] += d`` — on the eager
    call, on the capturing call and on replays.  Default mode (``config.hip__resident == "guard"``:
    write-protected pages, csrc/guard.hip)."""
    import threading

    pytensor, ptt = pt
    assert pytensor.config.hip__resident == "guard"
    rng = np.random.default_rng(110)
    n = 500_000
    d = pytensor.shared(rng.normal(size=n), name="d", borrow=True)
    s = ptt.dscalar("s")
    out = [(d * s).sum(), d[123_457] * s, d.max()]
    f = pytensor.function([s], out, mode="hip")
    f_ref = pytensor.function([s], out, mode=E.reference_mode())

    def check(what):
        for k, (a, b) in enumerate(zip(f(1.5), f_ref(1.5))):
            assert_close(a, b, f"{what}: output {k}", atol=1e-9)

    buf = d.get_value(borrow=True)
    old_view = buf[100_000:200_000]
    for c in range(4):  # eager, capture, replay, replay
        check(f"call {c}")
        buf[123_457] += 1000.0 + c  # one element, far from every sample point of round 2
    check("after the last edit")
    exe = E.hip_executable(f)
    assert exe._auto_plan is not None, "the edits must not have pushed the function off the replay path"
    uploads = exe.stats["resident_uploads"]
    check("clean call")
    check("clean call")
    assert exe.stats["resident_uploads"] == uploads, "a clean array must not be uploaded again"
    old_view[23_457] = -7.0e6  # the same memory through an older view object
    check("edit through a pre-existing view")
    t = threading.Thread(target=lambda: buf.__setitem__(n - 1, 9.0e6))  # last element: the hashed edge page
    t.start()
    t.join()
    check("edit of the last element from another thread")
    buf[0] = -9.0e6  # first element: the other edge
    check("edit of the first element")
    mine = rng.normal(size=n)
    d.set_value(mine, borrow=True)
    check("set_value(borrow=True)")
    mine[77_777] += 5.0e5
    check("borrowed set_value then a single-element edit")
    assert exe.stats["resident_uploads"] > uploads


def test_two_functions_over_one_large_shared_value_upload_it_once_each(pt):
    """Two compiled functions over the SAME large shared value (a predict and a loss function over one
    weight vector): each keeps its own HBM copy and its own write-protection slot on the one host array.
    The second function's upload READS pages the first already protects — through pinned bounce buffers
    (csrc/runtime.hip pthip_h2d), not by lifting the protection, so it does not mark the first one's copy
    stale: alternating calls upload nothing (they used to re-upload the whole array every call).
    A real store is still seen by both."""
    pytensor, ptt = pt
    assert pytensor.config.hip__resident == "guard"
    rng = np.random.default_rng(5)
    w = pytensor.shared(rng.normal(size=1 << 20), name="w", borrow=True)  # 8 MiB: guarded, not hashed
    x = ptt.dscalar("x")
    f = pytensor.function([x], (w * x).sum(), mode="hip")
    g = pytensor.function([x], (w + x).max(), mode="hip")
    exf, exg = E.hip_executable(f), E.hip_executable(g)
    vals = w.get_value(borrow=True)
    for _ in range(2):
        np.testing.assert_allclose(f(2.0), 2.0 * vals.sum(), rtol=1e-11)
        np.testing.assert_allclose(g(1.0), vals.max() + 1.0, rtol=1e-12)
    assert exf.stats["resident_uploads"] == 1 and exg.stats["resident_uploads"] == 1, (exf.stats, exg.stats)
    for _ in range(6):
        f(2.0)
        g(1.0)
    assert exf.stats["resident_uploads"] == 1 and exg.stats["resident_uploads"] == 1, (exf.stats, exg.stats)
    vals[12345] = 1e6  # an in-place edit through the borrowed array: both copies are stale now
    np.testing.assert_allclose(g(1.0), 1e6 + 1.0, rtol=1e-12)
    np.testing.assert_allclose(f(2.0), 2.0 * vals.sum(), rtol=1e-11)
    assert exf.stats["resident_uploads"] == 2 and exg.stats["resident_uploads"] == 2, (exf.stats, exg.stats)


def test_resident_mode_flag_sampled_is_opt_in(pt):
    """``hip__resident`` is a real config flag (configdefaults.py:183 ``config.add``): the round-2
    behaviour is reachable, and only, through it."""
    pytensor, ptt = pt
    from pytensor_amd import coherence

    with pytensor.config.change_flags(hip__resident="sampled"):
        assert coherence.mode() == "sampled"
        d = pytensor.shared(np.zeros(100_000), name="d", borrow=True)
        f = pytensor.function([], d.sum(), mode="hip")
        f()
        tok = E.hip_executable(f)._resident_cache[0].fp
        assert type(tok).__name__ == "_Sample"
    assert coherence.mode() == "guard"


def test_update_output_aliasing_the_updated_shared_input(pt):
    """``function([], w, updates={w: w + 1})`` returns the OLD w (``insert_deepcopy`` does not copy
    an output that aliases an updated input, compile/aliasing.py:165-260): the device feedback
    must not overwrite the resident before the output has been read."""
    pytensor, ptt = pt

    def build(mode):
        w = pytensor.shared(np.arange(6.0), name="w")
        return pytensor.function([], [w, w[1:4], w.dimshuffle("x", 0)], updates={w: w + 1}, mode=mode), w

    (f_hip, w_hip), (f_ref, w_ref) = build("hip"), build(E.reference_mode())
    for c in range(5):
        for k, (a, b) in enumerate(zip(f_hip(), f_ref())):
            assert_close(a, b, f"call {c} output {k}")
        assert_close(w_hip.get_value(), w_ref.get_value(), f"w after call {c}")


@pytest.mark.parametrize("order", ["prev_first", "w_first"])
def test_chained_updates_read_the_old_values(pt, order):
    """``updates={w_prev: w, w: f(w)}``: every update expression sees the values from BEFORE the
    call, whatever the dictionary order (compile/executor.py:712-716 stores them after the call)."""
    pytensor, ptt = pt
    rng = np.random.default_rng(112)
    w0, p0 = rng.normal(size=300), rng.normal(size=300)

    def build(mode):
        w, wp = pytensor.shared(w0.copy(), name="w"), pytensor.shared(p0.copy(), name="w_prev")
        lr = ptt.dscalar("lr")
        new_w = w - lr * (w - wp) + 0.25
        ups = [(wp, w), (w, new_w)] if order == "prev_first" else [(w, new_w), (wp, w)]
        return pytensor.function([lr], (w * wp).sum(), updates=ups, mode=mode), w, wp

    (f_hip, w_hip, p_hip), (f_ref, w_ref, p_ref) = build("hip"), build(E.reference_mode())
    for c in range(6):
        assert_close(f_hip(0.1), f_ref(0.1), f"call {c}", atol=1e-12)
        assert_close(w_hip.get_value(), w_ref.get_value(), f"w after call {c}")
        assert_close(p_hip.get_value(), p_ref.get_value(), f"w_prev after call {c}")
    exe = E.hip_executable(f_hip)
    assert exe._auto_plan is not None and exe.stats["capture_failures"] == 0, "the staged (two-phase) feed must be capturable"


def test_swap_updates(pt):
    """``updates={x: y, y: x}`` exchanges the two values on every call."""
    pytensor, ptt = pt

    def build(mode):
        x, y = pytensor.shared(np.arange(5.0), name="x"), pytensor.shared(-np.arange(5.0), name="y")
        return pytensor.function([], x - 2 * y, updates=[(x, y), (y, x)], mode=mode), x, y

    (f_hip, x_hip, y_hip), (f_ref, x_ref, y_ref) = build("hip"), build(E.reference_mode())
    for c in range(5):
        assert_close(f_hip(), f_ref(), f"call {c}")
        assert_close(x_hip.get_value(), x_ref.get_value(), f"x after call {c}")
        assert_close(y_hip.get_value(), y_ref.get_value(), f"y after call {c}")
    exe = E.hip_executable(f_hip)
    assert exe._auto_plan is not None and exe.stats["capture_failures"] == 0


def test_update_call_that_raises_commits_nothing(pt):
    """A call that raises (device-side IndexError) leaves the shared value as it was — on the eager
    path and on the replay path — and the next good call continues from it."""
    pytensor, ptt = pt
    t0 = np.arange(10.0)

    def build(mode):
        w = pytensor.shared(np.zeros(4), name="w")
        table = pytensor.shared(t0.copy(), name="table")
        i = ptt.lvector("i")
        return pytensor.function([i], w.sum(), updates={w: w + table[i]}, mode=mode), w

    (f_hip, w_hip), (f_ref, w_ref) = build("hip"), build(E.reference_mode())
    good, bad = np.array([1, 2, 3, 4]), np.array([1, 2, 3, 40])
    for c, idx in enumerate([good, bad, good, good, good, bad, good, bad, good]):
        if idx is bad:
            with pytest.raises(IndexError):
                f_hip(idx)
            with pytest.raises(IndexError):
                f_ref(idx)
        else:
            assert_close(f_hip(idx), f_ref(idx), f"call {c}")
        assert_close(w_hip.get_value(), w_ref.get_value(), f"w after call {c}")


def test_updates_sgd_loop_stays_on_device_and_replays(pt):
    """An SGD-style ``updates=`` function (compile/executor.py:712-728 update feedback): same
    trajectory as the reference C linker, the weight stays in HBM between calls and the calls
    after the second run on the captured plan."""
    pytensor, ptt = pt
    rng = np.random.default_rng(11)
    Xv, yv = rng.normal(size=(4096, 64)), rng.normal(size=4096)
    w0 = rng.normal(size=64) * 0.1

    def build(mode):
        X, y = pytensor.shared(Xv, name="X"), pytensor.shared(yv, name="y")
        w = pytensor.shared(w0.copy(), name="w")
        lr = ptt.dscalar("lr")
        loss = ((ptt.dot(X, w) - y) ** 2).mean()
        g = pytensor.grad(loss, w)
        return pytensor.function([lr], loss, updates={w: w - lr * g}, mode=mode), w

    f_hip, w_hip = build("hip")
    f_ref, w_ref = build(E.reference_mode())
    for step in range(8):
        a, b = f_hip(0.05), f_ref(0.05)
        assert_close(a, b, f"loss at step {step}", rtol=1e-11)
        assert_close(w_hip.get_value(), w_ref.get_value(), f"weights after step {step}", rtol=1e-10, atol=1e-13)
    exe = hip_executable(f_hip)
    assert exe._auto_plan is not None, "the update loop never reached the replay path"
    assert exe.stats["resident_uploads"] <= 4, exe.stats  # X, y, w once (+1 slack); not once per step
    # the user resets the weight: honoured, trajectory restarts identically
    w_hip.set_value(w0.copy())
    w_ref.set_value(w0.copy())
    for step in range(3):
        assert_close(f_hip(0.05), f_ref(0.05), f"loss after reset, step {step}", rtol=1e-11)
    assert_close(w_hip.get_value(), w_ref.get_value(), "weights after reset", rtol=1e-10, atol=1e-13)


def test_large_updated_shared_value_is_coherent_and_loud(pt):
    """An updated shared value of 160 kB: after the eager call its host mirror is an ordinary array
    (write-protected pages see an in-place edit); on the replay path it lives in a pinned result
    block, which must not be write-protected — it is handed out read-only, so the edit the
    reference would accept raises instead of going stale.  ``get_value()`` copies and
    ``set_value`` work as in the reference, and the trajectory matches the C linker throughout."""
    pytensor, ptt = pt
    rng = np.random.default_rng(113)
    n = 20_000
    w0, g0 = rng.normal(size=n), rng.normal(size=n)

    def build(mode):
        w = pytensor.shared(w0.copy(), name="w")
        g = pytensor.shared(g0.copy(), name="g")
        lr = ptt.dscalar("lr")
        return pytensor.function([lr], (w * g).sum(), updates={w: w - lr * g * ptt.tanh(w)}, mode=mode), w

    (f_hip, w_hip), (f_ref, w_ref) = build("hip"), build(E.reference_mode())

    def step(c, atol=1e-9):
        assert_close(f_hip(0.1), f_ref(0.1), f"call {c}", atol=atol)
        assert_close(w_hip.get_value(), w_ref.get_value(), f"w after call {c}")

    step(0)  # eager: the new value is an ordinary host array
    for w in (w_hip, w_ref):
        w.get_value(borrow=True)[17] += 3.0  # accepted by both, seen by the next call
    for c in range(1, 5):
        step(c)
    assert E.hip_executable(f_hip)._auto_plan is not None
    cur = w_hip.get_value(borrow=True)
    if not cur.flags.writeable:  # (pinned hand-out)
        with pytest.raises(ValueError, match="read-only"):
            cur[0] = 1.0
    copy = w_hip.get_value()
    copy[0] = 123.0  # a copy is the caller's
    step(5)
    new = rng.normal(size=n)
    w_hip.set_value(new.copy())
    w_ref.set_value(new.copy())
    for c in range(6, 9):
        step(c)


def test_function_without_outputs_only_updates(pt):
    pytensor, ptt = pt
    c = pytensor.shared(np.zeros(5), name="c")
    inc = ptt.dvector("inc")
    f = pytensor.function([inc], [], updates={c: c + inc}, mode="hip")
    for _ in range(4):
        assert f(np.arange(5.0)) == []
    np.testing.assert_array_equal(c.get_value(), 4 * np.arange(5.0))
    g = pytensor.function([inc], None, updates={c: c * 0 + inc}, mode="hip")
    g_ref = pytensor.function([inc], None, updates={c: c * 0 + inc}, mode=E.reference_mode())
    assert g(np.ones(5)) == g_ref(np.ones(5))
    np.testing.assert_array_equal(c.get_value(), np.ones(5))
    # no fgraph outputs at all: the JITLinker thunk asserts the callable returned None (link/basic.py:690-699)
    h = pytensor.function([inc], [], mode="hip", on_unused_input="ignore")
    assert h(np.ones(5)) == []


def test_scan_gru_small(pt):
    """BASELINE configs[4] shape at T=20, B=8, H=64 fp32."""
    pytensor, ptt = pt
    T, B, H = 20, 8, 64
    rng = np.random.default_rng(12)
    mats = [pytensor.shared((rng.normal(size=(H, H)) * 0.1).astype("float32"), name=f"W{k}") for k in range(6)]
    bias = [pytensor.shared((rng.normal(size=(1, H)) * 0.1).astype("float32"), name=f"b{k}", shape=(1, H)) for k in range(3)]
    Wz, Wr, Wh, Uz, Ur, Uh = mats
    bz, br, bh = bias
    xs, h0 = ptt.ftensor3("xs"), ptt.fmatrix("h0")

    def step(x, h):
        z = ptt.sigmoid(ptt.dot(x, Wz) + ptt.dot(h, Uz) + bz)
        r = ptt.sigmoid(ptt.dot(x, Wr) + ptt.dot(h, Ur) + br)
        hh = ptt.tanh(ptt.dot(x, Wh) + ptt.dot(r * h, Uh) + bh)
        return (1 - z) * h + z * hh

    hs = pytensor.scan(step, sequences=[xs], outputs_info=[h0], return_updates=False)
    f, _ = compare_hip_and_cvm([xs, h0], [hs[-1].sum(), hs], [rng.normal(size=(T, B, H)).astype("float32"),
                                                            np.zeros((B, H), dtype="float32")], rtol=2e-5, atol=2e-5)
    assert "Scan" in [n.op for n in hip_executable(f).source_graph.nodes]


def test_scan_cumulative_with_until_and_grad(pt):
    pytensor, ptt = pt
    xs, s0 = ptt.dmatrix("xs"), ptt.dvector("s0")
    out = pytensor.scan(lambda x, s: ptt.tanh(s + x), sequences=[xs], outputs_info=[s0], return_updates=False)
    cost = (out[-1] ** 2).sum()
    rng = np.random.default_rng(13)
    compare_hip_and_cvm([xs, s0], [cost, *pytensor.grad(cost, [xs, s0])], [rng.normal(size=(12, 5)), rng.normal(size=5)], rtol=1e-11,
                        atol=1e-14)


def test_scan_truncated_gradient_unreached_steps_are_zero(pt):
    """``truncate_gradient=k``: the backward Scan runs k steps into buffers of length T; "Scan is
    expected to return 0 for all entries for which the gradient is not actually computed"
    (scan/op.py:2280-2286, scan_perform.pyx:572-577).  Found by the reference's own
    ``TestScan::test_grad_multiple_outs_some_truncate`` under the hip linker: the device loop left
    those rows as the pool handed them out."""
    pytensor, ptt = pt
    rng = np.random.default_rng(114)
    T = 9
    u, x0, W = ptt.dmatrix("u"), ptt.dvector("x0"), ptt.dmatrix("W")

    def step(u_t, x_tm1, W):
        return ptt.tanh(ptt.dot(u_t, W) + x_tm1)

    xs = pytensor.scan(step, sequences=u, outputs_info=x0, non_sequences=W, truncate_gradient=3, return_updates=False)
    cost = (xs[-1] ** 2).sum()
    grads = pytensor.grad(cost, [u, x0, W])
    vals = [rng.normal(size=(T, 4)), rng.normal(size=4), 0.3 * rng.normal(size=(4, 4))]
    for _ in range(3):  # (garbage in recycled pool blocks differs from call to call)
        f, got = E.compare_hip_and_cvm([u, x0, W], grads, vals, atol=1e-12)
        assert not got[0][: T - 3].any(), "gradient rows of the steps beyond the truncation must be exactly zero"


def test_indexing_tier_is_bit_exact(pt):
    pytensor, ptt = pt
    x, M = ptt.dvector("x"), ptt.dmatrix("M")
    idx = ptt.lvector("idx")
    rng = np.random.default_rng(14)
    outs = [x[idx], M[idx], M[:, ::-2], M[1:-1, 3], ptt.set_subtensor(x[2:5], 7.0), ptt.inc_subtensor(ptt.zeros(11)[idx], x[idx]),
            ptt.concatenate([x, x[::-1]]), M.T.reshape((-1,))[:9], ptt.alloc(x[0], 3, 4), M.shape[0] * 2 + x.shape[0], ptt.argmax(M, axis=1),
            ptt.diagonal(M), ptt.cumsum(x), ptt.sort(x), ptt.argsort(x)]
    f, got = compare_hip_and_cvm([x, M, idx], outs, [rng.normal(size=11), rng.normal(size=(11, 13)), rng.integers(-11, 11, size=20)], rtol=0)
    with pytest.raises(IndexError):
        f(np.ones(11), np.ones((11, 13)), np.array([0, 11]))


def test_random_uniform_with_rng_update(pt):
    """A shared Generator advanced through ``updates=`` (tensor/random/op.py): three calls give
    the three blocks the reference's C linker gives (uniform is the bit-exact sampler)."""
    pytensor, ptt = pt

    def build(mode):
        rng = pytensor.shared(np.random.Generator(np.random.Philox(key=7)), name="rng")
        nr, u = ptt.random.uniform(-1.0, 2.0, size=(1000,), rng=rng).owner.outputs
        return pytensor.function([], u, updates={rng: nr}, mode=mode)

    f_hip, f_ref = build("hip"), build(E.reference_mode())
    for c in range(3):
        a, b = f_hip(), f_ref()
        np.testing.assert_array_equal(a, b, err_msg=f"draw {c}")


def test_function_copy_givens_and_trust_input(pt):
    pytensor, ptt = pt
    x = ptt.dvector("x")
    s = pytensor.shared(np.arange(4.0), name="s")
    f = pytensor.function([x], (x * s).sum(), mode="hip")
    assert float(f(np.ones(4))) == 6.0
    s2 = pytensor.shared(np.ones(4) * 2, name="s2")
    f2 = f.copy(swap={s: s2})  # compile/executor.py Function.copy: re-links with the same linker class
    assert float(f2(np.ones(4))) == 8.0 and float(f(np.ones(4))) == 6.0
    f.trust_input = True
    for _ in range(3):
        assert float(f(np.ones(4))) == 6.0
    y = ptt.dvector("y")
    g = pytensor.function([y], ptt.exp(x), givens={x: y * 0}, mode="hip")
    np.testing.assert_array_equal(g(np.ones(3)), np.ones(3))


def test_function_pickle_roundtrip_with_resident_data_and_frozen_plan(pt):
    """compile/executor.py:829-879 (``_pickle_Function`` / ``_constructor_Function``): a Function pickles as its maker +
    the values of its storage cells and re-links on load.  Under the hip linker that means: the lowered graph, the
    generated kernels and the device-resident copy of the shared data are all re-created lazily by the loaded
    function (SURVEY §5 "Linker must be copy/pickle-safe, device handles re-creatable lazily").  The original has a
    captured plan and resident buffers when it is pickled; the copy must not share any of it."""
    import copy
    import pickle

    pytensor, ptt = pt
    rng = np.random.default_rng(77)
    N, K = 50_000, 16
    X = pytensor.shared(rng.normal(size=(N, K)), name="X")  # resident: 6.4 MB (page-guarded coherence)
    y = pytensor.shared(rng.normal(size=N), name="y")
    step = pytensor.shared(np.zeros(()), name="step")
    beta, ls = ptt.dvector("beta"), ptt.dscalar("ls")
    r = (y - X @ beta) * ptt.exp(-ls)
    logp = (-0.5 * r**2 - ls).sum() - 0.5 * (beta**2).sum()
    f = pytensor.function([beta, ls], [logp, *pytensor.grad(logp, [beta, ls])], updates={step: step + 1.0}, mode="hip")
    bv, lv = rng.normal(size=K) * 0.1, np.asarray(0.2)
    want = [np.array(a) for a in f(bv, lv)]
    for _ in range(3):
        f(bv, lv)  # capture + replays
    exe = hip_executable(f)
    assert exe.stats["replays"] >= 1
    blob = pickle.dumps(f)
    assert len(blob) > X.get_value(borrow=True).nbytes  # the shared data travel with the function (as in the reference)
    g = pickle.loads(blob)
    exe_g = hip_executable(g)
    assert exe_g is not exe and exe_g.stats["replays"] == 0
    assert float(g.get_shared()[[v.name for v in g.get_shared()].index("step")].get_value()) == 4.0
    for call in range(4):  # eager -> capture -> replay, on the loaded function's own buffers
        got = g(bv, lv)
        for a, b in zip(got, want):
            np.testing.assert_array_equal(np.asarray(a), b)
    assert exe_g.stats["replays"] >= 1
    # independent state: the copy's update did not touch the original's shared variable, and vice versa
    assert float(step.get_value()) == 4.0
    gs = {v.name: v for v in g.get_shared()}
    assert float(gs["step"].get_value()) == 8.0
    # an in-place edit of the ORIGINAL's data is seen by the original only
    Xv = X.get_value(borrow=True)
    Xv[17, 3] += 1.0
    changed = f(bv, lv)
    assert not np.array_equal(np.asarray(changed[0]), want[0])
    np.testing.assert_array_equal(np.asarray(g(bv, lv)[0]), want[0])
    Xv[17, 3] -= 1.0
    np.testing.assert_array_equal(np.asarray(f(bv, lv)[0]), want[0])
    # copy.copy / copy.deepcopy of a Function go through Function.copy / __deepcopy__ (compile/executor.py:760-827)
    h = copy.copy(f)
    np.testing.assert_array_equal(np.asarray(h(bv, lv)[0]), want[0])
    # pickling a function whose plan was never captured, and a second round trip of the loaded one
    g2 = pickle.loads(pickle.dumps(g))
    np.testing.assert_array_equal(np.asarray(g2(bv, lv)[0]), want[0])


def test_scan_function_pickle_roundtrip(pt):
    """the reference's own pickling test is a Scan (tests/scan/test_basic.py:310, re-enabled in
    test_gpu_refsuite_scan.py); here with a sequence, a shared weight and a gradient through the loop"""
    import pickle

    pytensor, ptt = pt
    from pytensor.scan.basic import scan

    rng = np.random.default_rng(78)
    W = pytensor.shared(rng.normal(size=(6, 6)) * 0.3, name="W")
    xs, h0 = ptt.dmatrix("xs"), ptt.dvector("h0")
    hs = scan(lambda x, h: ptt.tanh(x + h @ W), sequences=[xs], outputs_info=[h0], return_updates=False)
    loss = (hs[-1] ** 2).sum()
    f = pytensor.function([xs, h0], [loss, pytensor.grad(loss, W)], mode="hip")
    args = [rng.normal(size=(9, 6)), rng.normal(size=6)]
    want = [np.array(a) for a in f(*args)]
    g = pickle.loads(pickle.dumps(f))
    for _ in range(3):
        for a, b in zip(g(*args), want):
            np.testing.assert_array_equal(np.asarray(a), b)


def test_scan_trace_read_at_the_end_is_shortened(pt):
    """a recurrence whose trajectory is read at [-1] only keeps taps + 1 rows, not n_steps + taps (the C backend's
    `scan_reduce_trace_prealloc`, scan/rewriting/trace.py:888; pytensor_amd/linker.py says why not the JIT variant) — and the
    value, the short-run value and the gradient are those of the full-trace evaluation.  Modelled on the reference's
    tests/scan/rewriting/test_trace.py::TestReduceTrace, which asserts the JIT backends' `taps` and is not run."""
    pytensor, ptt = pt
    from pytensor.scan.basic import scan
    from pytensor.scan.op import Scan

    n, x0, a = ptt.lscalar("n"), ptt.dvector("x0"), ptt.dscalar("a")
    ys = scan(lambda xm2, xm1, a_: a_ * xm1 + 0.5 * xm2, outputs_info=[{"initial": x0, "taps": [-2, -1]}], non_sequences=[a], n_steps=n, return_updates=False)
    f1 = pytensor.function([n, x0, a], ys[-1], mode="hip")
    [node] = [nd for nd in f1.maker.fgraph.apply_nodes if isinstance(nd.op, Scan)]
    [buf] = [v for v in node.inputs[1:] if v.type.ndim == 1 and v.type.dtype == "float64"]
    rows = pytensor.function([n, x0, a], buf.shape[0], mode="hip", on_unused_input="ignore")
    assert int(rows(1000, [1.0, 1.0], 0.9)) == 3  # 2 taps + 1, not 1002
    # (with the gradient in the same function the backward Scan reads the whole forward trajectory: that one stays long)
    f = pytensor.function([n, x0, a], [ys[-1], pytensor.grad(ys[-1], a)], mode="hip")

    def ref(nsteps, av):
        p2, p1, d2, d1 = 1.0, 1.0, 0.0, 0.0
        for _ in range(nsteps):
            p2, p1, d2, d1 = p1, av * p1 + 0.5 * p2, d1, p1 + av * d1 + 0.5 * d2
        return p1, d1

    for nsteps in (1000, 1, 7):
        got = f(nsteps, [1.0, 1.0], 0.9)
        want = ref(nsteps, 0.9)
        np.testing.assert_allclose(np.asarray(f1(nsteps, [1.0, 1.0], 0.9)), want[0], rtol=1e-12)
        np.testing.assert_allclose(np.asarray(got[0]), want[0], rtol=1e-12)
        np.testing.assert_allclose(np.asarray(got[1]), want[1], rtol=1e-10)


def test_outputs_are_fresh_and_do_not_alias_inputs(pt):
    """link/vm.py:860-880 no_recycling semantics + aliasing.py:165-260 (DeepCopyOp)."""
    pytensor, ptt = pt
    x = ptt.dvector("x")
    f = pytensor.function([x], [x, x[::2], ptt.exp(x)], mode="hip")
    xv = np.arange(6.0)
    a1 = f(xv)
    a2 = f(xv)
    a3 = f(xv)
    for r in (a1, a2, a3):
        assert not np.shares_memory(r[0], xv)
    keep = [r.copy() for r in a1]
    a3[2][...] = -1.0  # scribbling over a later result must not change an earlier one
    a2[0][...] = -1.0
    for k, r in zip(keep, a1):
        np.testing.assert_array_equal(k, r)


def test_wide_model_many_fused_kernels(pt):
    """north_star's "~200 fused Elemwise" scale: 40 independent likelihood terms in one graph."""
    pytensor, ptt = pt
    rng = np.random.default_rng(15)
    N = 4000
    mu, ls = ptt.dvector("mu"), ptt.dvector("ls")
    terms = []
    for k in range(40):
        yk = pytensor.shared(rng.normal(size=N) + 0.1 * k, name=f"y{k}")
        r = (yk - mu[k]) * ptt.exp(-ls[k])
        fam = k % 4
        if fam == 0:
            terms.append((-0.5 * r**2 - ls[k]).sum())
        elif fam == 1:
            terms.append((-ptt.log1p(r**2 / 3.0) * 2.0 - ls[k]).sum())
        elif fam == 2:
            terms.append((-ptt.abs(r) - ls[k]).sum())
        else:
            terms.append((-r - 2.0 * ptt.softplus(-r) - ls[k]).sum())
    logp = ptt.add(*terms)
    compare_hip_and_cvm([mu, ls], [logp, *pytensor.grad(logp, [mu, ls])], [rng.normal(size=40) * 0.1, rng.normal(size=40) * 0.1],
                        atol=1e-12 * N, must_freeze=True)


def test_profile_flag_reports_device_time(pt):
    """``pytensor.function(profile=True)``: the hip thunk feeds ``vm_call_time`` like
    any VM and exposes per-node device time through ``update_profile``."""
    pytensor, ptt = pt
    x = ptt.dvector("x")
    f = pytensor.function([x], [ptt.tanh(x).sum(), ptt.exp(x)], mode="hip", profile=True)
    for _ in range(4):
        f(np.ones(10_000))
    assert f.profile.fct_callcount >= 3 and f.profile.vm_call_time > 0


def test_library_functions_end_to_end(pt):
    """einsum / pad / interpolate / the linalg helper functions / signal / fft, compiled by the real
    ``pytensor.function(mode="hip")`` and run on the MI355X next to the C linker.  atol: eps-level
    multiples of the largest entry for outputs that are differences of O(1) numbers (padding means,
    decompositions); the integer-valued ones are bit-exact."""
    pytensor, ptt = pt
    from pytensor.tensor.fft import irfft, rfft
    from pytensor.tensor.interpolate import interp
    from pytensor.tensor.signal import convolve1d

    x, y, v, t3 = ptt.dmatrix("x"), ptt.dmatrix("y"), ptt.dvector("v"), ptt.dtensor3("t3")
    L = ptt.linalg
    outs = [
        ptt.einsum("ij,jk->ik", x, y), ptt.einsum("bij,jk->bik", t3, y), ptt.einsum("ij,ij->", x, x),
        ptt.pad(x, ((1, 2), (0, 3)), mode="constant", constant_values=2.0), ptt.pad(x, 2, mode="edge"), ptt.pad(x, 1, mode="reflect"),
        ptt.pad(x, 1, mode="wrap"), ptt.pad(x, 1, mode="mean"), interp(v, ptt.as_tensor(np.linspace(0, 1, 7)), ptt.as_tensor(np.linspace(0, 1, 7) ** 2)),
        ptt.tril(x), ptt.diff(v), ptt.roll(x, 2, axis=1), ptt.median(v), ptt.logsumexp(x, axis=1), ptt.bincount(ptt.cast(ptt.abs(v) * 3, "int64")),
        ptt.tile(v, (2, 3)), ptt.tensordot(x, y, axes=[[1], [0]]), ptt.ptp(x, axis=0), ptt.searchsorted(ptt.sort(v), v), ptt.unique(ptt.round(v)),
        ptt.repeat(v, 2), L.kron(x[:2, :2], y[:2, :2]), L.norm(x), L.matrix_power(x, 3), L.pinv(y), L.svd(y)[1], L.qr(y)[1], L.lstsq(y, v[:5], -1.0)[0],
        L.slogdet(x)[1], L.block_diag(x, y), L.lu_solve(L.lu_factor(x), v[:5]), L.expm(x * 0.1), L.solve_continuous_lyapunov(x - 4 * ptt.eye(5), x @ x.T),
        convolve1d(v, v[:3], mode="full"), rfft(x), irfft(rfft(x), is_odd=True),
    ]
    rng = np.random.default_rng(0)
    vals = [rng.normal(size=(5, 5)), rng.normal(size=(5, 4)), rng.normal(size=6), rng.normal(size=(2, 3, 5))]
    f, _ = compare_hip_and_cvm([x, y, v, t3], outs, vals, rtol=1e-11, atol=2e-13, calls=2, on_unused_input="ignore")
    ops = [n.op for n in f.maker.linker.last_ir.nodes]
    assert "HostPerform" not in ops and {"SVD", "QR", "Expm", "SolveSylvester", "RFFTOp", "IRFFTOp", "Unique", "SearchsortedOp"} <= set(ops), sorted(set(ops))
