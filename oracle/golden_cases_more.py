"""More golden cases (imported by ``make_golden.py``): ops and variants the first batch did not
reach.  TEST INFRASTRUCTURE.  Shapes follow the reference's own tests where cited."""

from __future__ import annotations

import numpy as np
import pytensor
import pytensor.tensor as pt
from pytensor.tensor.linalg import cho_solve, cholesky

from make_golden import case


@case("dot_shapes")
def dot_shapes():
    # tests/tensor/test_math.py TestDot: every rank combination; integer dots have no BLAS
    # path and stay ``Dot`` (bit-exact tier)
    rng = np.random.default_rng(40)
    a, b, c = pt.dvector("a"), pt.dvector("b"), pt.dvector("c")
    A, B = pt.dmatrix("A"), pt.dmatrix("B")
    iA, iB = pt.lmatrix("iA"), pt.lmatrix("iB")
    iv = pt.lvector("iv")
    outs = [pt.dot(a, c), pt.dot(A, a), pt.dot(b, A), pt.dot(A, B.T), pt.dot(iA, iB), pt.dot(iA, iv), pt.dot(iv, iv),
            pt.dot(A.T, A) + pt.outer(a, a)]
    vals = {"a": rng.normal(size=7), "b": rng.normal(size=5), "c": rng.normal(size=7), "A": rng.normal(size=(5, 7)), "B": rng.normal(size=(9, 7)),
            "iA": rng.integers(-9, 9, size=(4, 6)), "iB": rng.integers(-9, 9, size=(6, 3)), "iv": rng.integers(-9, 9, size=6)}
    return [a, b, c, A, B, iA, iB, iv], outs, vals


@case("shape_ops")
def shape_ops():
    rng = np.random.default_rng(41)
    x = pt.dmatrix("x")
    s = pt.dscalar("s")
    k = pt.lscalar("k")
    outs = [
        x.shape, pt.shape(x)[0] * pt.shape(x)[1], pt.specify_shape(x, (None, 5)) * 2.0, pt.tensor_from_scalar(pt.scalar_from_tensor(s) * 2.0),
        x.copy(), pt.zeros_like(x) + s, pt.ones((k, 3), dtype="float32"), pt.full((2, k), s), pt.arange(k) * 2,
        pt.eye(4, 5, 1) + x[:4], pt.tril(x[:5, :5]), pt.repeat(x[0], 2), pt.tile(x[:2, :2], (2, 2)), pt.flatten(x, 1),
        pt.expand_dims(x, 1).sum(axis=1), pt.squeeze(x[:, :1], axis=1), pt.swapaxes(x, 0, 1) * 1.0, pt.stack([x[0], x[1]], axis=1),
    ]
    return [x, s, k], outs, {"x": rng.normal(size=(6, 5)), "s": np.asarray(1.25), "k": np.asarray(4)}


@case("indexing_more")
def indexing_more():
    # tests/tensor/test_subtensor.py: stepped / negative slices, take along axis 1, matrix rows
    # with duplicate indices (inc accumulates, set keeps the last writer)
    rng = np.random.default_rng(42)
    x = pt.dmatrix("x")
    y = pt.dmatrix("y")
    v = pt.dvector("v")
    idx = pt.lvector("idx")
    jdx = pt.lvector("jdx")
    t3 = pt.dtensor3("t3")
    outs = [
        x[::-1, ::2], x[-2:0:-1], x[1:-1:3, -1], t3[1, :, ::-2], t3[:, 2], x[:, jdx], pt.take(x, jdx, axis=1), x[idx][:, jdx],
        pt.inc_subtensor(x[idx], y[: idx.shape[0]]), pt.set_subtensor(x[idx], y[: idx.shape[0]]),
        pt.inc_subtensor(x[1:6:2, ::3], 2.5), pt.set_subtensor(x[::-2], x[:4] * 0 + v[:7]), pt.inc_subtensor(t3[0, 1:3], 1.0),
        x[idx, jdx[: idx.shape[0]]], pt.set_subtensor(v[-3:], 0.0), pt.inc_subtensor(v[::4], v[:3]),
    ]
    return [x, y, v, idx, jdx, t3], outs, {
        "x": rng.normal(size=(8, 7)), "y": rng.normal(size=(6, 7)), "v": rng.normal(size=9),
        "idx": np.array([5, 0, 5, 2, 7]), "jdx": np.array([6, 1, 1, 0, 3, 6]), "t3": rng.normal(size=(3, 4, 5)),
    }


@case("scan_variants")
def scan_variants():
    # tests/scan/test_basic.py: map (nit-sot only), two recurrent states with a shared
    # non-sequence, n_steps shorter than the sequence, reversed sequence
    rng = np.random.default_rng(43)
    xs = pt.dmatrix("xs")
    ys = pt.dmatrix("ys")
    a0 = pt.dvector("a0")
    b0 = pt.dvector("b0")
    W = pt.dmatrix("W")
    sq = pytensor.scan(lambda x, y: x * y + 1.0, sequences=[xs, ys], return_updates=False)

    def step(x, a, b, W):
        a_new = pt.tanh(a @ W + x)
        b_new = b * 0.5 + a_new
        return a_new, b_new

    aa, bb = pytensor.scan(step, sequences=[xs], outputs_info=[a0, b0], non_sequences=[W], n_steps=5, return_updates=False)
    rr = pytensor.scan(lambda x, acc: acc + x, sequences=[xs], outputs_info=[pt.zeros_like(a0)], go_backwards=True, return_updates=False)
    return [xs, ys, a0, b0, W], [sq, aa, bb[-1], rr], {
        "xs": rng.normal(size=(7, 4)), "ys": rng.normal(size=(7, 4)), "a0": rng.normal(size=4), "b0": rng.normal(size=4),
        "W": rng.normal(size=(4, 4)) * 0.5,
    }


@case("blockwise_linalg")
def blockwise_linalg():
    # tests/tensor/test_blockwise.py: batched cho_solve and a broadcast (one matrix, many rhs) solve
    rng = np.random.default_rng(44)
    S = pt.dtensor3("S")
    b = pt.dmatrix("b")
    B = pt.dtensor3("B")
    L = cholesky(S)
    outs = [cho_solve((L, True), b, b_ndim=1), cho_solve((L, True), B), pt.linalg.solve_triangular(L[0], B, lower=True, b_ndim=2)]
    A = rng.normal(size=(3, 6, 9))
    Sv = A @ A.transpose(0, 2, 1) / 9 + np.eye(6)
    return [S, b, B], outs, {"S": Sv, "b": rng.normal(size=(3, 6)), "B": rng.normal(size=(3, 6, 2))}


@case("hier_nodesign")
def hier_nodesign():
    # the C4 model without a design matrix — the gather a[gidx] feeds the likelihood
    # Composite directly (fused_elemwise.py:107 FusedElemwise is the reference's numba-only rewrite)
    rng = np.random.default_rng(46)
    N, G = 4001, 37
    gv = rng.integers(0, G, size=N)
    yv = rng.normal(size=N) + 0.3 * gv / G
    y = pytensor.shared(yv, name="y")
    gidx = pytensor.shared(gv, name="gidx")
    mu_g, log_tau, log_sigma = pt.dscalar("mu_g"), pt.dscalar("log_tau"), pt.dscalar("log_sigma")
    z = pt.dvector("z")
    a = mu_g + pt.exp(log_tau) * z
    r = (y - a[gidx]) / pt.exp(log_sigma)
    logp = pt.sum(-0.5 * r**2 - log_sigma) + pt.sum(-0.5 * z**2) - 0.5 * (mu_g**2 + log_tau**2 + log_sigma**2)
    params = [mu_g, log_tau, z, log_sigma]
    vals = {"mu_g": np.asarray(0.2), "log_tau": np.asarray(-0.4), "z": rng.normal(size=G), "log_sigma": np.asarray(0.1)}
    return params, [logp, *pytensor.grad(logp, params)], vals


@case("gather_elemwise")
def gather_elemwise():
    # several gathers (two tables, two index vectors, negative indices) into one Composite,
    # stored and reduced outputs
    rng = np.random.default_rng(47)
    t1, t2, v = pt.dvector("t1"), pt.dvector("t2"), pt.dvector("v")
    idx, jdx = pt.lvector("idx"), pt.lvector("jdx")
    e = pt.exp(t1[idx]) * v + t2[jdx] - t1[jdx]
    outs = [e, e.sum(), (t2[idx] * v).max(), pt.tanh(t1[idx] + 1.0)]
    n = 513
    return [t1, t2, v, idx, jdx], outs, {
        "t1": rng.normal(size=19), "t2": rng.normal(size=19), "v": rng.normal(size=n),
        "idx": rng.integers(-19, 19, size=n), "jdx": rng.integers(0, 19, size=n),
    }


@case("softmax_shapes", rtol=1e-11)
def softmax_shapes():
    # tests/tensor/test_special.py TestSoftmax/TestLogSoftmax axis matrix + the cross-entropy
    # gradient (Softmax.pullback, special.py:49-53); narrow rows (<= 16) and wide rows take
    # different kernels
    rng = np.random.default_rng(48)
    a = pt.dmatrix("a")
    b = pt.dmatrix("b")
    t3 = pt.dtensor3("t3")
    lab = pt.lvector("lab")
    sm = pt.special.softmax(a, axis=-1)
    lsm = pt.special.log_softmax(b, axis=1)
    nll = -lsm[pt.arange(b.shape[0]), lab].mean()
    outs = [sm, pt.special.softmax(a, axis=0), pt.special.log_softmax(a, axis=-1), lsm, pt.special.softmax(t3, axis=(0, 2)),
            pt.special.softmax(t3, axis=None), nll, pytensor.grad(nll, b), pytensor.grad((sm * sm).sum(), a)]
    bv = rng.normal(size=(29, 5)) * 3
    bv[3] = [700.0, -700.0, 0.0, 1.0, 2.0]  # overflow-safe thanks to the max subtraction
    return [a, b, t3, lab], outs, {"a": rng.normal(size=(37, 300)) * 4, "b": bv, "t3": rng.normal(size=(4, 6, 9)),
                                   "lab": rng.integers(0, 5, size=29)}


@case("softmax_f32", rtol=2e-5)
def softmax_f32():
    rng = np.random.default_rng(49)
    a = pt.fmatrix("a")
    return [a], [pt.special.softmax(a, axis=1), pt.special.log_softmax(a, axis=1), pt.special.softmax(a[:, :7], axis=1)], {
        "a": (rng.normal(size=(11, 1000)) * 5).astype("float32")}


@case("softmax_long_rows", rtol=1e-11)
def softmax_long_rows():
    # few, long rows take the workgroup-per-row kernel (cols >= 4096)
    rng = np.random.default_rng(50)
    a = pt.dmatrix("a")
    c = pt.dmatrix("c")
    return [a, c], [pt.special.softmax(a, axis=1), pt.special.log_softmax(a, axis=1), pt.special.softmax(c, axis=None)], {
        "a": rng.normal(size=(3, 5000)) * 3, "c": rng.normal(size=(70, 70))}


@case("general_solve_det", rtol=1e-10)
def general_solve_det():
    # tests/tensor/linalg/test_solvers/test_general.py, test_summary.py: Solve (gen / pos, vector
    # and matrix rhs), Det, SLogDet, MatrixInverse and the gradient of log|det| (a MatrixInverse)
    # (the solves and the determinants use different matrices: with a shared one the reference's
    #  C-mode pipeline computes det from the LU factors it introduces for the solves and returned
    #  +7.1e9 where np.linalg.det, its own NumPy mode and its unshared graph all give -7.1e9)
    rng = np.random.default_rng(51)
    A = pt.dmatrix("A")
    A2 = pt.dmatrix("A2")
    S = pt.dmatrix("S")
    b = pt.dvector("b")
    B = pt.dmatrix("B")
    T3 = pt.dtensor3("T3")
    U3 = pt.dtensor3("U3")
    sl = pt.linalg.slogdet(A)
    outs = [
        pt.linalg.solve(A2, b), pt.linalg.solve(A2.T, B), pt.linalg.solve(S, b, assume_a="pos"), pt.linalg.solve(S, B, assume_a="pos"),
        pt.linalg.det(A), sl[0], sl[1], pt.linalg.inv(A), pytensor.grad(sl[1], A), pt.linalg.det(T3), pt.linalg.solve(U3, B[:5]),
    ]
    n = 23
    Av = rng.normal(size=(n, n)) + np.eye(n) * 0.5
    M = rng.normal(size=(n, n + 4))
    return [A, A2, S, b, B, T3, U3], outs, {
        "A": Av, "A2": rng.normal(size=(n, n)) + np.eye(n), "S": M @ M.T / n + np.eye(n), "b": rng.normal(size=n),
        "B": rng.normal(size=(n, 3)), "T3": rng.normal(size=(4, 5, 5)) + np.eye(5), "U3": rng.normal(size=(4, 5, 5)) + 2 * np.eye(5)}


@case("scan_grad", rtol=1e-10)
def scan_grad():
    # the gradient of a Scan is a Scan with mit-mot states (Scan.pullback, op.py:2955+;
    # tests/scan/test_basic.py TestGradScan patterns): an RNN-like recurrence with a
    # non-sequence weight, and a two-tap recurrence
    rng = np.random.default_rng(52)
    xs = pt.dmatrix("xs")
    h0 = pt.dvector("h0")
    W = pt.dmatrix("W")
    f0 = pt.dmatrix("f0")

    def step(x, h, W):
        return pt.tanh(h @ W + x)

    hs = pytensor.scan(step, sequences=[xs], outputs_info=[h0], non_sequences=[W], return_updates=False)
    loss = (hs[-1] ** 2).sum() + 0.1 * hs.sum()
    fs = pytensor.scan(lambda fm2, fm1: fm1 * 0.6 + pt.sin(fm2), outputs_info=[dict(initial=f0, taps=[-2, -1])],
                       n_steps=6, return_updates=False)
    loss2 = (fs[-1] * fs[2]).sum()
    return [xs, h0, W, f0], [loss, *pytensor.grad(loss, [xs, h0, W]), loss2, pytensor.grad(loss2, f0)], {
        "xs": rng.normal(size=(8, 5)), "h0": rng.normal(size=5), "W": rng.normal(size=(5, 5)) * 0.4, "f0": rng.normal(size=(2, 3))}


@case("careduce_more")
def careduce_more():
    # elemwise.py:1233 CAReduce: every scalar op x axis pattern on a 4-d tensor, keepdims, mean/var
    rng = np.random.default_rng(45)
    t = pt.dtensor4("t")
    i3 = pt.ltensor3("i3")
    outs = [
        t.sum(axis=(0, 2)), t.prod(axis=3), t.max(axis=(1, 3)), t.min(axis=0), t.sum(axis=(1, 2, 3)), t.mean(axis=2),
        t.var(axis=(0, 1)), t.sum(axis=1, keepdims=True), t.max(), pt.argmax(t[0, 0], axis=1),
        i3.sum(axis=(0, 2)), i3.max(axis=1), i3.prod(axis=2), pt.all(i3 > -8, axis=0), pt.any(i3 > 7, axis=(1, 2)),
        pt.logsumexp(t, axis=(2, 3)), pt.cumsum(t[0, 0], axis=1),
    ]
    return [t, i3], outs, {"t": rng.normal(size=(3, 4, 5, 6)), "i3": rng.integers(-9, 9, size=(4, 3, 5))}


@case("scalar_incomplete_f64", rtol=1e-11, py_rtol=1e-6)
def scalar_incomplete_f64():
    # scalar/math.py: TriGamma 502, GammaInc 627, GammaIncC 674, BetaInc 1342 — their C support
    # code (c_code/gamma.c, c_code/incbet.c): series and continued-fraction branches, the
    # (half-)integer table of logGamma, the symmetry flip and the logarithmic fallback of incbet
    rng = np.random.default_rng(46)
    k = pt.dvector("k")  # shape parameters, some integers and half-integers
    x = pt.dvector("x")
    a = pt.dvector("a")
    b = pt.dvector("b")
    u = pt.dvector("u")  # (0, 1)
    outs = [pt.gammainc(k, x), pt.gammaincc(k, x), pt.gammainc(k * 30, x * 30), pt.gammaincc(k * 30, x * 35),
            pt.betainc(a, b, u), pt.betainc(a * 40, b * 55, u), pt.betainc(b, a, u * u), pt.tri_gamma(k), pt.tri_gamma(x * 1e-4 + 1e-6),
            pt.exp(-x) * pt.gammaincc(k, x) + pt.betainc(a, b, u) ** 2]
    n = 211
    kv = rng.uniform(0.1, 9, size=n)
    kv[::7] = np.round(kv[::7]) + 1.0
    kv[3::7] = np.round(kv[3::7]) + 0.5
    uv = rng.uniform(0.001, 0.999, size=n)
    uv[:4] = [0.0, 1.0, 0.96, 0.5]
    xv = rng.uniform(0.0, 14, size=n)
    xv[5] = 0.0
    vals = {"k": kv, "x": xv, "a": rng.uniform(0.2, 6, size=n), "b": rng.uniform(0.2, 6, size=n), "u": uv}
    return [k, x, a, b, u], outs, vals


@case("scalar_incomplete_f32", rtol=2e-6, py_rtol=1e-4)
def scalar_incomplete_f32():
    # float32 storage: the C code evaluates in double and casts (math.py:651-655)
    rng = np.random.default_rng(47)
    k = pt.fvector("k")
    x = pt.fvector("x")
    u = pt.fvector("u")
    outs = [pt.gammainc(k, x), pt.gammaincc(k, x), pt.betainc(k, k * 0.5 + 1.0, u), pt.tri_gamma(k)]
    n = 97
    vals = {"k": rng.uniform(0.2, 7, size=n).astype("float32"), "x": rng.uniform(0.01, 12, size=n).astype("float32"),
            "u": rng.uniform(0.01, 0.99, size=n).astype("float32")}
    return [k, x, u], outs, vals


@case("scalar_loop_grads", rtol=1e-9, py_rtol=1e-6)
def scalar_loop_grads():
    # scalar/loop.py:10 ScalarLoop inside Elemwise: the shape-parameter gradients of the incomplete
    # gamma / beta functions (math.py gammainc_grad 772+, gammaincc_grad, betainc_grad 1386+) are
    # while-loops over series / continued fractions, fused with the surrounding scalar ops
    rng = np.random.default_rng(48)
    k = pt.dvector("k")
    x = pt.dvector("x")
    a = pt.dvector("a")
    b = pt.dvector("b")
    u = pt.dvector("u")
    gk = pytensor.grad(pt.gammainc(k, x).sum(), [k, x])
    gkc = pytensor.grad((pt.gammaincc(k, x) * x).sum(), k)
    gab = pytensor.grad(pt.betainc(a, b, u).sum(), [a, b, u])
    n = 61
    vals = {"k": rng.uniform(0.3, 7, size=n), "x": rng.uniform(0.1, 12, size=n), "a": rng.uniform(0.4, 5, size=n),
            "b": rng.uniform(0.4, 5, size=n), "u": rng.uniform(0.02, 0.98, size=n)}
    return [k, x, a, b, u], [*gk, gkc, *gab], vals


@case("scalar_bessel", rtol=1e-10)
def scalar_bessel():
    # scalar/math.py: J1 1011, J0 1039 (libm through c_code), I1 1066, I0 1090 (SciPy, no C code)
    rng = np.random.default_rng(49)
    x = pt.dvector("x")
    f = pt.fvector("f")
    outs = [pt.j0(x), pt.j1(x), pt.i0(x), pt.i1(x), pt.j0(x * 4) + pt.i0(x * 0.25), pt.j1(f), pt.i0(f)]
    return [x, f], outs, {"x": rng.uniform(-7, 7, size=173), "f": rng.uniform(-5, 5, size=64).astype("float32")}


@case("eigh_symmetric", rtol=1e-9)
def eigh_symmetric():
    # linalg/decomposition/eigen.py:102 Eigh (standard problem) and its pullback 214-300.  An
    # eigenvector's sign is arbitrary (LAPACK's and the Jacobi kernel's differ), so every output
    # is a sign-free function: w, |v|, V f(w) V^T, and gradients of sign-free losses.
    rng = np.random.default_rng(50)
    A = pt.dmatrix("A")
    B = pt.dtensor3("B")
    w, v = pt.linalg.eigh(A)
    wu, vu = pt.linalg.eigh(A, lower=False)
    wb, vb = pt.linalg.eigh(B)
    loss = (pt.log(w) * w).sum() + ((v * v) * A).sum()
    outs = [w, pt.abs(v), (v * pt.exp(-w)[None, :]) @ v.T, wu, pt.abs(vu), wb, pt.abs(vb), pytensor.grad(loss, A)]
    n = 24
    M = rng.normal(size=(n, n))
    S = M @ M.T / n + np.diag(np.linspace(0.5, 3.0, n))
    T = rng.normal(size=(n, n))  # not symmetric: lower and upper triangles are different problems
    Bv = rng.normal(size=(3, 9, 9))
    Bv = Bv @ Bv.transpose(0, 2, 1) + np.eye(9) * np.arange(1, 10)
    return [A, B], outs, {"A": S + np.triu(T, 1), "B": Bv}


@case("eigh_f32", rtol=2e-5)
def eigh_f32():
    rng = np.random.default_rng(51)
    F = pt.fmatrix("F")
    wf, vf = pt.linalg.eigh(F)
    Fv = rng.normal(size=(17, 17)).astype("float32")
    Fv = (Fv + Fv.T) / 2
    return [F], [wf, (vf * wf[None, :]) @ vf.T, pt.abs(vf)], {"F": Fv}


@case("random_uniform_philox")
def random_uniform_philox():
    # RandomVariable (tensor/random/op.py) with a shared Generator(Philox): uniform draws in whole
    # 4-word blocks are the one case where the reference's numbers can be reproduced bit for bit
    # (Generator.uniform = low + (high - low) * random()); every other sampler of NumPy consumes a
    # data-dependent number of raw words and is pinned distributionally (tests/test_random.py).
    rng = pytensor.shared(np.random.Generator(np.random.Philox(key=[20260924, 7], counter=[5, 0, 0, 0])), name="rng")
    lo = pt.dvector("lo")
    s = pt.dscalar("s")
    r1, u1 = pt.random.uniform(-1.0, 2.0, size=(8, 4), rng=rng).owner.outputs
    r2, u2 = pt.random.uniform(lo, lo + s, size=(3, 4), rng=r1).owner.outputs
    r3, u3 = pt.random.uniform(0.0, 1.0, size=(16,), rng=r2).owner.outputs
    outs = [u1, u2, pt.exp(u1).sum(axis=0) + u2.mean(axis=0), u3.reshape((4, 4)) @ u1.T]
    return [lo, s], outs, {"lo": np.array([-3.0, 0.0, 10.0, 0.5]), "s": np.asarray(2.5)}


@case("bool_mask_split")
def bool_mask_split():
    # boolean-mask indexing (subtensor.py:1932, 2275: x[mask] == x[mask.nonzero()]), Nonzero
    # (basic.py), Split (basic.py:2203 — the gradient of Join / stack); index tier: bit-exact
    rng = np.random.default_rng(52)
    x = pt.dvector("x")
    y = pt.dvector("y")
    M = pt.dmatrix("M")
    iM = pt.lmatrix("iM")
    cat = pt.concatenate([x, y * 2.0, x[:3]])
    stk = pt.stack([x[:5], y[:5], x[2:7]])
    outs = [
        x[x > 0], M[M[:, 0] > 0], M[M > 0.5], M[:, M[0] < 0], (x[x > 0] ** 2).sum(),
        pt.set_subtensor(x[x > 0], 1.0), pt.inc_subtensor(M[M[:, 1] > 0], 10.0), pt.set_subtensor(M[M < 0], 0.0),
        pt.where(x > 0)[0], pt.nonzero(iM)[0], pt.nonzero(iM)[1], x[x > 100.0],
        *pytensor.grad((cat ** 2).sum() + (stk[0] * stk[1] * stk[2]).sum(), [x, y]),
        *pt.split(M, [2, 0, 5], n_splits=3, axis=1),
    ]
    vals = {"x": rng.normal(size=4500), "y": rng.normal(size=9), "M": rng.normal(size=(6, 7)), "iM": rng.integers(-1, 2, size=(5, 4))}
    return [x, y, M, iM], outs, vals


@case("sort_argsort")
def sort_argsort():
    # tensor/sort.py:31 SortOp, 156 ArgSortOp: every axis, NaNs last, ties (sorted values are
    # unaffected; argsort only on inputs without ties, where every algorithm agrees), a row longer
    # than one LDS tile (4096), integer and float32 keys, median through sort
    rng = np.random.default_rng(53)
    x = pt.dvector("x")
    M = pt.dmatrix("M")
    T3 = pt.ftensor3("T3")
    iv = pt.lvector("iv")
    big = pt.dvector("big")
    outs = [pt.sort(x), pt.argsort(x), pt.sort(M, axis=0), pt.sort(M, axis=1), pt.argsort(M, axis=0), pt.argsort(M, axis=-1),
            pt.sort(T3, axis=1), pt.argsort(T3, axis=2), pt.sort(iv), pt.sort(big), pt.argsort(big), pt.sort(pt.round(big))[::97],
            x[pt.argsort(x)][:5]]
    xv = rng.normal(size=301)
    xv[[5, 77]] = np.nan
    xv[100] = -np.inf
    xv[200] = np.inf
    bigv = rng.normal(size=10007) * 3
    vals = {"x": xv, "M": rng.normal(size=(37, 19)), "T3": rng.normal(size=(3, 9, 130)).astype("float32"),
            "iv": rng.integers(-5, 5, size=64), "big": bigv}
    return [x, M, T3, iv, big], outs, vals


@case("fill_long")
def fill_long():
    # ARange / Eye (tensor/basic.py) beyond the host-resident size: generated by a kernel.
    # Float ranges follow NumPy's fill loop (start + i * ((start + step) - start)).
    s = pt.dscalar("s")
    k = pt.lscalar("k")
    x = pt.dvector("x")
    outs = [pt.arange(0.25, s, 0.125) * 2.0, pt.arange(k) * 3, pt.arange(5, k, 7), pt.arange(0.1, s, 0.3, dtype="float32"),
            pt.eye(70, 90, 3) * s, pt.eye(k, k, -2, dtype="int32")[:5, :9], x[pt.arange(k)[::-1]][:4], pt.arange(10.0, -s, -0.7)]
    return [s, k, x], outs, {"s": np.asarray(60.0), "k": np.asarray(500), "x": np.random.default_rng(54).normal(size=600)}
