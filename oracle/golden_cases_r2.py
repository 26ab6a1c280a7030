"""Golden cases added in round 2 (imported by ``make_golden.py``).  TEST INFRASTRUCTURE.

What they pin: mixed-dtype ``Dot``, BLAS nodes whose alpha/beta are computed on the
device, the SGD/mean-squared-error graph that made ``fuse_gemv_chain`` produce a cycle, the
boundary dtypes ``TensorType`` carries beyond round 1 (pytensor/tensor/type.py:40-57:
uint16/32/64 on the bit-exact tier, float16 as a storage type) and raw ``CAReduce`` nodes that
accumulate wide and store narrow (tests/tensor/test_elemwise.py:444 ``TestCAReduce``).
"""

from __future__ import annotations

import numpy as np
import pytensor
import pytensor.scalar as ps
import pytensor.tensor as pt
from pytensor.tensor.elemwise import CAReduce

from make_golden import case


@case("dot_mixed_dtypes")
def dot_mixed_dtypes():
    # Dot.make_node upcasts mixed operands (tensor/math.py Dot; np.dot does the same)
    rng = np.random.default_rng(60)
    X32, Xi = pt.fmatrix("X32"), pt.lmatrix("Xi")
    w, W = pt.dvector("w"), pt.dmatrix("W")
    v32 = pt.fvector("v32")
    outs = [pt.dot(X32, w), pt.dot(Xi, w), pt.dot(X32, W), pt.dot(Xi, W), pt.dot(w, W), pt.dot(v32, W), pt.dot(v32, w)]
    vals = {"X32": rng.normal(size=(9, 11)).astype("float32"), "Xi": rng.integers(-5, 5, size=(9, 11)), "w": rng.normal(size=11),
            "W": rng.normal(size=(11, 6)), "v32": rng.normal(size=11).astype("float32")}
    return [X32, Xi, w, W, v32], outs, vals


@case("blas_device_alpha")
def blas_device_alpha():
    # alpha / beta that are functions of an *input* scalar (an SGD step size): the rewritten graph
    # holds Gemv / Gemm / Dot22Scalar / Ger nodes whose scalars live on the device
    rng = np.random.default_rng(61)
    A, B, Cm = pt.dmatrix("A"), pt.dmatrix("B"), pt.dmatrix("C")
    x, y = pt.dvector("x"), pt.dvector("y")
    lr, mom = pt.dscalar("lr"), pt.dscalar("mom")
    outs = [
        y - lr * pt.dot(A, x),  # Gemv(y, -lr, A, x, 1)
        mom * y + lr * pt.dot(A, x),  # Gemv with both scalars computed
        Cm - lr * pt.dot(A, B),  # Gemm(C, -lr, A, B, 1)
        mom * Cm + (lr * 2.0) * pt.dot(A, B),
        lr * pt.dot(A, B),  # Dot22Scalar
        A + lr * pt.outer(y, x),  # Ger
    ]
    vals = {"A": rng.normal(size=(13, 9)), "B": rng.normal(size=(9, 7)), "C": rng.normal(size=(13, 7)), "x": rng.normal(size=9),
            "y": rng.normal(size=13), "lr": np.asarray(0.05), "mom": np.asarray(0.9)}
    return [A, B, Cm, x, y, lr, mom], outs, vals


@case("sgd_mse_update")
def sgd_mse_update():
    # mean squared error + one gradient step: the backward Gemv's vector is scaled by 2/N where N
    # reaches the graph through a *reduction of the forward Gemv's output*; fusing forward and
    # backward Gemv into one GemvChain would make the chain its own ancestor (round-2 regression)
    rng = np.random.default_rng(62)
    X, y, w = pt.dmatrix("X"), pt.dvector("y"), pt.dvector("w")
    lr = pt.dscalar("lr")
    loss = ((pt.dot(X, w) - y) ** 2).mean()
    g = pytensor.grad(loss, w)
    return [X, y, w, lr], [loss, w - lr * g], {"X": rng.normal(size=(300, 17)), "y": rng.normal(size=300), "w": rng.normal(size=17) * 0.1,
                                                "lr": np.asarray(0.05)}


@case("unsigned_ints")
def unsigned_ints():
    # bit-exact tier on the unsigned dtypes: arithmetic wraps, Sum accumulates in uint64
    # (elemwise.py:1383-1417), comparisons between signed and unsigned go through the common type
    rng = np.random.default_rng(63)
    a8, a16, a32, a64 = (pt.vector(n, dtype=d) for n, d in (("a8", "uint8"), ("a16", "uint16"), ("a32", "uint32"), ("a64", "uint64")))
    M16 = pt.matrix("M16", dtype="uint16")
    i = pt.lvector("i")
    idx = pt.lvector("idx")
    u = lambda v, d: np.asarray(v, dtype=d)  # same-dtype constants: a Python int constant would be int8 and
    # upcast the result to int64, where the reference's C code multiplies in 32 bits (C promotion
    # of uint32 * int8) but its NumPy linker in int64 — not pinnable, the two disagree
    outs = [
        a8 + a8, a16 * a16, a32 - a32[::-1], a64 // u(3, "uint64") + a64 % u(7, "uint64"), a16 + a8, a32 * u(3, "uint32"),
        pt.cast(a64 % u(1000, "uint64"), "uint16"), pt.cast(abs(i), "uint32"),
        pt.cast(a32, "float64") / 7.0, a8 < a16, pt.eq(a32, a32[::-1]), pt.maximum(a16, a16[::-1]), a16 & u(0xFF, "uint16"),
        a32 | u(1, "uint32"), a64 ^ a64[::-1],
        ~a16, a8.sum(), a16.sum(), a32.sum(), a64.sum(), a16.prod(), M16.sum(axis=0), M16.sum(axis=1), M16.max(axis=1), M16.max(axis=0),
        a32.max(), a64.max(), a16[idx], M16[idx], pt.set_subtensor(a32[1:4], u(7, "uint32")), pt.inc_subtensor(a64[idx], a64[idx]), pt.sort(a16),
        pt.argsort(a32), pt.cumsum(a64), pt.argmax(M16, axis=1), pt.concatenate([a16, a16[::-1]]), pt.switch(a8 > u(100, "uint8"), a16, a16 + a16),
        abs(a32),
    ]
    n = 37
    vals = {"a8": rng.integers(0, 256, size=n).astype("uint8"), "a16": rng.integers(0, 65536, size=n).astype("uint16"),
            "a32": rng.integers(0, 2**32, size=n).astype("uint32"), "a64": rng.integers(0, 2**52, size=n).astype("uint64") * 2 + 1,  # (< 2^53: Maximum.c_code's `nan("")` arm makes
            # the C ternary a double, so the reference's C linker rounds 64-bit integer maxima above 2^53 — not pinnable)
            "M16": rng.integers(0, 65536, size=(n, 11)).astype("uint16"), "i": rng.integers(-50, 50, size=n), "idx": np.array([3, 0, 36, 3, 11])}
    return [a8, a16, a32, a64, M16, i, idx], outs, vals


@case("careduce_narrow_out")
def careduce_narrow_out():
    # a raw CAReduce keeps the input dtype on the output and accumulates wide (TestCAReduce cases)
    rng = np.random.default_rng(64)
    xi8, xu8 = pt.matrix("xi8", dtype="int8"), pt.matrix("xu8", dtype="uint8")
    xi16, xb = pt.matrix("xi16", dtype="int16"), pt.matrix("xb", dtype="bool")
    # NOT pinned here: Minimum over an unsigned dtype.  The reference's C code seeds the
    # accumulator with the literal 1 for every `uint*` input (elemwise.py:1609-1611), so its C
    # linker returns min(1, true minimum) while its NumPy linker returns the true minimum; the hip
    # linker follows NumPy (tests/test_gpu_dtypes.py checks that against np.minimum.reduce).
    outs = []
    for x in (xi8, xu8, xi16):
        for axis in (None, (0,), (1,)):
            outs += [CAReduce(ps.add, axis=axis)(x), CAReduce(ps.mul, axis=axis)(x), CAReduce(ps.maximum, axis=axis)(x),
                     CAReduce(ps.or_, axis=axis)(x), CAReduce(ps.and_, axis=axis)(x), CAReduce(ps.xor, axis=axis)(x)]
            if x is not xu8:
                outs.append(CAReduce(ps.minimum, axis=axis)(x))
    outs += [CAReduce(ps.add, axis=None)(xb), CAReduce(ps.mul, axis=(0,))(xb), CAReduce(ps.or_, axis=(1,))(xb), CAReduce(ps.xor, axis=None)(xb)]
    vals = {"xi8": rng.integers(-128, 128, size=(5, 6)).astype("int8"), "xu8": rng.integers(0, 256, size=(5, 6)).astype("uint8"),
            "xi16": rng.integers(-3000, 3000, size=(5, 6)).astype("int16"), "xb": rng.integers(0, 2, size=(5, 6)).astype("bool")}
    return [xi8, xu8, xi16, xb], outs, vals


@case("float16_storage", rtol=1e-3, py_rtol=1e-3)
def float16_storage():
    # float16 is a storage type: the reference has no C code for it (Elemwise falls back to
    # `perform`: NumPy ufuncs, one rounding to half per scalar op); Sum accumulates in float32.
    # rtol 1e-3 = one half ulp (2^-10): +,-,*,/ are reproduced exactly, libm-style ops may differ
    # from NumPy's float32 kernels by one float32 ulp before the rounding to half
    rng = np.random.default_rng(65)
    h, k = pt.vector("h", dtype="float16"), pt.vector("k", dtype="float16")
    H = pt.matrix("H", dtype="float16")
    f = pt.fvector("f")
    outs = [h + k, h * k - h, h / (abs(k) + 1), pt.exp(h * 0.5), pt.tanh(h) * k + pt.sqrt(abs(h)), pt.cast(h, "float32") * f, pt.cast(f, "float16"),
            h.sum(), H.sum(axis=0), H.sum(axis=1), H.max(axis=1), pt.maximum(h, k), h > k, pt.switch(h > 0, h, k), H[1:, ::2], H.T * 2,
            pt.cast(h * 100, "int16")]
    n = 41
    vals = {"h": rng.normal(size=n).astype("float16"), "k": rng.normal(size=n).astype("float16"), "H": rng.normal(size=(7, n)).astype("float16"),
            "f": rng.normal(size=n).astype("float32")}
    return [h, k, H, f], outs, vals


@case("indexing_nd")
def indexing_nd():
    # tests/tensor/test_subtensor.py (TestAdvancedSubtensor: test_index_w_int_and_vec, test_adv_sub_3d,
    # test_advanced_indexing, multi-dimensional index arrays): N-d index arrays, advanced indices
    # mixed with slices, non-adjacent advanced axes (broadcast dims go first), scatter on axis != 0
    rng = np.random.default_rng(70)
    x = pt.dmatrix("x")
    t3 = pt.dtensor3("t3")
    t4 = pt.dtensor4("t4")
    I = pt.lmatrix("I")
    J = pt.lmatrix("J")
    iv = pt.lvector("iv")
    jv = pt.lvector("jv")
    y = pt.dmatrix("y")
    outs = [
        x[I],  # 2-d index array on axis 0: (2, 3, 7)
        x[:, I],  # on axis 1: (8, 2, 3)
        x[I, J],  # pointwise pair of 2-d index arrays: (2, 3)
        x[I, jv[:3]],  # broadcast (2,3) with (3,)
        t3[1:, iv],  # slice then vector on axis 1
        t3[iv, :, jv[: iv.shape[0]]],  # non-adjacent advanced axes: broadcast dim first
        t3[:, iv[:2], jv[:2]],  # adjacent advanced axes in the middle/back
        t4[::2, I, :, 1:3],  # mixed stepped slice, 2-d index, full, slice
        pt.inc_subtensor(x[:, jv], y[:, : jv.shape[0]]),  # scatter on axis 1 with duplicates
        pt.set_subtensor(x[I], 1.5),  # N-d index on axis 0 (last writer wins per duplicate: same value)
        pt.inc_subtensor(x[I, J], y[:2, :3]),  # pointwise scatter-add with duplicates
        pt.inc_subtensor(t3[:, iv[:2], jv[:2]], 2.0),
        pt.inc_subtensor(t3[1:, iv], t3[1:, iv] * 0 + 1.0),
    ]
    vals = {
        "x": rng.normal(size=(8, 7)), "t3": rng.normal(size=(4, 5, 6)), "t4": rng.normal(size=(4, 5, 3, 4)),
        "I": np.array([[0, 4, 4], [2, 0, 3]]), "J": np.array([[6, 1, 1], [0, 1, 6]]), "iv": np.array([3, 0, 3, 1]),
        "jv": np.array([5, 0, 5, 2, 1]), "y": rng.normal(size=(8, 7)),
    }
    return [x, t3, t4, I, J, iv, jv, y], outs, vals


@case("lu_reuse")
def lu_reuse():
    # rewriting/linalg/solvers.py:615-632 (reuse_decomposition_multiple_solves): several Solve nodes
    # against one matrix (and its transpose) factor it ONCE — LUFactor + PivotToPermutations +
    # SolveTriangular pairs; tests/tensor/linalg/test_rewriting (test_lu_decomposition_reused_*).
    # A batched matrix goes through Blockwise(LUFactor); a singular one NaN-fills (perform, lu.py:293).
    rng = np.random.default_rng(71)
    A, b1, B2 = pt.dmatrix("A"), pt.dvector("b1"), pt.dmatrix("B2")
    A3, b3 = pt.dtensor3("A3"), pt.dmatrix("b3")
    S = pt.dmatrix("S")
    from pytensor.tensor.linalg.decomposition.lu import lu_factor, pivot_to_permutation

    LU, piv = lu_factor(A)
    outs = [
        pt.linalg.solve(A, b1), pt.linalg.solve(A, B2), pt.linalg.solve(A.T, b1),
        LU, piv, pivot_to_permutation(piv, inverse=False), pivot_to_permutation(piv, inverse=True),
        pt.linalg.solve(A3, b3[None, :, :] * 1.0)[:, :, 0], pt.linalg.solve(A3, b3[None, :, :] * 2.0)[:, :, 1],
        lu_factor(S)[0],
    ]
    n = 23
    Sv = rng.normal(size=(6, 6))
    Sv[:, 3] = 0.0  # exactly singular: a zero pivot
    vals = {"A": rng.normal(size=(n, n)) + 0.5 * np.eye(n), "b1": rng.normal(size=n), "B2": rng.normal(size=(n, 4)),
            "A3": rng.normal(size=(3, 9, 9)) + 2 * np.eye(9), "b3": rng.normal(size=(9, 2)), "S": Sv}
    return [A, b1, B2, A3, b3, S], outs, vals


@case("solve_sym_eigh_generalised")
def solve_sym_eigh_generalised():
    # linalg/solvers/general.py:17 Solve(assume_a="sym") — scipy sysv reads ONE triangle of A;
    # linalg/decomposition/eigen.py:177-186 Eigh(a, b): the generalised problem A v = w B v
    # (scipy.linalg.eigh(a, b=b, lower=)).  Eigenvector signs are arbitrary: sign-free outputs.
    rng = np.random.default_rng(72)
    A, Bm, b = pt.dmatrix("A"), pt.dmatrix("B"), pt.dvector("b")
    R = pt.dmatrix("R")
    from pytensor.tensor.linalg.decomposition.eigen import Eigh

    w, v = Eigh(lower=True)(A, Bm)
    wu, vu = Eigh(lower=False)(A, Bm)
    outs = [
        pt.linalg.solve(A, b, assume_a="sym"), pt.linalg.solve(A, R, assume_a="sym", lower=True),
        w, pt.abs(v), (v * pt.exp(-w)[None, :]) @ v.T, wu, pt.abs(vu),
    ]
    n = 19

    def sym():
        M = rng.normal(size=(n, n))
        return M + M.T + np.diag(np.linspace(3.0, 9.0, n))  # symmetric, indefinite-ish, well conditioned

    def spd():
        Q = rng.normal(size=(n, n))
        return Q @ Q.T / n + np.eye(n)

    # the two triangles hold DIFFERENT symmetric problems: `lower` decides which one is solved
    Av = np.triu(sym()) + np.tril(sym(), -1)
    Bv = np.triu(spd()) + np.tril(spd(), -1)
    return [A, Bm, b, R], outs, {"A": Av, "B": Bv, "b": rng.normal(size=n), "R": rng.normal(size=(n, 3))}


@case("wide_terms")
def wide_terms():
    # "the harness should also include a wide synthetic model" — north_star's
    # "≈200 fused Elemwise" scale in miniature: 12 independent likelihood terms of four families,
    # logp and its gradient wrt 24 parameters.  Per term one fused Elemwise+Sum kernel and scalar
    # IncSubtensor bookkeeping (widefuse.py: one MultiElemwise launch + one tail kernel).
    rng = np.random.default_rng(73)
    T, N = 12, 301
    mu, ls = pt.dvector("mu"), pt.dvector("ls")
    ys = [pt.dvector(f"y{k}") for k in range(T)]
    terms = []
    for k in range(T):
        r = (ys[k] - mu[k]) * pt.exp(-ls[k])
        fam = k % 4
        if fam == 0:
            terms.append((-0.5 * r**2 - ls[k]).sum())
        elif fam == 1:
            terms.append((-pt.log1p(r**2 / 3.0) * 2.0 - ls[k]).sum())
        elif fam == 2:
            terms.append((-pt.abs(r) - ls[k]).sum())
        else:
            terms.append((-r - 2.0 * pt.softplus(-r) - ls[k]).sum())
    logp = pt.add(*terms)
    vals = {"mu": rng.normal(size=T) * 0.1, "ls": rng.normal(size=T) * 0.1}
    vals.update({f"y{k}": rng.normal(size=N + 7 * k) + 0.1 * k for k in range(T)})  # ragged lengths
    return [mu, ls, *ys], [logp, *pytensor.grad(logp, [mu, ls])], vals
