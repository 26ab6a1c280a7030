"""Golden cases for the dense decompositions lowered late in round 2 (imported by ``make_golden.py``).
TEST INFRASTRUCTURE.

What they pin (SURVEY §8f row 3 leftovers): ``QR`` in its four modes on tall / wide / square inputs
with its gradient, ``SVD`` / ``MatrixPinv`` / ``Lstsq`` (sign- and basis-free functions of the singular
vectors, which LAPACK does not fix either), the tridiagonal LU pair and ``Solve(assume_a=
"tridiagonal")``, ``Eigvalsh``, ``TensorInv`` / ``TensorSolve``, ``BlockDiagonal``.
"""

from __future__ import annotations

import numpy as np
import pytensor
import pytensor.tensor as pt

from make_golden import case


@case("qr_modes")
def qr_modes():
    # linalg/decomposition/qr.py:153-221 (geqrf + orgqr): R's diagonal carries LAPACK's signs, so Q
    # and R are compared entry by entry; pullback 223-318 (static shapes: no IfElse)
    from pytensor.tensor.linalg.decomposition.qr import QR

    rng = np.random.default_rng(81)
    T, W, S = pt.dmatrix("T"), pt.dmatrix("W"), pt.dmatrix("S")
    F = pt.fmatrix("F")
    Ts = pt.tensor("Ts", shape=(13, 7), dtype="float64")
    Ws = pt.tensor("Ws", shape=(6, 11), dtype="float64")
    C1, C2 = pt.dmatrix("C1"), pt.dmatrix("C2")
    Qe, Re = QR(mode="economic")(Ts)
    Qw, Rw = QR(mode="economic")(Ws)
    cost = (Qe * C1).sum() + (Re**2).sum() + (Qw**3).sum() + (Rw * pt.tanh(Rw)).sum()
    outs = [
        *QR(mode="full")(T), *QR(mode="economic")(T), QR(mode="r")(T), *QR(mode="raw")(T),
        *QR(mode="full")(W), *QR(mode="economic")(W), QR(mode="r")(W),
        *QR(mode="full")(S), *QR(mode="economic")(F),
        *pytensor.grad(cost, [Ts, Ws]),
    ]
    Tv, Wv = rng.normal(size=(13, 7)), rng.normal(size=(6, 11))
    vals = {"T": Tv, "W": Wv, "S": rng.normal(size=(9, 9)), "F": rng.normal(size=(8, 5)).astype("float32"),
            "Ts": Tv + 0.1, "Ws": Wv - 0.1, "C1": rng.normal(size=(13, 7)), "C2": rng.normal(size=(7, 7))}
    return [T, W, S, F, Ts, Ws, C1, C2], outs, vals


@case("svd_pinv_lstsq")
def svd_pinv_lstsq():
    # linalg/decomposition/svd.py:19 (np.linalg.svd), inverse.py:14 MatrixPinv (np.linalg.pinv),
    # solvers/lstsq.py:10 Lstsq (np.linalg.lstsq).  Singular vectors are unique up to a sign per pair
    # (distinct singular values) and the complement of an economy factor up to a rotation: the
    # outputs are |U|, |Vt|, reconstructions and projectors.
    from pytensor.tensor.linalg.decomposition.svd import SVD
    from pytensor.tensor.linalg.inverse import MatrixPinv
    from pytensor.tensor.linalg.solvers.lstsq import Lstsq

    rng = np.random.default_rng(82)
    A, B, D = pt.dmatrix("A"), pt.dmatrix("B"), pt.dmatrix("D")
    C = pt.fmatrix("C")
    y, y1 = pt.dmatrix("y"), pt.dvector("y1")
    U, s, Vt = SVD(full_matrices=False)(A)
    Uf, sf, Vtf = SVD(full_matrices=True)(B)
    Ua, sa, Vta = SVD(full_matrices=True)(A)
    Uc, sc, Vtc = SVD(full_matrices=False)(C)
    xs = Lstsq()(A, y, pt.constant(-1.0))
    x1 = Lstsq()(A, y1, pt.constant(1e-3))
    xw = Lstsq()(B, y[:5], pt.constant(-1.0))
    outs = [
        SVD(compute_uv=False)(A), s, pt.abs(U), pt.abs(Vt), (U * s[None, :]) @ Vt,
        sf, pt.abs(Uf), pt.abs(Vtf[:5]), Vtf[5:].T @ Vtf[5:],
        sa, Ua[:, 7:] @ Ua[:, 7:].T, pt.abs(Vta),
        sc, pt.abs(Uc), pt.abs(Vtc),
        MatrixPinv(hermitian=False)(A), MatrixPinv(hermitian=False)(B), MatrixPinv(hermitian=False)(D),
        *xs, *x1, *xw,
    ]
    Av = rng.normal(size=(12, 7))
    Dv = rng.normal(size=(6, 3)) @ rng.normal(size=(3, 8))  # rank 3: the cutoff zeroes three singular values
    vals = {"A": Av, "B": rng.normal(size=(5, 9)), "D": Dv, "C": rng.normal(size=(6, 6)).astype("float32"),
            "y": rng.normal(size=(12, 2)), "y1": rng.normal(size=12)}
    return [A, B, D, C, y, y1], outs, vals


@case("tridiagonal_solve")
def tridiagonal_solve():
    # linalg/solvers/tridiagonal.py:18, 92 (LAPACK gttrf / gttrs, ipiv 1-based) and
    # Solve(assume_a="tridiagonal") (general.py:17; scipy.linalg.solve reads the three diagonals)
    from pytensor.tensor.linalg.solvers.tridiagonal import LUFactorTridiagonal, SolveLUFactorTridiagonal

    rng = np.random.default_rng(83)
    dl, d, du = pt.dvector("dl"), pt.dvector("d"), pt.dvector("du")
    b, Bm = pt.dvector("b"), pt.dmatrix("Bm")
    Am = pt.dmatrix("Am")
    f = LUFactorTridiagonal()(dl, d, du)
    outs = [
        *f,
        SolveLUFactorTridiagonal(b_ndim=1, transposed=False)(*f, b),
        SolveLUFactorTridiagonal(b_ndim=2, transposed=True)(*f, Bm),
        SolveLUFactorTridiagonal(b_ndim=2, transposed=False)(*f, Bm),
        pt.linalg.solve(Am, b, assume_a="tridiagonal"),
        pt.linalg.solve(Am, Bm, assume_a="tridiagonal"),
    ]
    n = 17
    dv = rng.normal(size=n) + np.where(np.arange(n) % 3 == 0, 0.0, 3.0)  # every third pivot is small: rows swap
    dlv, duv = rng.normal(size=n - 1), rng.normal(size=n - 1)
    Av = np.diag(dv) + np.diag(dlv, -1) + np.diag(duv, 1)
    return [dl, d, du, b, Bm, Am], outs, {"dl": dlv, "d": dv, "du": duv, "b": rng.normal(size=n), "Bm": rng.normal(size=(n, 3)), "Am": Av}


@case("linalg_misc")
def linalg_misc():
    # decomposition/eigen.py:363 Eigvalsh (scipy.linalg.eigvalsh, standard and generalised),
    # inverse.py:169 TensorInv, solvers/lstsq.py:39 TensorSolve (np.linalg.tensorinv / tensorsolve),
    # constructors.py:52 BlockDiagonal (scipy.linalg.block_diag, largest common dtype)
    from pytensor.tensor.linalg.constructors import BlockDiagonal
    from pytensor.tensor.linalg.decomposition.eigen import Eigvalsh
    from pytensor.tensor.linalg.inverse import TensorInv
    from pytensor.tensor.linalg.solvers.lstsq import TensorSolve

    rng = np.random.default_rng(84)
    A, Bm = pt.dmatrix("A"), pt.dmatrix("B")
    t4, t3, tb = pt.dtensor4("t4"), pt.dtensor3("t3"), pt.dmatrix("tb")
    m1, m2, m3 = pt.dmatrix("m1"), pt.fmatrix("m2"), pt.bmatrix("m3")
    outs = [
        Eigvalsh(lower=True)(A), Eigvalsh(lower=False)(A, Bm),
        TensorInv(ind=2)(t4), TensorSolve()(t3, tb),
        BlockDiagonal(n_inputs=3)(m1, m2, m3), BlockDiagonal(n_inputs=2)(m2, m2.T),
    ]
    n = 11
    M = rng.normal(size=(n, n))
    Q = rng.normal(size=(n, n))
    vals = {"A": M + M.T, "B": Q @ Q.T / n + np.eye(n),
            "t4": rng.normal(size=(4, 6, 8, 3)) + 0.0, "t3": rng.normal(size=(2, 3, 6)), "tb": rng.normal(size=(2, 3)),
            "m1": rng.normal(size=(3, 2)), "m2": rng.normal(size=(2, 4)).astype("float32"), "m3": np.array([[7, -3]], dtype="int8")}
    return [A, Bm, t4, t3, tb, m1, m2, m3], outs, vals


@case("ifelse_lazy", lazy=True)
def ifelse_lazy():
    # pytensor/ifelse.py:42 IfElse, evaluated lazily by the VM (thunk 300-345): only the branch taken
    # runs — here the branch not taken would raise (a Solve / Dot with mismatched shapes).  Nested
    # conditionals, several outputs per IfElse, a branch shared with an unconditional consumer.
    from pytensor.ifelse import ifelse

    rng = np.random.default_rng(85)
    T, W = pt.dmatrix("T"), pt.dmatrix("W")
    x, y = pt.dvector("x"), pt.dvector("y")
    s = pt.dscalar("s")

    a, b = ifelse(s > 0, (pt.exp(x), x.sum()), (y[:3] * 2.0, y.prod()))
    shared = pt.log1p(x**2)
    inner = ifelse(x[0] > 10.0, shared * 3.0, ifelse(s < 1.0, shared + y[: x.shape[0]], pt.cumsum(shared)))
    outs = [
        ifelse(pt.ge(T.shape[0], T.shape[1]), pt.linalg.solve(T.T @ T, T.T @ y[: T.shape[0]]), pt.linalg.solve(T @ T.T, x)),
        ifelse(pt.ge(W.shape[0], W.shape[1]), pt.linalg.solve(W.T @ W, W.T @ y[: W.shape[0]]), pt.linalg.solve(W @ W.T, x)),
        a, b, inner, shared.sum(),
        ifelse(pt.eq(x.shape[0], 5), pt.dot(x, y[:5]), pt.as_tensor(np.float64(-1.0))),
    ]
    vals = {"T": rng.normal(size=(8, 6)), "W": rng.normal(size=(5, 9)), "x": rng.normal(size=5), "y": rng.normal(size=8), "s": 0.5}
    return [T, W, x, y, s], outs, vals


@case("index_layout_misc")
def index_layout_misc():
    # tensor/extra_ops.py: SearchsortedOp 111, Repeat 639, Bartlett 776, FillDiagonal 839,
    # FillDiagonalOffset 943, Unique 1189, UnravelIndex 1287, RavelMultiIndex 1365, CpuContiguous 47;
    # tensor/reshape.py JoinDims / SplitDims; linalg/decomposition/lu.py:22 LU; signal/conv.py Convolve1d
    from pytensor.tensor.extra_ops import (
        Bartlett, CpuContiguous, FillDiagonal, FillDiagonalOffset, RavelMultiIndex, Repeat, SearchsortedOp, Unique, UnravelIndex,
    )
    from pytensor.tensor.linalg.decomposition.lu import LU
    from pytensor.tensor.reshape import JoinDims, SplitDims
    from pytensor.tensor.signal.conv import Convolve1d

    rng = np.random.default_rng(86)
    xs, v = pt.dvector("xs"), pt.dmatrix("v")
    xi, vi = pt.lvector("xi"), pt.ivector("vi")
    perm = pt.lvector("perm")
    M3 = pt.dtensor3("M3")
    reps = pt.lvector("reps")
    A, Tl, C3 = pt.dmatrix("A"), pt.dmatrix("Tl"), pt.dtensor3("C3")
    val = pt.dscalar("val")
    u = pt.lvector("u")
    uf = pt.dmatrix("uf")
    flat, dims3 = pt.lmatrix("flat"), pt.tensor("dims3", shape=(3,), dtype="int64")
    i0, i1, i2 = pt.lvector("i0"), pt.lvector("i1"), pt.lvector("i2")
    sig, ker = pt.dvector("sig"), pt.dvector("ker")
    fs, fk = pt.fvector("fs"), pt.fvector("fk")
    uniq_all = Unique(return_index=True, return_inverse=True, return_counts=True)(u)
    outs = [
        SearchsortedOp(side="left")(xs, v), SearchsortedOp(side="right")(xs, v), SearchsortedOp(side="left")(xi, vi),
        SearchsortedOp(side="right")(xs[perm], v, pt.argsort(xs[perm])),
        Repeat(axis=1)(M3, reps), Repeat(axis=0)(xs, pt.arange(xs.shape[0]) % 3),
        Bartlett()(pt.as_tensor(np.int64(12))), Bartlett()(xi.shape[0]),
        FillDiagonal()(A, val), FillDiagonal()(Tl, val * 2), FillDiagonal()(C3, val), 
        FillDiagonalOffset()(A, val, pt.as_tensor(np.int64(2))), FillDiagonalOffset()(Tl, val, pt.as_tensor(np.int64(-3))),
        *uniq_all, Unique()(uf), Unique(return_counts=True)(uf)[1],
        *UnravelIndex(order="C")(flat, dims3), *UnravelIndex(order="F")(flat[0], dims3),
        RavelMultiIndex(mode="raise", order="C")(i0, i1, i2, dims3), RavelMultiIndex(mode="wrap", order="F")(i0 - 7, i1 + 9, i2, dims3),
        RavelMultiIndex(mode="clip", order="C")(i0 - 7, i1 + 9, i2, dims3),
        CpuContiguous()(v.T), JoinDims(start_axis=0, n_axes=2)(M3), JoinDims(start_axis=1, n_axes=2)(M3.transpose(1, 0, 2)),
        SplitDims(axis=0)(xs[:12], pt.as_tensor(np.array([3, 2, 2]))),
        *LU()(A), *LU(permute_l=True)(A), *LU(p_indices=True)(A),
        Convolve1d()(sig, ker, pt.as_tensor(np.array(True))), Convolve1d()(sig, ker, pt.as_tensor(np.array(False))), Convolve1d()(ker, sig, pt.as_tensor(np.array(False))),
        Convolve1d()(fs, fk, pt.as_tensor(np.array(True))),
    ]
    xsv = np.sort(rng.normal(size=40))
    xsv[7] = xsv[8] = xsv[9]  # ties: left and right differ
    vv = rng.normal(size=(3, 5))
    vv[0, 0], vv[1, 1] = xsv[8], np.nan
    xiv = np.sort(rng.integers(-20, 20, size=25))
    vals = {"xs": xsv, "v": vv, "xi": xiv, "vi": rng.integers(-25, 25, size=9).astype("int32"), "perm": rng.permutation(40),
            "M3": rng.normal(size=(2, 3, 4)), "reps": np.array([2, 0, 3]), "A": rng.normal(size=(6, 6)), "Tl": rng.normal(size=(7, 4)),
            "C3": rng.normal(size=(3, 3, 3)), "val": -1.25, "u": rng.integers(0, 9, size=50), "uf": np.round(rng.normal(size=(6, 5)), 1),
            "flat": rng.integers(0, 4 * 5 * 3, size=(2, 6)), "dims3": np.array([4, 5, 3]),
            "i0": rng.integers(0, 4, size=8), "i1": rng.integers(0, 5, size=8), "i2": rng.integers(0, 3, size=8),
            "sig": rng.normal(size=50), "ker": rng.normal(size=7), "fs": rng.normal(size=20).astype("float32"), "fk": rng.normal(size=20).astype("float32")}
    return [xs, v, xi, vi, perm, M3, reps, A, Tl, C3, val, u, uf, flat, dims3, i0, i1, i2, sig, ker, fs, fk], outs, vals


@case("expm_and_grad")
def expm_and_grad():
    # linalg/products.py:13 Expm (scipy.linalg.expm) and its pullback 48-77: the Frechet derivative as
    # the upper-right block of expm([[A^T, G], [0, A^T]]).  Norms from 0.02 to 60: no scaling, and
    # several squarings; a float32 input.
    from pytensor.tensor.linalg.products import Expm

    rng = np.random.default_rng(87)
    A, B, Z = pt.dmatrix("A"), pt.dmatrix("B"), pt.dmatrix("Z")
    F = pt.fmatrix("F")
    Wt = pt.dmatrix("Wt")
    cost = (Expm()(A) * Wt).sum()
    outs = [Expm()(A), Expm()(B), Expm()(Z), Expm()(F), pytensor.grad(cost, A), Expm()(-B)]
    n = 9
    Av = rng.normal(size=(n, n)) * 0.4
    Bv = rng.normal(size=(n, n)) * 4.0 - 6.0 * np.eye(n)  # norm ~ 60 with a decaying spectrum
    vals = {"A": Av, "B": Bv, "Z": rng.normal(size=(5, 5)) * 0.004, "F": (rng.normal(size=(6, 6)) * 0.7).astype("float32"),
            "Wt": rng.normal(size=(n, n))}
    return [A, B, Z, F, Wt], outs, vals


@case("choose_permute_conv2d")
def choose_permute_conv2d():
    # tensor/basic.py:4135 Choose (np.choose, tensor choices), 3426 PermuteRowElements (a permutation per
    # row, forward and inverse, broadcast leading dims), signal/conv.py Convolve2d (scipy.signal.convolve)
    from pytensor.tensor.basic import Choose, PermuteRowElements
    from pytensor.tensor.signal.conv import Convolve2d

    rng = np.random.default_rng(88)
    a = pt.lmatrix("a")
    ch = pt.dtensor3("ch")
    chv = pt.dmatrix("chv")
    x2, p2, p1 = pt.dmatrix("x2"), pt.lmatrix("p2"), pt.lvector("p1")
    x1 = pt.dvector("x1")
    img, ker = pt.dmatrix("img"), pt.dmatrix("ker")
    fi, fk = pt.fmatrix("fi"), pt.fmatrix("fk")
    T, F = pt.as_tensor(np.array(True)), pt.as_tensor(np.array(False))
    outs = [
        Choose("raise")(a, ch), Choose("wrap")(a - 5, ch), Choose("clip")(a * 3 - 2, ch), Choose("raise")(a[0], chv),
        PermuteRowElements(inverse=False)(x2, p2), PermuteRowElements(inverse=True)(x2, p2),
        PermuteRowElements(inverse=False)(x2, p1), PermuteRowElements(inverse=True)(x1, p2),
        Convolve2d(method="direct")(img, ker, T), Convolve2d(method="direct")(img, ker, F), Convolve2d(method="auto")(ker, img, F),
        Convolve2d(method="direct")(fi, fk, T),
    ]
    vals = {"a": rng.integers(0, 4, size=(3, 5)), "ch": rng.normal(size=(4, 3, 5)), "chv": rng.normal(size=(4, 5)),
            "x2": rng.normal(size=(4, 6)), "p2": np.stack([rng.permutation(6) for _ in range(4)]), "p1": rng.permutation(6),
            "x1": rng.normal(size=6), "img": rng.normal(size=(14, 11)), "ker": rng.normal(size=(3, 5)),
            "fi": rng.normal(size=(9, 9)).astype("float32"), "fk": rng.normal(size=(4, 4)).astype("float32")}
    return [a, ch, chv, x2, p2, p1, x1, img, ker, fi, fk], outs, vals


@case("rfft_irfft")
def rfft_irfft():
    # tensor/fft.py:11 RFFTOp (np.fft.rfftn over the trailing axes of a batch, re/im stacked last),
    # 76 IRFFTOp (np.fft.irfftn * prod(s)), their pullbacks (each is the other with the interior
    # frequencies halved / doubled).  Even and odd lengths, a 2-d transform, zero padding through s.
    from pytensor.tensor.fft import irfft, rfft

    rng = np.random.default_rng(89)
    x = pt.dmatrix("x")     # (3, 16)
    xo = pt.dmatrix("xo")   # (2, 15)
    x2 = pt.dtensor3("x2")  # (2, 6, 9)
    xf = pt.fmatrix("xf")   # (4, 12)
    w = pt.dtensor3("w")
    X = rfft(x)
    Xo = rfft(xo)
    X2 = rfft(x2)
    cost = (rfft(x, norm="ortho") * w).sum()
    outs = [X, Xo, X2, rfft(xf), irfft(X), irfft(Xo, is_odd=True), irfft(X2, is_odd=True),
            irfft(X * w), pytensor.grad(cost, x), pytensor.grad((irfft(X2, is_odd=True) ** 2).sum(), x2)]
    vals = {"x": rng.normal(size=(3, 16)), "xo": rng.normal(size=(2, 15)), "x2": rng.normal(size=(2, 6, 9)),
            "xf": rng.normal(size=(4, 12)).astype("float32"), "w": rng.normal(size=(3, 9, 2))}
    return [x, xo, x2, xf, w], outs, vals


@case("special_inverse_polygamma", rtol=1e-10)
def special_inverse_polygamma():
    # scalar/math.py:595 PolyGamma, 721 GammaIncInv, 746 GammaIncCInv, 1601 BetaIncInv: no C code in the
    # reference — both of its linkers call scipy.special.  Inside fused Elemwise kernels here.
    import pytensor.scalar as ps
    from pytensor.scalar.math import BetaIncInv, GammaIncCInv, GammaIncInv, NdtriExp, PolyGamma
    from pytensor.tensor.elemwise import Elemwise

    rng = np.random.default_rng(90)
    x, a, b = pt.dvector("x"), pt.dvector("a"), pt.dvector("b")
    p = pt.dvector("p")
    n = pt.lvector("n")
    pg = Elemwise(PolyGamma(ps.upgrade_to_float, name="polygamma"))
    gi = Elemwise(GammaIncInv(ps.upgrade_to_float, name="gammaincinv"))
    gci = Elemwise(GammaIncCInv(ps.upgrade_to_float, name="gammainccinv"))
    bi = Elemwise(BetaIncInv(ps.upgrade_to_float, name="betaincinv"))
    nde = Elemwise(NdtriExp(ps.upgrade_to_float, name="ndtri_exp"))
    ly, lyg = pt.dvector("ly"), pt.dvector("lyg")
    # (the gradient sqrt(2 pi) exp(y + z^2 / 2) magnifies an error of z by z^2: it is pinned where scipy's own
    #  ndtri_exp is accurate to the last bits, y >= -40; far in the tail log_ndtr(scipy's z) misses y by 3e-7)
    outs = [pg(n, x), pg(n[:, None], x[None, :5]) * 1e-3, gi(a, p), gci(a, p), bi(a, b, p), nde(ly), pytensor.grad(nde(lyg).sum(), lyg),
            pt.exp(-gi(a, p)) + pt.log1p(bi(a, b, p)),  # fused with neighbours
            gi(a, pt.constant(np.array(1e-9))) , gci(a, pt.constant(np.array(1e-9))), bi(a, b, pt.constant(np.array(1.0 - 1e-9)))]
    m = 60
    vals = {"x": np.concatenate([rng.uniform(0.05, 30.0, size=m - 8), -rng.uniform(0.1, 4.9, size=8)]), "n": rng.integers(0, 6, size=m),
            "a": 10 ** rng.uniform(-1, 2.5, size=m), "b": 10 ** rng.uniform(-1, 2.5, size=m), "p": rng.uniform(0.001, 0.999, size=m),
            # log-probabilities from 1 - 1e-9 down to where exp underflows (not within 0.05 of log 1/2, where the quantile crosses 0)
            "ly": np.concatenate([-10 ** rng.uniform(-9, -1.5, size=15), -rng.uniform(0.75, 2.0, size=10), -rng.uniform(2.0, 700.0, size=25), -10 ** rng.uniform(3, 4, size=10)])}
    vals["lyg"] = -rng.uniform(0.75, 40.0, size=30)
    return [x, a, b, p, n, ly, lyg], outs, vals


@case("sylvester_lyapunov", rtol=1e-10)
def sylvester_lyapunov():
    # linalg/solvers/linear_control.py: solve_sylvester 123 (Schur + TRSYL in the reference), the continuous
    # Lyapunov equation 167, the discrete one by the bilinear transform 197 and directly (kron + solve),
    # and the gradient of a Sylvester solve (another Sylvester solve, 100-115)
    rng = np.random.default_rng(91)
    A, B, C = pt.dmatrix("A"), pt.dmatrix("B"), pt.dmatrix("C")
    S, Q = pt.dmatrix("S"), pt.dmatrix("Q")
    X = pt.linalg.solve_sylvester(A, B, C)
    outs = [X, pt.linalg.solve_continuous_lyapunov(S, Q), pt.linalg.solve_discrete_lyapunov(S * 0.2, Q, method="bilinear"),
            pt.linalg.solve_discrete_lyapunov(S * 0.2, Q, method="direct"), *pytensor.grad((X**2).sum(), [A, C])]
    m, n = 7, 5
    Qv = rng.normal(size=(6, 6))
    vals = {"A": rng.normal(size=(m, m)) + 3.0 * np.eye(m), "B": rng.normal(size=(n, n)) + 3.0 * np.eye(n), "C": rng.normal(size=(m, n)),
            "S": rng.normal(size=(6, 6)) - 3.0 * np.eye(6), "Q": Qv @ Qv.T}
    return [A, B, C, S, Q], outs, vals


@case("gp_marginal_likelihood", rtol=1e-10)
def gp_marginal_likelihood():
    # a Gaussian-process marginal likelihood with its gradient — the op mix of pymc's gp.Marginal: pairwise
    # squared distances by broadcasting, Exp, Cholesky (linalg/decomposition/cholesky.py:18), two triangular
    # solves, log-determinant from the diagonal, and the pullbacks of all of them
    rng = np.random.default_rng(92)
    X, y = pt.dmatrix("X"), pt.dvector("y")
    ls, eta, sigma = pt.dvector("ls"), pt.dscalar("eta"), pt.dscalar("sigma")
    Xs = X / ls[None, :]
    d2 = (Xs**2).sum(axis=1)[:, None] + (Xs**2).sum(axis=1)[None, :] - 2.0 * Xs @ Xs.T
    K = eta**2 * pt.exp(-0.5 * pt.maximum(d2, 0.0)) + (sigma**2 + 1e-6) * pt.eye(X.shape[0])
    L = pt.linalg.cholesky(K)
    alpha = pt.linalg.solve_triangular(L, y, lower=True)
    logp = -0.5 * (alpha**2).sum() - pt.log(pt.diag(L)).sum() - 0.5 * y.shape[0] * np.log(2 * np.pi)
    n = 60
    Xv = rng.normal(size=(n, 3))
    vals = {"X": Xv, "y": np.sin(Xv[:, 0]) + 0.1 * rng.normal(size=n), "ls": np.array([0.9, 1.4, 2.0]), "eta": 1.3, "sigma": 0.35}
    return [X, y, ls, eta, sigma], [logp, *pytensor.grad(logp, [ls, eta, sigma]), alpha], vals


@case("mixture_ordinal_logp", rtol=1e-10)
def mixture_ordinal_logp():
    # a normal mixture (logsumexp over components) plus an ordered-logistic likelihood (cutpoints by
    # cumsum of positives, category probabilities as sigmoid differences, picked per row by advanced
    # indexing), logp and gradients — the op mix of pm.NormalMixture / pm.OrderedLogistic
    rng = np.random.default_rng(93)
    yv, w_raw, mu, ls = pt.dvector("yv"), pt.dvector("w_raw"), pt.dvector("mu"), pt.dvector("ls")
    eta, cut_raw = pt.dvector("eta"), pt.dvector("cut_raw")
    cat = pt.lvector("cat")
    logw = w_raw - pt.logsumexp(w_raw)
    z = (yv[:, None] - mu[None, :]) * pt.exp(-ls)[None, :]
    comp = logw[None, :] - 0.5 * z**2 - ls[None, :] - 0.5 * np.log(2 * np.pi)
    lp_mix = pt.logsumexp(comp, axis=1).sum()
    cuts = pt.cumsum(pt.concatenate([cut_raw[:1], pt.exp(cut_raw[1:])]))
    cdf = pt.sigmoid(cuts[None, :] - eta[:, None])
    probs = pt.concatenate([cdf[:, :1], cdf[:, 1:] - cdf[:, :-1], 1.0 - cdf[:, -1:]], axis=1)
    lp_ord = pt.log(probs[pt.arange(eta.shape[0]), cat]).sum()
    logp = lp_mix + lp_ord
    n, k, c = 300, 4, 5
    vals = {"yv": rng.normal(size=n) * 2.0, "w_raw": rng.normal(size=k), "mu": np.linspace(-3, 3, k), "ls": rng.normal(size=k) * 0.2,
            "eta": rng.normal(size=n), "cut_raw": np.array([-1.5, 0.1, -0.3, 0.2]), "cat": rng.integers(0, c, size=n)}
    return [yv, w_raw, mu, ls, eta, cut_raw, cat], [logp, *pytensor.grad(logp, [w_raw, mu, ls, eta, cut_raw]), probs.sum(axis=1)], vals


@case("kalman_filter_scan", rtol=1e-9, py_optimizer="fast_compile")  # (the unoptimised PerformLinker run of this graph raises "expected an ndarray" inside the reference)
def kalman_filter_scan():
    # a linear-Gaussian state-space log-likelihood by a Kalman filter inside Scan (3 states, 2 observables,
    # 25 steps): per step Dot / Solve / Det on small matrices, the filtered state carried as two sit-sot
    # outputs, the gradient wrt the transition and noise parameters through the scan
    rng = np.random.default_rng(94)
    Y = pt.dmatrix("Y")
    T_, Z, q, r = pt.dmatrix("T"), pt.dmatrix("Z"), pt.dvector("q"), pt.dvector("r")
    a0, P0 = pt.dvector("a0"), pt.dmatrix("P0")

    def step(y, a, P, T_, Z, Qm, Rm):
        a_p = T_ @ a
        P_p = T_ @ P @ T_.T + Qm
        v = y - Z @ a_p
        F = Z @ P_p @ Z.T + Rm
        Kg = pt.linalg.solve(F, Z @ P_p).T  # P_p Z^T F^-1
        a_n = a_p + Kg @ v
        P_n = P_p - Kg @ Z @ P_p
        ll = -0.5 * (pt.log(pt.linalg.det(F)) + v @ pt.linalg.solve(F, v) + 2 * np.log(2 * np.pi))
        return a_n, P_n, ll

    (a_s, P_s, lls), _ = pytensor.scan(step, sequences=[Y], outputs_info=[a0, P0, None],
                                        non_sequences=[T_, Z, pt.diag(pt.exp(q)), pt.diag(pt.exp(r))])
    ll = lls.sum()
    Tv = np.array([[0.8, 0.1, 0.0], [0.0, 0.7, 0.2], [0.1, 0.0, 0.9]])
    vals = {"Y": rng.normal(size=(25, 2)), "T": Tv, "Z": rng.normal(size=(2, 3)), "q": np.array([-1.0, -0.5, -1.5]), "r": np.array([-0.7, -0.2]),
            "a0": np.zeros(3), "P0": np.eye(3)}
    # (the filtered state itself is not an output: with `a_s[-1]` beside the gradient the reference's own C linker raises "expected an ndarray")
    return [Y, T_, Z, q, r, a0, P0], [ll, *pytensor.grad(ll, [T_, q, r])], vals


@case("glm_binomial_studentt", rtol=1e-10, py_rtol=1e-7)  # (d gammaln = Psi: 10-digit constants in the reference's C code, scipy in its NumPy backend)
def glm_binomial_studentt():
    # two likelihood families with their priors, logp and gradients: a hierarchical binomial regression
    # (logit link, group effects gathered by index, binomial coefficient through gammaln) and a robust
    # regression with a Student-t likelihood whose degrees of freedom are a parameter (gammaln of nu / 2),
    # Gamma, Beta and half-Cauchy prior densities
    rng = np.random.default_rng(95)
    X = pt.dmatrix("X")
    kk, nn, gidx = pt.lvector("kk"), pt.lvector("nn"), pt.lvector("gidx")
    beta, u, log_su = pt.dvector("beta"), pt.dvector("u"), pt.dscalar("log_su")
    yt = pt.dvector("yt")
    log_nu, log_s, p_b = pt.dscalar("log_nu"), pt.dscalar("log_s"), pt.dscalar("p_b")
    eta = X @ beta + pt.exp(log_su) * u[gidx]
    n_f, k_f = nn.astype("float64"), kk.astype("float64")
    lp_binom = (pt.gammaln(n_f + 1) - pt.gammaln(k_f + 1) - pt.gammaln(n_f - k_f + 1) + k_f * pt.log(pt.sigmoid(eta)) + (n_f - k_f) * pt.log1p(-pt.sigmoid(eta))).sum()
    lp_u = (-0.5 * u**2).sum() - pt.log1p(pt.exp(2 * log_su)) + log_su  # half-Cauchy on exp(log_su), with the Jacobian
    nu, s = pt.exp(log_nu) + 1.0, pt.exp(log_s)
    r = (yt - X @ beta) / s
    lp_t = (pt.gammaln((nu + 1) / 2) - pt.gammaln(nu / 2) - 0.5 * pt.log(nu * np.pi) - log_s - (nu + 1) / 2 * pt.log1p(r**2 / nu)).sum()
    lp_nu = 2.0 * pt.log(0.1) - pt.gammaln(2.0) + (2.0 - 1.0) * pt.log(nu) - 0.1 * nu + log_nu  # Gamma(2, 0.1)
    pb = pt.sigmoid(p_b)
    lp_beta_prior = pt.gammaln(5.0) - pt.gammaln(2.0) - pt.gammaln(3.0) + 1.0 * pt.log(pb) + 2.0 * pt.log1p(-pb) + pt.log(pb) + pt.log1p(-pb)  # Beta(2, 3) + Jacobian
    logp = lp_binom + lp_u + lp_t + lp_nu + lp_beta_prior + pb * beta.sum()
    n, k, g = 400, 5, 12
    nv = rng.integers(1, 30, size=n)
    vals = {"X": rng.normal(size=(n, k)), "kk": rng.binomial(nv, 0.4), "nn": nv, "gidx": rng.integers(0, g, size=n), "beta": rng.normal(size=k) * 0.3,
            "u": rng.normal(size=g), "log_su": -0.4, "yt": rng.standard_t(4, size=n), "log_nu": 1.1, "log_s": 0.2, "p_b": 0.3}
    wrt = [beta, u, log_su, log_nu, log_s, p_b]
    return [X, kk, nn, gidx, beta, u, log_su, yt, log_nu, log_s, p_b], [logp, *pytensor.grad(logp, wrt)], vals


@case("softmax_dirichlet_censored", rtol=1e-10)
def softmax_dirichlet_censored():
    # a softmax (multinomial-logit) regression with a categorical likelihood (LogSoftmax, row gather), a
    # Dirichlet density over simplex weights built by softmax, and a right-censored normal likelihood
    # (log survival function through erfcx in the tail, log1p(-erfc / 2) elsewhere), logp and gradients
    rng = np.random.default_rng(96)
    X, W = pt.dmatrix("X"), pt.dmatrix("W")
    cat = pt.lvector("cat")
    a_raw, conc = pt.dvector("a_raw"), pt.dvector("conc")
    yc, mu_c, ls_c = pt.dvector("yc"), pt.dscalar("mu_c"), pt.dscalar("ls_c")
    cens = pt.bvector("cens")
    lsm = pt.special.log_softmax(X @ W, axis=1)
    lp_cat = lsm[pt.arange(X.shape[0]), cat].sum()
    w = pt.special.softmax(a_raw)
    lp_dir = pt.gammaln(conc.sum()) - pt.gammaln(conc).sum() + ((conc - 1.0) * pt.log(w)).sum()
    z = (yc - mu_c) * pt.exp(-ls_c)
    log_pdf = -0.5 * z**2 - ls_c - 0.5 * np.log(2 * np.pi)
    log_sf = pt.switch(z > 1.0, pt.log(0.5 * pt.erfcx(z / np.sqrt(2.0))) - 0.5 * z**2, pt.log1p(-0.5 * pt.erfc(-z / np.sqrt(2.0))))
    lp_cens = pt.switch(cens, log_sf, log_pdf).sum()
    logp = lp_cat + lp_dir + lp_cens
    n, k, c = 250, 4, 6
    vals = {"X": rng.normal(size=(n, k)), "W": rng.normal(size=(k, c)) * 0.5, "cat": rng.integers(0, c, size=n), "a_raw": rng.normal(size=7),
            "conc": rng.uniform(0.5, 3.0, size=7), "yc": rng.normal(size=n) * 1.5 + 0.3, "mu_c": 0.1, "ls_c": 0.2, "cens": (rng.random(n) < 0.3).astype("int8")}
    return [X, W, cat, a_raw, conc, yc, mu_c, ls_c, cens], [logp, *pytensor.grad(logp, [W, a_raw, mu_c, ls_c]), w], vals


@case("hmm_garch_scans", rtol=1e-9)
def hmm_garch_scans():
    # two more recurrences through Scan with their gradients: the forward algorithm of a 3-state hidden Markov
    # model in log space (a logsumexp over a matrix inside every step) and a GARCH(1,1) variance recursion with
    # taps -1 / -2 on its state and a Switch in the step
    rng = np.random.default_rng(97)
    obs = pt.dvector("obs")
    logA_raw, mu_s, ls_s = pt.dmatrix("logA_raw"), pt.dvector("mu_s"), pt.dvector("ls_s")
    ret = pt.dvector("ret")
    om, al, be = pt.dscalar("om"), pt.dscalar("al"), pt.dscalar("be")
    logA = logA_raw - pt.logsumexp(logA_raw, axis=1, keepdims=True)

    def emis(y):
        z = (y - mu_s) * pt.exp(-ls_s)
        return -0.5 * z**2 - ls_s - 0.5 * np.log(2 * np.pi)

    def fwd(y, la, logA, mu_s, ls_s):
        z = (y - mu_s) * pt.exp(-ls_s)
        e = -0.5 * z**2 - ls_s - 0.5 * np.log(2 * np.pi)
        return pt.logsumexp(la[:, None] + logA, axis=0) + e

    la0 = emis(obs[0]) - np.log(3.0)
    las, _ = pytensor.scan(fwd, sequences=[obs[1:]], outputs_info=[la0], non_sequences=[logA, mu_s, ls_s])
    ll_hmm = pt.logsumexp(las[-1])

    def garch(r_prev, h1, h2, om, al, be):
        h = pt.exp(om) + pt.sigmoid(al) * 0.3 * r_prev**2 + pt.sigmoid(be) * 0.6 * pt.switch(h1 > h2, h1, 0.5 * (h1 + h2))
        return h

    hs, _ = pytensor.scan(garch, sequences=[ret[:-1]], outputs_info=[dict(initial=pt.stack([ret.var(), ret.var()]), taps=[-1, -2])],
                          non_sequences=[om, al, be])
    ll_garch = (-0.5 * pt.log(hs) - 0.5 * ret[1:] ** 2 / hs).sum()
    ll = ll_hmm + ll_garch
    T = 40
    vals = {"obs": rng.normal(size=T) + np.repeat([-2.0, 0.0, 2.0, 0.0], T // 4), "logA_raw": rng.normal(size=(3, 3)), "mu_s": np.array([-2.0, 0.1, 2.2]),
            "ls_s": np.array([-0.2, 0.1, 0.0]), "ret": rng.normal(size=T) * 0.8, "om": -1.0, "al": 0.2, "be": 0.5}
    return [obs, logA_raw, mu_s, ls_s, ret, om, al, be], [ll, *pytensor.grad(ll, [logA_raw, mu_s, ls_s, om, al, be]), hs[-1]], vals


@case("lkj_packed_spline", rtol=1e-10)
def lkj_packed_spline():
    # a covariance built from a packed Cholesky vector (set_subtensor on tril indices, exp on the diagonal) with
    # a multivariate-normal logp through triangular solves, and a piecewise-linear basis regression whose knots
    # are located by SearchsortedOp and gathered by index; logp and gradients
    rng = np.random.default_rng(98)
    packed = pt.dvector("packed")
    Yv = pt.dmatrix("Yv")
    knots, coef, xs, ys = pt.dvector("knots"), pt.dvector("coef"), pt.dvector("xs"), pt.dvector("ys")
    k = 4
    rows, cols = np.tril_indices(k)
    L0 = pt.set_subtensor(pt.zeros((k, k))[rows, cols], packed)
    L = pt.set_subtensor(L0[np.arange(k), np.arange(k)], pt.exp(pt.diag(L0)))
    Zs = pt.linalg.solve_triangular(L, Yv.T, lower=True)
    lp_mvn = -0.5 * (Zs**2).sum() - Yv.shape[0] * pt.log(pt.diag(L)).sum() - 0.5 * k * Yv.shape[0] * np.log(2 * np.pi)
    seg = pt.clip(pt.searchsorted(knots, xs, side="right") - 1, 0, knots.shape[0] - 2)
    t = (xs - knots[seg]) / (knots[seg + 1] - knots[seg])
    fit = coef[seg] * (1.0 - t) + coef[seg + 1] * t
    lp_fit = (-0.5 * (ys - fit) ** 2).sum()
    logp = lp_mvn + lp_fit
    n = 120
    xv = np.sort(rng.uniform(0.0, 1.0, size=n))
    vals = {"packed": rng.normal(size=k * (k + 1) // 2) * 0.4, "Yv": rng.normal(size=(35, k)), "knots": np.linspace(0.0, 1.0, 9),
            "coef": rng.normal(size=9), "xs": xv, "ys": np.sin(6 * xv) + 0.1 * rng.normal(size=n)}
    return [packed, Yv, knots, coef, xs, ys], [logp, *pytensor.grad(logp, [packed, coef]), seg], vals
