"""Dense decompositions and what is composed from them: ``QR``, ``SVD``, ``MatrixPinv``, ``Lstsq``,
the tridiagonal LU pair, ``Eigvalsh``, ``TensorInv`` / ``TensorSolve``, ``BlockDiagonal``.

Reference: linalg/decomposition/qr.py:20 (perform 153-221: geqrf + orgqr), decomposition/svd.py:19
(np.linalg.svd), inverse.py:14 ``MatrixPinv`` (np.linalg.pinv), 169 ``TensorInv``,
solvers/lstsq.py:10 ``Lstsq`` (np.linalg.lstsq), 39 ``TensorSolve``, solvers/tridiagonal.py:18, 92
(gttrf / gttrs), decomposition/eigen.py:363 ``Eigvalsh``, constructors.py:52 ``BlockDiagonal``
(scipy.linalg.block_diag).  Kernels: csrc/decomp.hip (correct-first tier).
Leading batch dimensions arrive through ``Blockwise`` (dispatch/linalg.py loops the items).
"""

from __future__ import annotations

import numpy as np

from pytensor_amd import ffi
from pytensor_amd.device import DeviceArray, contiguous_strides, copy_into
from pytensor_amd.dispatch import handler
from pytensor_amd.dispatch.linalg import _dt, _require_float
from pytensor_amd.executor import HostValue


def _cs(shape):
    return contiguous_strides(tuple(shape))


def _reshape(x: DeviceArray, shape) -> DeviceArray:
    x = x.contiguous()
    return x.view(tuple(shape), _cs(shape))


def _t(x: DeviceArray) -> DeviceArray:
    """transposed view of a matrix"""
    return x.view((x.shape[1], x.shape[0]), (x.strides[1], x.strides[0]))


def _matrix(env, v, what):
    from pytensor_amd.dispatch.linalg import _lapack_operands

    (x,) = _lapack_operands(env, what, v)
    if x.ndim != 2:
        # (np.linalg raises LinAlgError: "%d-dimensional array given. Array must be two-dimensional")
        raise np.linalg.LinAlgError(f"{what}: {x.ndim}-dimensional array given. Array must be two-dimensional")
    return x


# ---- QR -------------------------------------------------------------------------------------------
def geqrf_device(env, x: DeviceArray):
    """(packed factors m x n, tau[min(m, n)]) — LAPACK's geqrf on a copy of ``x``"""
    m, n = x.shape
    qr = x.contiguous_copy()
    tau = DeviceArray.empty((min(m, n),), x.dtype)
    ffi.check(env.lib.pthip_geqrf(_dt(x), 1, m, n, qr.ptr, tau.ptr))
    return qr, tau


def orgqr_device(env, qr: DeviceArray, tau: DeviceArray, ncols: int) -> DeviceArray:
    m, n = qr.shape
    q = DeviceArray.empty((m, ncols), qr.dtype)
    ffi.check(env.lib.pthip_orgqr(_dt(qr), 1, m, ncols, tau.shape[0], qr.ptr, n, m * n, tau.ptr, q.ptr))
    return q


def _triu(env, qr: DeviceArray, rows: int, lower=False, unit=False) -> DeviceArray:
    m, n = qr.shape
    r = DeviceArray.empty((rows, n), qr.dtype)
    ffi.check(env.lib.pthip_triu(_dt(qr), 1, rows, n, qr.ptr, n, m * n, r.ptr, int(lower), int(unit)))
    return r


@handler("QR")
def qr_op(node, inputs, env):
    mode = node.params["mode"]
    x = _matrix(env, inputs[0], "QR")
    m, n = x.shape
    qr, tau = geqrf_device(env, x)
    # perform 171-174: economic / raw keep the leading n rows of R when m >= n
    R = _triu(env, qr, m if (mode not in ("economic", "raw") or m < n) else n)
    if mode == "r":
        return [R]
    if mode == "raw":
        return [qr, tau, R]
    if m < n:
        Q = orgqr_device(env, qr, tau, m)
    elif mode == "economic":
        Q = orgqr_device(env, qr, tau, n)
    else:
        Q = orgqr_device(env, qr, tau, m)
    return [Q, R]


# ---- SVD ------------------------------------------------------------------------------------------
def svd_device(env, x: DeviceArray, full_matrices: bool, compute_uv: bool):
    """(U, S, Vt) or S alone.  One-sided Jacobi on the rows of the wide orientation of ``x``
    (csrc/decomp.hip); the tall factor's null / complementary directions are completed from a
    Householder Q, which LAPACK leaves just as arbitrary."""
    m, n = x.shape
    tall = m >= n
    X = (_t(x).contiguous_copy() if tall else x.contiguous_copy())  # r x c, r <= c
    r, c = X.shape
    dt = x.dtype
    S = DeviceArray.empty((r,), dt)
    if r == 0:
        if not compute_uv:
            return [S]
        eye = lambda k: env.to_device(HostValue(np.eye(k, dtype=dt)))
        if full_matrices:
            return [eye(m), S, eye(n)]
        return [DeviceArray.empty((m, 0), dt), S, DeviceArray.empty((0, n), dt)]
    work = DeviceArray.empty((r, r), dt) if compute_uv else DeviceArray.empty((1,), dt)
    Wt = DeviceArray.empty((r, c), dt) if compute_uv else work
    Pt = DeviceArray.empty((r, r), dt) if compute_uv else work
    ffi.check(env.lib.pthip_svd_rows(_dt(x), 1, r, c, int(compute_uv), X.ptr, work.ptr, S.ptr, Wt.ptr, Pt.ptr))
    env.keepalive.extend((X, work))
    if not compute_uv:
        return [S]
    # completion: Householder Q of Wt^T (c x r); its column i replaces a null row i, its columns
    # r.. (full_matrices) span the complement
    rows = c if full_matrices else r
    qr, tau = geqrf_device(env, _t(Wt))
    Q = orgqr_device(env, qr, tau, rows)
    if rows == r:
        big = Wt
    else:
        big = DeviceArray.empty((rows, c), dt)
        copy_into(big.view((r, c), _cs((r, c))), Wt)
    ffi.check(env.lib.pthip_fill_null_rows(_dt(x), 1, rows, c, big.ptr, S.ptr, r, Q.ptr, rows))
    env.keepalive.extend((Q, qr, tau))
    if tall:  # x^T = P S Wt  ->  x = Wt^T S P^T
        return [_t(big).contiguous(), S, Pt]
    return [_t(Pt).contiguous(), S, big]


@handler("SVD")
def svd_op(node, inputs, env):
    x = _matrix(env, inputs[0], "SVD")
    return svd_device(env, x, bool(node.params["full_matrices"]), bool(node.params["compute_uv"]))


def _pinv_apply(env, U, S, Vt, rcond_dev, rhs=None):
    """Vt^T diag(1/s where s > rcond * s_max) U^T [rhs]  (np.linalg.pinv / the minimum-norm lstsq)"""
    from pytensor_amd.dispatch.blas import gemm_device
    from pytensor_amd.dispatch.elemwise import launch_elemwise

    dt = str(S.dtype)
    k = S.shape[0]
    # sinv_i = s_i > rcond * s_0 ? 1 / s_i : 0   (s is descending: s_0 is the maximum)
    body = {"in_dtypes": [dt, dt, dt], "out_dtypes": [dt],
            "body": [{"op": "Mul", "in": [["i", 1], ["i", 2]], "dtype": dt},
                     {"op": "GT", "in": [["i", 0], ["t", 0]], "dtype": "bool"},
                     {"op": "Reciprocal", "in": [["i", 0]], "dtype": dt},
                     {"op": "Switch", "in": [["t", 1], ["t", 2], ["c", 0.0, dt]], "dtype": dt}],
            "outs": [["t", 3]]}
    s0 = S.view((k,), (0,))
    (sinv,), _, _ = launch_elemwise(body, [S, s0, rcond_dev.view((k,), (0,))], (k,), [dt], None, env)
    right = _t(U) if rhs is None else gemm_device(env, 1.0, _t(U), rhs)  # k x m  |  k x nrhs
    cols = right.shape[1]
    scale = {"in_dtypes": [dt, dt], "out_dtypes": [dt], "body": [{"op": "Mul", "in": [["i", 0], ["i", 1]], "dtype": dt}], "outs": [["t", 0]]}
    (scaled,), _, _ = launch_elemwise(scale, [right, sinv.view((k, cols), (1, 0))], (k, cols), [dt], None, env)
    return gemm_device(env, 1.0, _t(Vt), scaled)


@handler("MatrixPinv")
def matrix_pinv(node, inputs, env):
    x = _matrix(env, inputs[0], "MatrixPinv")
    m, n = x.shape
    if m == 0 or n == 0:
        return [DeviceArray.empty((n, m), x.dtype)]
    U, S, Vt = svd_device(env, x, False, True)
    rcond = env.to_device(HostValue(np.asarray([1e-15], dtype=x.dtype)))  # np.linalg.pinv's default
    return [_pinv_apply(env, U, S, Vt, rcond)]


@handler("Lstsq")
def lstsq(node, inputs, env):
    """np.linalg.lstsq(x, y, rcond) -> (solution, residuals, rank, singular values): the minimum-norm
    solution through the SVD as gelsd computes it; ``rank`` decides the shape of ``residuals`` (one
    host read, like every data-dependent shape)."""
    from pytensor_amd.dispatch.blas import gemm_device
    from pytensor_amd.dispatch.elemwise import _cast, launch_elemwise

    a = _matrix(env, inputs[0], "Lstsq")
    b = env.to_device(inputs[1])
    rc = np.asarray(env.to_host(inputs[2]))
    m, n = a.shape
    if b.shape[0] != m:
        raise np.linalg.LinAlgError("Incompatible dimensions")
    dt = np.dtype(a.dtype)
    if np.dtype(b.dtype) != dt:
        b = _cast(env, b.contiguous(), dt)
    eps = float(np.finfo(dt).eps)
    rcond = eps * max(m, n) if rc.dtype == object or rc.size == 0 else float(rc)
    if rcond <= 0 or rcond >= 1:
        rcond = eps  # LAPACK gelsd -> dlalsd: "IF( (RCOND.LE.ZERO) .OR. (RCOND.GE.ONE) ) RCND = EPS"
    b2 = b if b.ndim == 2 else b.view((m, 1), (b.strides[0], 0))
    U, S, Vt = svd_device(env, a, False, True)
    s_host = np.asarray(env.to_host(S))
    rank = int((s_host > rcond * (s_host[0] if s_host.size else 0.0)).sum())
    x = _pinv_apply(env, U, S, Vt, env.to_device(HostValue(np.asarray([rcond], dtype=dt))), rhs=b2.contiguous())
    nrhs = b2.shape[1]
    if rank == n and m > n:
        res = gemm_device(env, -1.0, a, x, 1.0, b2.contiguous())  # b - a x
        sq = {"in_dtypes": [str(dt)], "out_dtypes": [str(dt)], "body": [{"op": "Sqr", "in": [["i", 0]], "dtype": str(dt)}], "outs": [["t", 0]]}
        (r2,), _, _ = launch_elemwise(sq, [res], (m, nrhs), [str(dt)], None, env)
        resid = _column_sums(env, r2)
    else:
        resid = DeviceArray.empty((0,), dt)
    xs = x if b.ndim == 2 else x.view((n,), (x.strides[0],))
    f64 = np.dtype("float64")
    up = lambda v: v if np.dtype(v.dtype) == f64 else _cast(env, v.contiguous(), f64)  # Lstsq.make_node: d-typed outputs
    return [up(xs), up(resid), HostValue(np.asarray(rank, dtype="int32")), up(S)]


def _column_sums(env, x: DeviceArray) -> DeviceArray:
    """sum over axis 0 of a contiguous matrix, as ones^T x on the GEMM path"""
    from pytensor_amd.dispatch.blas import gemm_device

    m, n = x.shape
    ones = env.to_device(HostValue(np.ones((1, m), dtype=x.dtype)))
    return gemm_device(env, 1.0, ones, x).view((n,), (1,))


# ---- tridiagonal LU -------------------------------------------------------------------------------
def gttrf_device(env, dl, d, du):
    n = d.shape[0]
    if dl.shape[0] != max(n - 1, 0) or du.shape[0] != max(n - 1, 0):
        raise ValueError("LUFactorTridiagonal: diagonals of inconsistent lengths")
    dl, d, du = (v.contiguous_copy() for v in (dl, d, du))
    du2 = DeviceArray.empty((max(n - 2, 0),), d.dtype)
    ipiv = DeviceArray.empty((n,), "int32")
    ffi.check(env.lib.pthip_gttrf(_dt(d), 1, n, dl.ptr, d.ptr, du.ptr, du2.ptr, ipiv.ptr))
    return dl, d, du, du2, ipiv


def gttrs_device(env, dl, d, du, du2, ipiv, b, transposed):
    n = d.shape[0]
    if b.shape[0] != n:
        raise ValueError(f"SolveLUFactorTridiagonal: b has {b.shape[0]} rows, the system {n}")
    out = b.contiguous_copy()
    nrhs = out.size // n if n else 0
    ffi.check(env.lib.pthip_gttrs(_dt(d), 1, n, nrhs, int(transposed), dl.contiguous().ptr, d.contiguous().ptr, du.contiguous().ptr,
                                  du2.contiguous().ptr, ipiv.contiguous().ptr, out.ptr))
    return out


def _same_float(env, vals, what):
    from pytensor_amd.dispatch.linalg import _lapack_operands

    return _lapack_operands(env, what, *vals)


@handler("LUFactorTridiagonal")
def lu_factor_tridiagonal(node, inputs, env):
    dl, d, du = _same_float(env, inputs, "LUFactorTridiagonal")
    return list(gttrf_device(env, dl, d, du))


@handler("SolveLUFactorTridiagonal")
def solve_lu_factor_tridiagonal(node, inputs, env):
    dl, d, du, du2, b = _same_float(env, [inputs[0], inputs[1], inputs[2], inputs[3], inputs[5]], "SolveLUFactorTridiagonal")
    ipiv = env.to_device(inputs[4])
    if str(ipiv.dtype) != "int32":
        from pytensor_amd.dispatch.elemwise import _cast

        ipiv = _cast(env, ipiv.contiguous(), "int32")
    return [gttrs_device(env, dl, d, du, du2, ipiv, b, bool(node.params["transposed"]))]


def solve_tridiagonal(env, A: DeviceArray, b: DeviceArray, b_ndim: int) -> DeviceArray:
    """``Solve(assume_a="tridiagonal")`` (scipy.linalg.solve -> gttrf + gttrs on the three diagonals
    of ``A``; everything else in ``A`` is ignored, as scipy ignores it)"""
    n = A.shape[-1]
    if A.ndim != 2 or b.ndim != b_ndim:
        # batched (Blockwise; reference tests/tensor/rewriting/linalg/test_solvers.py): the (small) batch on the host, like
        # dispatch/lu.solve_general — each item is a factorisation and a solve launch
        from pytensor_amd.device import contiguous_strides, copy_into
        from pytensor_amd.dispatch.lu import _batchify

        bshape = tuple(np.broadcast_shapes(A.shape[:-2], b.shape[: b.ndim - b_ndim]))
        core_b = b.shape[b.ndim - b_ndim:]
        Ab, bbm = _batchify(A, 2, bshape), _batchify(b, b_ndim, bshape)
        out = DeviceArray.empty((*bshape, *core_b), b.dtype)
        step = int(np.prod(core_b)) if core_b else 1
        for k in range(Ab.shape[0]):
            xk = solve_tridiagonal(env, Ab.view((n, n), (n, 1), k * n * n), bbm.view(core_b, contiguous_strides(core_b), k * step), b_ndim)
            copy_into(out.view(core_b, contiguous_strides(core_b), k * step), xk)
        return out
    diag = lambda off: A.view((max(n - abs(off), 0),), (A.strides[0] + A.strides[1],), (-off) * A.strides[0] if off < 0 else off * A.strides[1])
    f = gttrf_device(env, diag(-1), diag(0), diag(1))
    return gttrs_device(env, *f, b, False)


# ---- composed from existing device ops ----------------------------------------------------------------
@handler("Eigvalsh")
def eigvalsh(node, inputs, env):
    from pytensor_amd.dispatch.lu import eigh

    ins = [i for i in inputs if not (isinstance(i, HostValue) and i.a.dtype == object)]
    fake = type("_N", (), {"params": {"lower": bool(node.params["lower"])}})
    return [eigh(fake, ins, env)[0]]


@handler("TensorInv")
def tensor_inv(node, inputs, env):
    from pytensor_amd.dispatch.lu import matrix_inverse

    a = env.to_device(inputs[0])
    ind = int(node.params["ind"])
    if ind <= 0:
        raise ValueError("Invalid ind argument.")
    old = tuple(a.shape)
    inv_shape = old[ind:] + old[:ind]
    prod = int(np.prod(old[ind:])) if old[ind:] else 1
    a2 = _reshape(a, (prod, a.size // prod if prod else 0))
    if a2.shape[0] != a2.shape[1]:
        raise np.linalg.LinAlgError("Last 2 dimensions of the array must be square")
    fake = type("_N", (), {"params": {}})
    (ia,) = matrix_inverse(fake, [a2], env)
    return [_reshape(ia, inv_shape)]


@handler("TensorSolve")
def tensor_solve(node, inputs, env):
    from pytensor_amd.dispatch.lu import solve_general

    a, b = _same_float(env, inputs, "TensorSolve")
    axes = node.params.get("axes")
    an = a.ndim
    if axes is not None:
        order = list(range(an))
        for k in axes:
            order.remove(k)
            order.insert(an, k)
        a = a.view([a.shape[d] for d in order], [a.strides[d] for d in order])
    old = tuple(a.shape[-(an - b.ndim):]) if an > b.ndim else ()
    prod = int(np.prod(old)) if old else 1
    if a.size != prod * prod:
        raise np.linalg.LinAlgError("Input arrays must satisfy the requirement prod(a.shape[b.ndim:]) == prod(a.shape[:b.ndim])")
    x = solve_general(env, _reshape(a, (prod, prod)), _reshape(b, (prod,)), 1)
    return [_reshape(x, old)]


@handler("BlockDiagonal")
def block_diagonal(node, inputs, env):
    from pytensor_amd.dispatch.elemwise import _cast

    out_dt = np.dtype(node.params["dtype"])
    mats = [env.to_device(i) for i in inputs]
    rows = sum(m.shape[0] for m in mats)
    cols = sum(m.shape[1] for m in mats)
    out = DeviceArray.empty((rows, cols), out_dt)
    if out.size:
        ffi.check(env.lib.pthip_memset(out.ptr, 0, out.nbytes))
    r0 = c0 = 0
    for m in mats:
        if m.size:
            src = m if np.dtype(m.dtype) == out_dt else _cast(env, m.contiguous(), out_dt)
            copy_into(out.view(m.shape, (cols, 1), r0 * cols + c0), src)
        r0 += m.shape[0]
        c0 += m.shape[1]
    return [out]


# ---- matrix exponential ---------------------------------------------------------------------------
_PADE13 = (64764752532480000.0, 32382376266240000.0, 7771770303897600.0, 1187353796428800.0, 129060195264000.0, 10559470521600.0,
           670442572800.0, 33522128640.0, 1323241920.0, 40840800.0, 960960.0, 16380.0, 182.0, 1.0)
_THETA13 = 5.371920351148152  # Higham (2005), Table 2.3: ||A||_1 up to which the [13/13] Pade approximant is exact to fp64


@handler("Expm")
def expm(node, inputs, env):
    """``Expm`` (linalg/products.py:13; perform = scipy.linalg.expm): scaling and squaring with the
    [13/13] Pade approximant (Higham 2005, the algorithm scipy's expm descends from) out of device
    GEMMs, fused linear combinations and one LU solve.  The scaling power is chosen from ||A||_1 (one
    host read of n column sums).  scipy additionally picks lower Pade orders for small norms; both are
    backward stable to unit roundoff."""
    from pytensor_amd.dispatch.blas import gemm_device
    from pytensor_amd.dispatch.lu import solve_general

    A = _matrix(env, inputs[0], "Expm")
    n = A.shape[0]
    if A.shape[1] != n:
        raise ValueError("expected a square matrix")
    if n == 0:
        return [DeviceArray.empty((0, 0), A.dtype)]
    dt = str(A.dtype)
    A = A.contiguous()
    absA = _ew1(env, [{"op": "Abs", "in": [["i", 0]], "dtype": dt}], [A], [dt], dt, (n, n))
    norm1 = float(np.asarray(env.to_host(_column_sums(env, absA))).max())
    if not np.isfinite(norm1):
        return [env.to_device(HostValue(np.full((n, n), np.nan, dtype=A.dtype)))]
    s = max(0, int(np.ceil(np.log2(norm1 / _THETA13)))) if norm1 > _THETA13 else 0
    if s:
        A = _ew1(env, [{"op": "Mul", "in": [["i", 0], ["c", float(2.0 ** -s).hex(), dt]], "dtype": dt}], [A], [dt], dt, (n, n))
    eye = DeviceArray.empty((n, n), A.dtype)
    ffi.check(env.lib.pthip_eye(ffi.np_dtype_code(A.dtype), n, n, 0, eye.ptr))
    b = _PADE13
    A2 = gemm_device(env, 1.0, A, A)
    A4 = gemm_device(env, 1.0, A2, A2)
    A6 = gemm_device(env, 1.0, A4, A2)

    def comb(coefs, mats):
        ops, acc = [], None
        for k, c in enumerate(coefs):
            ops.append({"op": "Mul", "in": [["i", k], ["c", float(c).hex(), dt]], "dtype": dt})
            if acc is not None:
                ops.append({"op": "Add", "in": [acc, ["t", len(ops) - 1]], "dtype": dt})
            acc = ["t", len(ops) - 1]
        return _ew1(env, ops, mats, [dt] * len(mats), dt, (n, n))

    W1 = comb((b[13], b[11], b[9]), [A6, A4, A2])
    W2 = comb((b[7], b[5], b[3], b[1]), [A6, A4, A2, eye])
    Z1 = comb((b[12], b[10], b[8]), [A6, A4, A2])
    Z2 = comb((b[6], b[4], b[2], b[0]), [A6, A4, A2, eye])
    W = gemm_device(env, 1.0, A6, W1, 1.0, W2)  # A6 W1 + W2
    U = gemm_device(env, 1.0, A, W)
    V = gemm_device(env, 1.0, A6, Z1, 1.0, Z2)
    P = comb((1.0, 1.0), [V, U])
    Q = comb((1.0, -1.0), [V, U])
    X = solve_general(env, Q, P, 2)
    for _ in range(s):
        X = gemm_device(env, 1.0, X, X)
    return [X]


def _ew1(env, body_ops, ins, in_dtypes, out_dtype, shape):
    from pytensor_amd.dispatch.elemwise import launch_elemwise

    body = {"in_dtypes": list(in_dtypes), "out_dtypes": [out_dtype], "body": body_ops, "outs": [["t", len(body_ops) - 1]]}
    (out,), _, _ = launch_elemwise(body, ins, tuple(shape), [out_dtype], None, env)
    return out


# ---- Sylvester equation -----------------------------------------------------------------------------
MAX_SYLVESTER = 4096  # m * n: the Kronecker system is (m n) x (m n)


@handler("SolveSylvester")
def solve_sylvester(node, inputs, env):
    """``A X + X B = C`` (linalg/solvers/linear_control.py:117 ``SolveSylvester``; the reference builds
    it from real Schur forms and LAPACK ``trsyl``).  Here the equation is solved as the linear system it
    is: (A (x) I_n + I_m (x) B^T) vec(X) = vec(C) with row-major vec, assembled by one generated kernel
    and handed to the LU solver — O((m n)^3), for the m n <= 4096 of state-space models; the unique
    solution whenever the spectra of A and -B are disjoint, like Bartels-Stewart's."""
    from pytensor_amd.dispatch.lu import solve_general

    A, B, Cm = _same_float(env, inputs, "SolveSylvester")
    m, n = A.shape[0], B.shape[0]
    if A.shape != (m, m) or B.shape != (n, n) or Cm.shape != (m, n):
        raise ValueError(f"SolveSylvester: incompatible shapes {A.shape}, {B.shape}, {Cm.shape}")
    if m * n == 0:
        return [DeviceArray.empty((m, n), A.dtype)]
    if m * n > MAX_SYLVESTER:
        raise NotImplementedError(f"hip linker: SolveSylvester with m*n = {m * n} (Kronecker tier, up to {MAX_SYLVESTER})")
    dt = str(A.dtype)
    eye = lambda k: env.to_device(HostValue(np.eye(k, dtype=A.dtype)))
    Im, In = eye(m), eye(n)
    shape = (m, n, m, n)  # K[(i, j), (k, l)] = A[i, k] d_jl + d_ik B[l, j]
    ops = [{"op": "Mul", "in": [["i", 0], ["i", 1]], "dtype": dt}, {"op": "Mul", "in": [["i", 2], ["i", 3]], "dtype": dt},
           {"op": "Add", "in": [["t", 0], ["t", 1]], "dtype": dt}]
    K = _ew1(env, ops, [A.view(shape, (A.strides[0], 0, A.strides[1], 0)), In.view(shape, (0, In.strides[0], 0, In.strides[1])),
                        Im.view(shape, (Im.strides[0], 0, Im.strides[1], 0)), B.view(shape, (0, B.strides[1], 0, B.strides[0]))],
             [dt] * 4, dt, shape)
    mn = m * n
    x = solve_general(env, K.view((mn, mn), (mn, 1)), _reshape(Cm, (mn,)), 1)
    return [x.view((m, n), (n, 1))]
