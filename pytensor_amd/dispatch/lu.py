"""General dense solves on LU with partial pivoting: ``Solve`` (gen / pos), ``Det``, ``SLogDet``,
``MatrixInverse`` and their ``Blockwise`` batching.

Reference: pytensor/tensor/linalg/solvers/general.py:17 ``Solve`` (perform 62-75:
``scipy.linalg.solve``; a singular system NaN-fills), linalg/summary.py:34 ``Det`` /
84 ``SLogDet`` (``np.linalg.det`` / ``slogdet``), linalg/inverse.py:87 ``MatrixInverse``
(``np.linalg.inv``: LinAlgError when singular).  SURVEY §8f row 3.
"""

from __future__ import annotations

import numpy as np

from pytensor_amd import ffi
from pytensor_amd.device import DeviceArray, contiguous_strides
from pytensor_amd.dispatch import handler
from pytensor_amd.dispatch.linalg import _batchify, _dt, _require_float, cho_solve_device, cholesky_device, trsm_device


def getrf_device(env, a: DeviceArray, flag_singular=False):
    """(LU, perm, sign, logabsdet) of a (..., n, n) array; batch dims flattened."""
    from pytensor_amd.dispatch.linalg import _lapack_operands

    (a,) = _lapack_operands(env, "LU", a)
    n = a.shape[-1]
    if a.shape[-2] != n:
        raise ValueError("expected a square matrix")
    bshape = a.shape[:-2]
    ab = _batchify(a, 2, bshape)
    nb = ab.shape[0]
    LU = DeviceArray.empty((nb, n, n), a.dtype)
    perm = DeviceArray.empty((nb, n), "int64")
    sign = DeviceArray.empty((nb,), a.dtype)
    logabs = DeviceArray.empty((nb,), a.dtype)
    if nb and n:
        ffi.check(env.lib.pthip_getrf(_dt(a), nb, n, ab.ptr, LU.ptr, perm.ptr, sign.ptr, logabs.ptr, int(flag_singular)))
    elif nb:  # 0 x 0: det = 1
        from pytensor_amd.executor import HostValue

        sign = env.to_device(HostValue(np.ones((nb,), dtype=a.dtype)))
        logabs = env.to_device(HostValue(np.zeros((nb,), dtype=a.dtype)))
    return LU, perm, sign, logabs, bshape


def _permute_rows(env, b: DeviceArray, perm: DeviceArray) -> DeviceArray:
    """b[perm] along axis 0 of a contiguous (n, nrhs) / (n,) array (P applied to the rhs)."""
    bc = b.contiguous()
    inner = int(np.prod(bc.shape[1:])) if bc.ndim > 1 else 1
    out = DeviceArray.empty(bc.shape, bc.dtype)
    if out.size:
        ffi.check(env.lib.pthip_take_rows(bc.itemsize, perm.size, inner, bc.ptr, bc.shape[0], inner, perm.ptr, out.ptr))
    return out


def solve_general(env, A: DeviceArray, b: DeviceArray, b_ndim: int) -> DeviceArray:
    """x = A^-1 b through P A = L U: permute b, unit-lower solve, upper solve.  A singular U has a
    zero pivot, which the triangular solve turns into the reference's NaN fill."""
    n = A.shape[-1]
    if b.shape[b.ndim - b_ndim] != n:
        raise ValueError(f"Solve: incompatible shapes {A.shape} and {b.shape}")
    if str(b.dtype) != str(A.dtype):
        raise TypeError("Solve: dtype mismatch")
    if A.ndim == 2 and b.ndim == b_ndim:
        LU, perm, _, _, _ = getrf_device(env, A)
        lu = LU.view((n, n), (n, 1))
        pb = _permute_rows(env, b, perm.view((n,), (1,)))
        y = trsm_device(env, lu, pb, True, True, b_ndim)
        return trsm_device(env, lu, y, False, False, b_ndim)
    # batched: loop the (small) batch on the host — each item is three launches
    bA, bb = A.shape[:-2], b.shape[: b.ndim - b_ndim]
    bshape = tuple(np.broadcast_shapes(bA, bb))
    core_b = b.shape[b.ndim - b_ndim :]
    Ab = _batchify(A, 2, bshape)
    bbm = _batchify(b, b_ndim, bshape)
    out = DeviceArray.empty((*bshape, *core_b), b.dtype)
    nb = Ab.shape[0]
    step = int(np.prod(core_b)) if core_b else 1
    for k in range(nb):
        Ak = Ab.view((n, n), (n, 1), k * n * n)
        bk = bbm.view(core_b, contiguous_strides(core_b), k * step)
        xk = solve_general(env, Ak, bk, b_ndim)
        from pytensor_amd.device import copy_into

        copy_into(out.view(core_b, contiguous_strides(core_b), k * step), xk)
    return out


def _solve(env, p, A, b):
    from pytensor_amd.dispatch.linalg import _lapack_operands

    A, b = _lapack_operands(env, "Solve", A, b)  # (integer / float16 operands: LAPACK's working type)
    assume = p["assume_a"]
    if assume == "pos":
        # scipy posv reads the triangle named by `lower` (default: upper)
        c = cholesky_device(env, A, bool(p["lower"]))
        return cho_solve_device(env, c, b, bool(p["lower"]), p["b_ndim"])
    if assume in ("gen", "sym", "her"):
        if assume != "gen":
            # scipy sysv reads only the triangle named by `lower`
            A = _symmetrize(env, A, bool(p["lower"]))
        return solve_general(env, A, b, p["b_ndim"])
    if assume == "tridiagonal":
        from pytensor_amd.dispatch.decomp import solve_tridiagonal

        return solve_tridiagonal(env, A, b, p["b_ndim"])
    raise NotImplementedError(f"hip linker: Solve(assume_a={assume!r}) is not lowered")


def lu_factor_device(env, a: DeviceArray):
    """``LUFactor`` (linalg/decomposition/lu.py:239; perform 279-299: scipy ``getrf``): the packed
    factors and LAPACK's 0-based interchange vector (int32); an exactly zero pivot NaN-fills LU.
    Leading dims are a batch (``Blockwise``).  What the reference's decomposition-reuse rewrites
    (rewriting/linalg/solvers.py:615-632) factor once for several ``Solve`` nodes."""
    LU, perm, _, _, bshape = getrf_device(env, a)
    nb, n = perm.shape
    piv = DeviceArray.empty((nb, n), "int32")
    if nb and n:
        ffi.check(env.lib.pthip_lu_factor_finish(_dt(a), nb, n, LU.ptr, perm.ptr, piv.ptr))
    return (LU.view((*bshape, n, n), contiguous_strides((*bshape, n, n))),
            piv.view((*bshape, n), contiguous_strides((*bshape, n))))


@handler("LUFactor")
def lu_factor(node, inputs, env):
    return list(lu_factor_device(env, env.to_device(inputs[0])))


@handler("PivotToPermutations")
def pivot_to_permutations(node, inputs, env):
    """lu.py:206-231: the permutation (or its inverse) LAPACK's sequential row interchanges amount to."""
    piv = env.to_device(inputs[0])
    if piv.dtype.kind not in "iu" or piv.itemsize not in (4, 8):
        raise TypeError(f"PivotToPermutations: integer pivots expected, got {piv.dtype}")
    n = piv.shape[-1]
    bshape = piv.shape[:-1]
    pc = piv.contiguous()
    nb = int(np.prod(bshape)) if bshape else 1
    out = DeviceArray.empty(piv.shape, "int64")
    if nb and n:
        ffi.check(env.lib.pthip_pivots_to_perm(pc.itemsize, int(bool(node.params["inverse"])), nb, n, pc.ptr, out.ptr))
    return [out]


@handler("Solve")
def solve(node, inputs, env):
    A, b = (env.to_device(i) for i in inputs)
    return [_solve(env, node.params, A, b)]


def _det(env, x):
    LU, perm, sign, logabs, bshape = getrf_device(env, x)
    return sign, logabs, bshape


@handler("Det")
def det(node, inputs, env):
    from pytensor_amd.dispatch.elemwise import launch_elemwise

    sign, logabs, bshape = _det(env, env.to_device(inputs[0]))
    dt = str(sign.dtype)
    # det = sign * exp(log|det|): how np.linalg.det finishes too (umath_linalg det_from_slogdet)
    body = {
        "in_dtypes": [dt, dt], "out_dtypes": [dt],
        "body": [{"op": "Exp", "in": [["i", 1]], "dtype": dt}, {"op": "Mul", "in": [["i", 0], ["t", 0]], "dtype": dt}],
        "outs": [["t", 1]],
    }
    outs, _, _ = launch_elemwise(body, [sign, logabs], sign.shape, [dt], None, env)
    return [outs[0].view(bshape, contiguous_strides(bshape))]


@handler("SLogDet")
def slogdet(node, inputs, env):
    sign, logabs, bshape = _det(env, env.to_device(inputs[0]))
    return [sign.view(bshape, contiguous_strides(bshape)), logabs.view(bshape, contiguous_strides(bshape))]


@handler("MatrixInverse")
def matrix_inverse(node, inputs, env):
    x = env.to_device(inputs[0])
    n = x.shape[-1]
    if x.ndim != 2:
        raise NotImplementedError("hip linker: batched MatrixInverse")
    LU, perm, _, _, _ = getrf_device(env, x, flag_singular=True)  # np.linalg.inv raises when singular
    lu = LU.view((n, n), (n, 1))
    pb = DeviceArray.empty((n, n), x.dtype)  # P * I, built on the device (no upload per call)
    if n:
        ffi.check(env.lib.pthip_permuted_identity(_dt(x), n, perm.ptr, pb.ptr))
    y = trsm_device(env, lu, pb, True, True, 2)
    return [trsm_device(env, lu, y, False, False, 2)]


@handler("Eigh")
def eigh(node, inputs, env):
    """``Eigh`` of the standard problem (linalg/decomposition/eigen.py:102; perform 177-195):
    eigenvalues ascending, eigenvectors as columns, the chosen triangle only (csrc/eigh.hip).
    Leading dims are a batch (``Blockwise``)."""
    if len(inputs) == 2:
        return _eigh_generalised(node, inputs, env)
    from pytensor_amd.dispatch.linalg import _lapack_operands

    (a,) = _lapack_operands(env, "Eigh", inputs[0])
    n = a.shape[-1]
    if a.shape[-2] != n:
        raise ValueError("Eigh: expected a square matrix")
    bshape = a.shape[:-2]
    ab = _batchify(a, 2, bshape)
    nb = ab.shape[0]
    w = DeviceArray.empty((nb, n), a.dtype)
    v = DeviceArray.empty((nb, n, n), a.dtype)
    if nb and n:
        ffi.check(env.lib.pthip_eigh(_dt(a), nb, n, int(node.params["lower"]), ab.ptr, w.ptr, v.ptr))
    return [w.view((*bshape, n), contiguous_strides((*bshape, n))), v.view((*bshape, n, n), contiguous_strides((*bshape, n, n)))]



def _symmetrize(env, x: DeviceArray, lower: bool) -> DeviceArray:
    n = x.shape[-1]
    bshape = x.shape[:-2]
    xb = _batchify(x, 2, bshape)
    out = DeviceArray.empty(xb.shape, x.dtype)
    if out.size:
        ffi.check(env.lib.pthip_symmetrize(_dt(x), xb.shape[0], n, int(lower), xb.ptr, out.ptr))
    return out.view((*bshape, n, n), contiguous_strides((*bshape, n, n)))


def _eigh_generalised(node, inputs, env):
    """``A v = w B v`` (Eigh with two inputs; perform = ``scipy.linalg.eigh(a, b, lower=)``, LAPACK
    sygvd): the same reduction LAPACK's ``sygst`` does, out of kernels that exist — ``B = L L^T``
    (potrf), ``C = L^-1 A L^-T`` (two multi-rhs triangular solves), the standard problem for C
    (Jacobi), ``v = L^-T y``.  Eigenvectors come out B-orthonormal (``v^T B v = I``) like scipy's."""
    from pytensor_amd.dispatch.linalg import _lapack_operands

    a, b = _lapack_operands(env, "Eigh", *inputs)
    n = a.shape[-1]
    if a.shape[-2] != n or b.shape[-2:] != (n, n):
        raise ValueError(f"Eigh: incompatible shapes {a.shape} and {b.shape}")
    if a.ndim != 2 or b.ndim != 2:
        raise NotImplementedError("hip linker: batched generalised Eigh")
    lower = bool(node.params["lower"])
    A = _symmetrize(env, a, lower)
    B = _symmetrize(env, b, lower)
    L = cholesky_device(env, B, True)
    Y = trsm_device(env, L, A, True, False, 2)  # L Y = A
    Yt = Y.view((n, n), (Y.strides[1], Y.strides[0])).contiguous()
    Cm = trsm_device(env, L, Yt, True, False, 2)  # L C = Y^T  ->  C = L^-1 A L^-T (symmetric)
    fake = type("_N", (), {"params": {"lower": True}})
    w, y = eigh(fake, [Cm], env)
    v = trsm_device(env, L, y, True, False, 2, trans=True)  # L^T v = y
    return [w, v]
