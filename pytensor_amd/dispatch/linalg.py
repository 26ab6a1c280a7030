"""``Cholesky`` / ``SolveTriangular`` / ``CholeskySolve`` and their ``Blockwise`` batching.

Reference: pytensor/tensor/linalg/decomposition/cholesky.py:18-83,
solvers/triangular.py:13-71, solvers/psd.py:14-53, pytensor/tensor/blockwise.py:153
(batch dims → grid dimension here instead of a Python loop).
"""

from __future__ import annotations

import numpy as np

from pytensor_amd import ffi
from pytensor_amd.device import DeviceArray, copy_into
from pytensor_amd.dispatch import handler
from pytensor_amd.executor import HostValue


def _dt(x):
    return ffi.np_dtype_code(x.dtype)


def _require_float(x, what):
    if x.dtype.kind != "f" or x.dtype.itemsize < 4:
        raise NotImplementedError(f"{what}: only float32/float64 on the device")


def _lapack_dtype(dt) -> np.dtype:
    """The working dtype LAPACK would use for an operand of dtype ``dt``
    (pytensor/tensor/linalg/dtype_utils.py:9-26, i.e. scipy's ``find_best_lapack_type``): float64 for
    doubles and integers wider than 16 bits, float32 for everything narrower."""
    dt = np.dtype(dt)
    if dt.kind == "c":
        raise NotImplementedError("complex dtypes are not lowered by the hip linker")
    if (dt.kind == "f" and dt.itemsize > 4) or (dt.kind in "ibu" and dt.itemsize > 2):
        return np.dtype("float64")
    return np.dtype("float32")


def _lapack_operands(env, what, *vals):
    """Operands of a LAPACK-family op on the device, promoted as the reference promotes them: each to
    its LAPACK type, then to their common type (integer, bool and float16 matrices are legal inputs
    of ``cholesky`` / ``solve`` / ``lstsq``; the reference's test_solve_dtype walks all pairs)."""
    from pytensor_amd.dispatch.elemwise import _cast

    xs = [env.to_device(v) for v in vals]
    work = np.result_type(*(_lapack_dtype(x.dtype) for x in xs))
    return [x if np.dtype(x.dtype) == work else _cast(env, x.contiguous(), work) for x in xs]


def _batchify(x: DeviceArray, core_ndim: int, bshape):
    """Broadcast leading dims to ``bshape`` and return a contiguous (batch, *core) array."""
    core = x.shape[x.ndim - core_ndim :]
    lead = x.shape[: x.ndim - core_ndim]
    lead = (1,) * (len(bshape) - len(lead)) + lead
    nb = int(np.prod(bshape)) if bshape else 1
    if lead == tuple(bshape) and x.is_contiguous():
        return x.view((nb, *core), _cs((nb, *core)))
    out = DeviceArray.empty((*bshape, *core), x.dtype)
    copy_into(out, x.view((*lead, *core), (0,) * (len(lead) - (x.ndim - core_ndim)) + x.strides))
    return out.view((nb, *core), _cs((nb, *core)))


def _cs(shape):
    st, acc = [], 1
    for s in reversed(shape):
        st.append(acc)
        acc *= max(int(s), 1)
    return tuple(reversed(st))


def cholesky_device(env, a: DeviceArray, lower: bool) -> DeviceArray:
    (a,) = _lapack_operands(env, "Cholesky", a)
    if a.shape[-1] != a.shape[-2]:
        raise ValueError("Cholesky: matrix must be square")
    n = a.shape[-1]
    bshape = a.shape[:-2]
    ab = _batchify(a, 2, bshape)
    out = DeviceArray.empty(a.shape, a.dtype)
    if out.size:
        ffi.check(env.lib.pthip_potrf(_dt(a), int(lower), ab.shape[0], n, ab.ptr, out.ptr))
    return out


def trsm_device(env, T: DeviceArray, b: DeviceArray, lower, unit, b_ndim, trans=False) -> DeviceArray:
    T, b = _lapack_operands(env, "SolveTriangular", T, b)
    n = T.shape[-1]
    if T.shape[-2] != n:
        raise ValueError("SolveTriangular: matrix must be square")
    if b.shape[b.ndim - b_ndim] != n:
        raise ValueError(f"SolveTriangular: incompatible shapes {T.shape} and {b.shape}")
    bT, bb = T.shape[:-2], b.shape[: b.ndim - b_ndim]
    bshape = tuple(np.broadcast_shapes(bT, bb))
    nb = int(np.prod(bshape)) if bshape else 1
    nrhs = 1 if b_ndim == 1 else b.shape[-1]
    core_b = b.shape[b.ndim - b_ndim :]
    out = DeviceArray.empty((*bshape, *core_b), b.dtype)
    if out.size == 0:
        return out
    if not bshape:
        Tm, sTb, sT0, sT1 = T, 0, T.strides[0], T.strides[1]
        if any(s < 0 for s in T.strides):
            Tm = T.contiguous()
            sT0, sT1 = Tm.strides
        bm = b.contiguous()
        sBb = 0
    else:
        Tm = _batchify(T, 2, bshape)
        sTb, sT0, sT1 = Tm.strides
        bm = _batchify(b, b_ndim, bshape)
        sBb = bm.strides[0]
    ffi.check(
        env.lib.pthip_trsm(_dt(T), int(lower), int(trans), int(unit), nb, n, nrhs, Tm.ptr, sTb, sT0, sT1, bm.ptr, sBb, out.ptr)
    )
    return out


@handler("Cholesky")
def cholesky(node, inputs, env):
    return [cholesky_device(env, env.to_device(inputs[0]), node.params["lower"])]


@handler("CholeskyTrsv")
def cholesky_trsv(node, inputs, env):
    """``L = cholesky(S, lower); x = L^-1 b`` in one launch (fusion.fuse_cholesky_solve); falls
    back to the two separate kernels when the matrix does not fit the LDS-resident one."""
    S, b = _lapack_operands(env, "Cholesky", *inputs)
    n = S.shape[-1]
    if S.shape[0] != n:
        raise ValueError("Cholesky: matrix must be square")
    if b.shape != (n,):
        raise ValueError(f"SolveTriangular: incompatible shapes {S.shape} and {b.shape}")
    fits = 0 < n <= 256 and n * (n | 1) * S.itemsize <= 160 * 1024 - 2560 and str(b.dtype) == str(S.dtype)
    if not fits:
        L = cholesky_device(env, S, True)
        return [L, trsm_device(env, L, b, True, False, 1)]
    Sc, bc = S.contiguous(), b.contiguous()
    L = DeviceArray.empty((n, n), S.dtype)
    x = DeviceArray.empty((n,), S.dtype)
    ffi.check(env.lib.pthip_potrf_trsv(_dt(S), 1, n, Sc.ptr, bc.ptr, L.ptr, x.ptr))
    return [L, x]


@handler("SolveTriangular")
def solve_triangular(node, inputs, env):
    p = node.params
    A, b = (env.to_device(i) for i in inputs)
    return [trsm_device(env, A, b, p["lower"], p["unit_diagonal"], p["b_ndim"])]


def cho_solve_device(env, c, b, lower, b_ndim):
    # potrs (psd.py:35-53): lower: L y = b, L^T x = y ; upper: U^T y = b, U x = y
    y = trsm_device(env, c, b, lower, False, b_ndim, trans=not lower)
    return trsm_device(env, c, y, lower, False, b_ndim, trans=lower)


@handler("CholeskySolve")
def cholesky_solve(node, inputs, env):
    c, b = (env.to_device(i) for i in inputs)
    return [cho_solve_device(env, c, b, node.params["lower"], node.params["b_ndim"])]


@handler("Blockwise")
def blockwise(node, inputs, env):
    p = node.params
    cp = p["core_params"]
    ins = [env.to_device(i) for i in inputs]
    if p["core_op"] == "Cholesky":
        return [cholesky_device(env, ins[0], cp["lower"])]
    if p["core_op"] == "SolveTriangular":
        return [trsm_device(env, ins[0], ins[1], cp["lower"], cp["unit_diagonal"], cp["b_ndim"])]
    if p["core_op"] == "CholeskySolve":
        return [cho_solve_device(env, ins[0], ins[1], cp["lower"], cp["b_ndim"])]
    if p["core_op"] in ("Solve", "Det", "SLogDet"):
        from pytensor_amd.dispatch import lu

        if p["core_op"] == "Solve":
            return [lu._solve(env, cp, ins[0], ins[1])]
        fake = type("_N", (), {"params": cp})
        return (lu.det if p["core_op"] == "Det" else lu.slogdet)(fake, ins, env)
    if p["core_op"] == "LUFactor":
        from pytensor_amd.dispatch import lu

        return list(lu.lu_factor_device(env, ins[0]))
    if p["core_op"] == "PivotToPermutations":
        from pytensor_amd.dispatch import lu

        return lu.pivot_to_permutations(type("_N", (), {"params": cp}), ins, env)
    if p["core_op"] == "Eigh" and len(ins) == 1:  # (the generalised problem loops its items below)
        from pytensor_amd.dispatch import lu

        return lu.eigh(type("_N", (), {"params": cp}), ins, env)
    return _blockwise_loop(node, ins, env, inputs)



def _blockwise_loop(node, ins, env, raw=None):
    """Any other core op with a device handler: loop the broadcast batch on the host, one core
    call per item on views of the operands (``Blockwise.perform``, pytensor/tensor/blockwise.py:
    542: gufunc semantics).  What ``vectorize`` makes of small helper ops — the row gather
    ``b[perm]`` and the pivot bookkeeping of a batched ``lu_solve`` — not a hot path."""
    from pytensor_amd.device import copy_into
    from pytensor_amd.dispatch import HANDLERS

    p = node.params
    core = HANDLERS.get(p["core_op"])
    if p["core_op"] == "__inline__":
        # an inlined OpFromGraph core: run its lowered inner graph per item (like a Scan step)
        from pytensor_amd.executor import HipExecutable

        # (the executable rides on the inner graph object itself: it lives exactly as long as the node that owns the
        #  graph — a process-wide dict keyed by id() kept every one alive for good)
        ig = p["core_params"]["inner"]
        inner = getattr(ig, "_pthip_exe", None)
        if inner is None:
            inner = HipExecutable(ig, device=env.exe._device, tail=False)
            ig._pthip_exe = inner

        def core(_node, item, _env):
            return inner.run_device(item, _env)[0]

    if core is None:
        raise NotImplementedError(f"Blockwise({p['core_op']})")
    sig_in = p["signature"].split("->")[0]
    core_ndims = [t.count(",") + 1 if t.strip("()") else 0 for t in sig_in.split("),(")]
    if len(core_ndims) != len(ins):
        raise NotImplementedError(f"Blockwise({p['core_op']}): signature {p['signature']!r} does not match {len(ins)} operands")
    batch_shapes = [a.shape[: a.ndim - c] for a, c in zip(ins, core_ndims)]
    bshape = tuple(np.broadcast_shapes(*batch_shapes))
    views = []
    for a, c in zip(ins, core_ndims):
        lead = a.shape[: a.ndim - c]
        pad = len(bshape) - len(lead)
        strides = (0,) * pad + tuple(0 if (s == 1 and bshape[pad + d] != 1) else st for d, (s, st) in enumerate(zip(lead, a.strides[: len(lead)])))
        views.append((a, strides, a.shape[a.ndim - c :], a.strides[a.ndim - c :]))
    fake = type("_Core", (), {"params": p["core_params"], "inputs": list(node.inputs), "outputs": list(node.outputs)})
    outs = None
    saved = env.donated
    env.donated = frozenset()
    try:
        for idx in np.ndindex(*bshape):
            item = [a.view(cs, cst, sum(i * st for i, st in zip(idx, bst))) for a, bst, cs, cst in views]
            if raw is not None:
                # an unbatched host-known operand (the step count of a vectorised Scan) stays on the host: the core
                # handler reads it without a device round trip
                item = [r if (isinstance(r, HostValue) and r.a.ndim == c) else it for it, r, c in zip(item, raw, core_ndims)]
            rs = [env.to_device(r) for r in core(fake, item, env)]
            if outs is None:
                outs = [DeviceArray.empty((*bshape, *r.shape), r.dtype) for r in rs]
            for o, r in zip(outs, rs):
                core_shape = o.shape[len(bshape):]
                copy_into(o.view(core_shape, o.strides[len(bshape):], sum(i * st for i, st in zip(idx, o.strides[: len(bshape)]))), r)
    finally:
        env.donated = saved
    if outs is None:
        raise NotImplementedError(f"Blockwise({p['core_op']}) over an empty batch")
    return outs
