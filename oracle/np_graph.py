"""CPU oracle — TEST INFRASTRUCTURE ONLY.

A NumPy/SciPy restatement of the reference's *Python* (``Op.perform``) semantics
for every op on the hot path, interpreting the portable IR
(``pytensor_amd/ir.py``).  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this module; the product path
(``pytensor_amd.executor``) never does.

Pinned against the reference itself: ``tests/golden/*.npz`` hold outputs of the
reference C linker (``mode="CVM"``) and NumPy linker (``Mode("py")``) produced in
the build container by ``oracle/make_golden.py`` from ``/root/reference``;
``tests/test_oracle.py`` checks this interpreter against them.

Third-party arithmetic that is *not* under ``/root/reference``:
BLAS ``gemm/gemv`` and LAPACK ``potrf/trtrs/potrs`` come from the NumPy/SciPy
wheels (OpenBLAS 0.3.29 here; the reference pins only ``numpy>=2.0``,
``scipy>=1,<2`` in pyproject.toml:49-55).  We call the same entry points the
reference's ``perform`` methods call.

Every handler cites the reference ``perform`` it follows.
"""

from __future__ import annotations

import numpy as np
import scipy.linalg
import scipy.special

import special_c

# ---------------------------------------------------------------------------
# scalar ops — reference: pytensor/scalar/basic.py, pytensor/scalar/math.py
# (``impl`` / ``nfunc_spec`` of each ScalarOp = what the NumPy linker runs)
# ---------------------------------------------------------------------------


def _softplus(x):
    # scalar/math.py:1227-1245 (Softplus.impl), vectorised
    x = np.asarray(x)
    with np.errstate(over="ignore", invalid="ignore"):
        return np.where(
            x < -37.0,
            np.exp(x),
            np.where(x < 18.0, np.log1p(np.exp(x)), np.where(x < 33.3, x + np.exp(-x), x)),
        )


def _log1mexp(x):
    # scalar/math.py:1295-1340 (Log1mexp.impl)
    x = np.asarray(x)
    with np.errstate(all="ignore"):
        return np.where(x < np.log(0.5), np.log1p(-np.exp(x)), np.log(-np.expm1(x)))


def _sign(x):
    return np.sign(x)


def _psi_as103(x):
    """Digamma as the reference's *C* backend computes it (scalar/math.py:427-487, Bernardo's
    AS 103 with 10-digit constants).  ``Psi.impl`` is ``scipy.special.psi``; the two differ by
    up to ~4e-9 relative, and the golden vectors hold the C linker's values (the reference's
    default runtime), so this is the restatement the device code is checked against."""
    x = np.asarray(x, dtype=np.float64)
    S, C, S3, S4, S5, D1 = 1.0e-5, 8.5, 8.333333333e-2, 8.333333333e-3, 3.968253968e-3, -0.5772156649

    def pos(y):
        y = np.array(y, dtype=np.float64, copy=True)
        res = np.zeros_like(y)
        small = y <= S
        acc = np.zeros_like(y)
        yy = y.copy()
        for _ in range(10):  # while (y < C): at most 9 unit steps from y > 0
            m = (yy < C) & ~small
            acc = np.where(m, acc - 1.0 / np.where(m, yy, 1.0), acc)
            yy = np.where(m, yy + 1.0, yy)
        R = 1.0 / np.where(small, 1.0, yy)
        v = acc + np.log(np.where(small, 1.0, yy)) - 0.5 * R
        R2 = R * R
        v = v - R2 * (S3 - R2 * (S4 - R2 * S5))
        res = np.where(small, D1 - 1.0 / np.where(small, y, 1.0), v)
        return res

    with np.errstate(all="ignore"):
        neg = x <= 0
        out = pos(np.where(neg, 1.0, x))
        if np.any(neg):
            xn = np.where(neg, x, -0.5)
            refl = pos(1.0 - xn) - np.pi * (np.cos(np.pi * xn) / np.sin(np.pi * xn))
            refl = np.where(xn == np.floor(xn), np.inf, refl)
            out = np.where(neg, refl, out)
    return out


def _int_div(x, y):
    # scalar/basic.py IntDiv.impl: x // y
    with np.errstate(all="ignore"):
        return np.floor_divide(x, y)


def _mod(x, y):
    with np.errstate(all="ignore"):
        return np.mod(x, y)


def _variadic(f):
    def g(*a):
        r = a[0]
        for b in a[1:]:
            r = f(r, b)
        return r

    return g


def _switch(c, a, b):
    return np.where(c, a, b)


def _clip(x, lo, hi):
    # scalar/basic.py:2335 Clip.impl: min/max chain
    return np.where(x < lo, lo, np.where(x > hi, hi, x))


def _round_half_to_even(x):
    return np.round(x)


def _round_half_away(x):
    # scalar/basic.py RoundHalfAwayFromZero.impl
    return np.where(x < 0, -np.floor(0.5 - x), np.floor(0.5 + x))


SCALAR = {
    "Add": _variadic(np.add),
    "Mul": _variadic(np.multiply),
    "Sub": np.subtract,
    "TrueDiv": np.true_divide,
    "IntDiv": _int_div,
    "Mod": _mod,
    "Pow": np.power,
    "Neg": np.negative,
    "Abs": np.abs,
    "Sign": _sign,
    "Sgn": _sign,
    "Sqr": np.square,
    "Sqrt": np.sqrt,
    "Exp": np.exp,
    "Exp2": np.exp2,
    "Expm1": np.expm1,
    "Log": np.log,
    "Log2": np.log2,
    "Log10": np.log10,
    "Log1p": np.log1p,
    "Sin": np.sin,
    "Cos": np.cos,
    "Tan": np.tan,
    "ArcSin": np.arcsin,
    "ArcCos": np.arccos,
    "ArcTan": np.arctan,
    "ArcTan2": np.arctan2,
    "Sinh": np.sinh,
    "Cosh": np.cosh,
    "Tanh": np.tanh,
    "ArcSinh": np.arcsinh,
    "ArcCosh": np.arccosh,
    "ArcTanh": np.arctanh,
    "Sigmoid": scipy.special.expit,
    "Softplus": _softplus,
    "Log1mexp": _log1mexp,
    "Erf": scipy.special.erf,
    "Erfc": scipy.special.erfc,
    "Erfinv": scipy.special.erfinv,
    "Erfcinv": scipy.special.erfcinv,
    "Erfcx": scipy.special.erfcx,
    "GammaLn": scipy.special.gammaln,
    "Gamma": scipy.special.gamma,
    "Psi": _psi_as103,
    "TriGamma": special_c.TriGamma,
    "GammaInc": special_c.GammaInc,
    "GammaIncC": special_c.GammaIncC,
    "BetaInc": special_c.BetaInc,
    # (these four have no C code in the reference: both linkers evaluate them through SciPy)
    "PolyGamma": scipy.special.polygamma,  # scalar/math.py:607
    "NdtriExp": scipy.special.ndtri_exp,  # scalar/math.py:281
    "GammaIncInv": scipy.special.gammaincinv,  # scalar/math.py:728
    "GammaIncCInv": scipy.special.gammainccinv,  # scalar/math.py:753
    "BetaIncInv": scipy.special.betaincinv,  # scalar/math.py:1608
    "J0": scipy.special.j0,
    "J1": scipy.special.j1,
    "I0": scipy.special.i0,
    "I1": scipy.special.i1,
    "Reciprocal": np.reciprocal,
    "Maximum": _variadic(np.maximum),
    "Minimum": _variadic(np.minimum),
    "ScalarMaximum": _variadic(np.maximum),
    "ScalarMinimum": _variadic(np.minimum),
    "EQ": np.equal,
    "NEQ": np.not_equal,
    "LT": np.less,
    "GT": np.greater,
    "LE": np.less_equal,
    "GE": np.greater_equal,
    "AND": _variadic(np.bitwise_and),
    "OR": _variadic(np.bitwise_or),
    "XOR": _variadic(np.bitwise_xor),
    "Invert": np.invert,
    "IsNan": np.isnan,
    "IsInf": np.isinf,
    "Switch": _switch,
    "Clip": _clip,
    "Identity": lambda x: x,
    "Second": lambda a, b: np.broadcast_to(b, np.broadcast(a, b).shape),
    "Floor": np.floor,
    "Ceil": np.ceil,
    "Trunc": np.trunc,
    "RoundHalfToEven": _round_half_to_even,
    "RoundHalfAwayFromZero": _round_half_away,
    "Cast": lambda x: x,  # the dtype conversion is applied to every node below
    "Deg2Rad": np.deg2rad,
    "Rad2Deg": np.rad2deg,
}


def eval_scalar_body(body: dict, inputs):
    """Evaluate a lowered scalar graph on (broadcastable) arrays."""
    vals = []

    def get(r):
        if r[0] == "i":
            return inputs[r[1]]
        if r[0] == "t":
            return vals[r[1]]
        dt = np.dtype(r[2])
        v = float.fromhex(r[1]) if dt.kind == "f" else r[1]
        return np.asarray(v, dtype=dt)[()]

    with np.errstate(all="ignore"):
        for n in body["body"]:
            args = [get(r) for r in n["in"]]
            if n["op"] == "ScalarLoop":
                vals.append(_eval_scalar_loop(n["loop"], args))  # a tuple of final states (+ until)
                continue
            if n["op"] == "LoopOut":
                vals.append(np.asarray(args[0][n["k"]]).astype(n["dtype"], copy=False))
                continue
            out = SCALAR[n["op"]](*args)
            vals.append(np.asarray(out).astype(n["dtype"], copy=False))
    return [get(r) for r in body["outs"]]


def _eval_scalar_loop(loop: dict, args):
    """``ScalarLoop`` the way its C code runs it (pytensor/scalar/loop.py:181-290), lane-wise over
    arrays: each element iterates ``n_steps`` times or until its ``until`` turns true (the flag
    starts true, so zero steps report "done"); finished elements keep their state."""
    inner = loop["body"]
    S = loop["n_state"]
    n_steps = np.asarray(args[0])
    shape = np.broadcast_shapes(*[np.shape(a) for a in args])
    state = [np.broadcast_to(np.asarray(a).astype(dt, copy=False), shape).copy() for a, dt in zip(args[1 : 1 + S], inner["in_dtypes"][:S])]
    consts = [np.broadcast_to(np.asarray(a).astype(dt, copy=False), shape) for a, dt in zip(args[1 + S :], inner["in_dtypes"][S:])]
    done = np.ones(shape, dtype=bool)
    active = np.broadcast_to(n_steps > 0, shape).copy()
    it = 0
    while active.any():
        outs = eval_scalar_body(inner, state + consts)
        for j in range(S):
            state[j] = np.where(active, np.broadcast_to(outs[j], shape), state[j]).astype(inner["in_dtypes"][j], copy=False)
        if loop["is_while"]:
            done = np.where(active, np.broadcast_to(outs[S], shape).astype(bool), done)
        else:
            done = np.where(active, False, done)
        it += 1
        active = active & (it < n_steps)
        if loop["is_while"]:
            active = active & ~done
    return tuple(state) + ((done,) if loop["is_while"] else ())


# ---------------------------------------------------------------------------
# tensor op handlers:  fn(params, inputs, node, graph) -> list of outputs
# ---------------------------------------------------------------------------

OPS = {}


def op(name):
    def deco(f):
        OPS[name] = f
        return f

    return deco


def _check_runtime_broadcast(node, graph, inputs):
    # pytensor/tensor/elemwise.py:825-840: a dim that is not *statically*
    # broadcastable may not be broadcast at run time.
    ndim = max((np.ndim(i) for i in inputs), default=0)
    for d in range(ndim):
        lens = [np.shape(i)[d] for i in inputs if np.ndim(i) == ndim]
        if len(set(lens)) > 1:
            for vid, arr in zip(node.inputs, inputs):
                if np.ndim(arr) != ndim:
                    continue
                static = graph.vars[vid].shape
                if arr.shape[d] == 1 and (len(static) <= d or static[d] != 1):
                    raise ValueError(
                        f"Runtime broadcasting not allowed. Input has shape 1 along dimension {d}, "
                        "but the static type does not declare it broadcastable"
                    )


@op("Elemwise")
def _elemwise(p, inputs, node, graph):
    # pytensor/tensor/elemwise.py:755-823 (Elemwise.perform)
    inputs = _sum_partial_inputs(p, inputs)
    if not p.get("partial_inputs") and not p.get("gather"):
        _check_runtime_broadcast(node, graph, inputs)
    outs = eval_scalar_body(p["scalar"], inputs)
    shape = np.broadcast(*inputs).shape if inputs else ()
    res = []
    for o, vid in zip(outs, node.outputs):
        dt = graph.vars[vid].dtype
        res.append(np.array(np.broadcast_to(o, shape), dtype=dt, order="C"))
    return res


_REDUCE = {
    "Add": np.add,
    "Mul": np.multiply,
    "Maximum": np.maximum,
    "Minimum": np.minimum,
    "ScalarMaximum": np.maximum,
    "ScalarMinimum": np.minimum,
    "AND": np.bitwise_and,
    "OR": np.bitwise_or,
    "XOR": np.bitwise_xor,
}


@op("CAReduce")
def _careduce(p, inputs, node, graph):
    # pytensor/tensor/elemwise.py:1493-1511 (CAReduce.perform): ufunc.reduce
    # over the axes with ``dtype=acc_dtype`` and a final cast to ``dtype``.
    (x,) = inputs
    axis = tuple(p["axis"])
    acc = np.dtype(p["acc_dtype"])
    if p["scalar_op"] == "MulWithoutZeros":
        # pytensor/tensor/math.py:3786-3825 (ProdWithoutZeros: y if x == 0, x if y == 0, else x * y; identity 0): the product
        # of the non-zero entries, 0 where there is none
        xa = np.asarray(x).astype(acc)
        r = np.where((xa != 0).any(axis=axis), np.multiply.reduce(np.where(xa == 0, acc.type(1), xa), axis=axis, dtype=acc), acc.type(0))
        return [np.asarray(r).astype(p["dtype"], copy=False)]
    uf = _REDUCE[p["scalar_op"]]
    if x.dtype.kind == "b" and p["scalar_op"] in ("AND", "OR", "XOR"):
        r = uf.reduce(x, axis=axis)
    else:
        # (NumPy itself raises "zero-size array to reduction operation ... which has no identity" exactly when a REDUCED
        #  axis is empty and the ufunc has no identity; an empty KEPT axis just gives an empty result)
        r = uf.reduce(x, axis=axis, dtype=acc)
    return [np.asarray(r).astype(p["dtype"], copy=False)]


@op("DimShuffle")
def _dimshuffle(p, inputs, node, graph):
    # pytensor/tensor/elemwise.py:301-320 (DimShuffle.perform): transpose +
    # reshape; dropped dims must have length 1.
    (x,) = inputs
    x = np.asarray(x)
    order = p["new_order"]
    keep = [o for o in order if o != "x"]
    drop = [d for d in range(x.ndim) if d not in keep]
    for d in drop:
        if x.shape[d] != 1:
            raise ValueError("cannot drop a non-broadcastable dimension")
    t = x.transpose(keep + drop)
    shape = []
    k = 0
    for o in order:
        if o == "x":
            shape.append(1)
        else:
            shape.append(t.shape[k])
            k += 1
    return [t.reshape(shape)]


@op("Dot22")
def _dot22(p, inputs, node, graph):
    # pytensor/tensor/blas/gemm.py:274 (Dot22.perform): np.dot
    x, y = inputs
    return [np.dot(x, y)]


@op("Dot22Scalar")
def _dot22s(p, inputs, node, graph):
    # pytensor/tensor/blas/gemm.py:298+ (Dot22Scalar.perform): scalar * np.dot
    x, y, a = inputs
    return [np.asarray(a * np.dot(x, y))]


@op("Dot")
def _dot(p, inputs, node, graph):
    # pytensor/tensor/math.py:3041+ (Dot.perform)
    x, y = inputs
    return [np.asarray(np.dot(x, y))]


@op("Gemm")
def _gemm(p, inputs, node, graph):
    # pytensor/tensor/blas/gemm.py:183-216 (Gemm.perform): z <- b*z + a*dot(x,y);
    # z broadcast along length-1 dims (194-198).
    z, a, x, y, b = inputs
    xy = np.dot(x, y)
    if z.shape != xy.shape:
        z = np.broadcast_to(z, xy.shape)
    if b == 0.0:
        out = a * xy if a != 1.0 else xy
    else:
        out = b * z + a * xy
    return [np.asarray(out, dtype=z.dtype)]


@op("Gemv")
def _gemv(p, inputs, node, graph):
    # pytensor/tensor/blas/gemv.py:64-108 (Gemv.perform): y <- beta*y + alpha*dot(A,x);
    # beta == 0 => y is not read (79-86).
    y, alpha, A, x, beta = inputs
    if beta == 0.0:
        out = alpha * np.dot(A, x)
    else:
        out = beta * y + alpha * np.dot(A, x)
    return [np.asarray(out, dtype=y.dtype)]


@op("Ger")
def _ger(p, inputs, node, graph):
    # pytensor/tensor/blas/ger.py (Ger.perform): A + alpha*outer(x,y)
    A, alpha, x, y = inputs
    return [np.asarray(A + alpha * np.outer(x, y), dtype=A.dtype)]


@op("BatchedDot")
def _bdot(p, inputs, node, graph):
    # pytensor/tensor/blas/batched.py:69-79 (BatchedDot.perform): np.matmul,
    # batch sizes must match exactly (44-61)
    x, y = inputs
    if x.shape[0] != y.shape[0]:
        raise TypeError(f"Inputs {x.shape}, {y.shape} must have the same size in axis 0")
    return [np.matmul(x, y)]


@op("Cholesky")
def _cholesky(p, inputs, node, graph):
    # pytensor/tensor/linalg/decomposition/cholesky.py:48-83: LAPACK potrf,
    # clean=True zeros the other triangle, info != 0 => NaN-filled result.
    (a,) = inputs
    if a.size == 0:
        return [np.empty_like(a)]
    (potrf,) = scipy.linalg.get_lapack_funcs(("potrf",), (a,))
    c, info = potrf(a, lower=p["lower"], overwrite_a=False, clean=True)
    if info != 0:
        c = np.full(a.shape, np.nan, dtype=a.dtype)
    return [np.asarray(c, dtype=a.dtype)]


@op("SolveTriangular")
def _solve_tri(p, inputs, node, graph):
    # pytensor/tensor/linalg/solvers/triangular.py:32-71: LAPACK trtrs; NaN on info != 0
    A, b = inputs
    if A.size == 0 or b.size == 0:
        return [np.empty_like(b)]
    (trtrs,) = scipy.linalg.get_lapack_funcs(("trtrs",), (A, b))
    x, info = trtrs(A, b, lower=p["lower"], trans=0, unitdiag=p["unit_diagonal"])
    if info != 0:
        x = np.full(b.shape, np.nan, dtype=x.dtype)
    return [np.asarray(x)]


@op("CholeskySolve")
def _cho_solve(p, inputs, node, graph):
    # pytensor/tensor/linalg/solvers/psd.py:35-53: LAPACK potrs
    c, b = inputs
    if c.size == 0 or b.size == 0:
        return [np.empty_like(b)]
    (potrs,) = scipy.linalg.get_lapack_funcs(("potrs",), (c, b))
    x, info = potrs(c, b, lower=p["lower"])
    if info != 0:
        x = np.full(b.shape, np.nan, dtype=x.dtype)
    return [np.asarray(x)]


@op("Blockwise")
def _blockwise(p, inputs, node, graph):
    # pytensor/tensor/blockwise.py:542 (Blockwise.perform): loop the core op
    # over broadcast leading batch dims (gufunc semantics).
    core = OPS[p["core_op"]]
    sig_in = p["signature"].split("->")[0]
    core_ndims = [s.count(",") + 1 if s.strip("()") else 0 for s in sig_in.split("),(")]
    batch_shapes = [np.shape(i)[: np.ndim(i) - c] for i, c in zip(inputs, core_ndims)]
    bshape = np.broadcast_shapes(*batch_shapes)
    bcast = [
        np.broadcast_to(i, bshape + np.shape(i)[np.ndim(i) - c :]) for i, c in zip(inputs, core_ndims)
    ]
    outs = None
    for idx in np.ndindex(*bshape):
        rs = [np.asarray(r) for r in core(p["core_params"], [b[idx] for b in bcast], node, graph)]
        if outs is None:
            outs = [np.empty(bshape + r.shape, dtype=r.dtype) for r in rs]
        for o, r in zip(outs, rs):
            o[idx] = r
    if outs is None:  # empty batch
        dt = graph.vars[node.outputs[0]].dtype
        core_shape = np.shape(inputs[-1])[np.ndim(inputs[-1]) - core_ndims[-1] :]
        outs = [np.empty(bshape + core_shape, dtype=dt)]
    return outs


def unflatten_index(idx_list, index_values):
    """pytensor/tensor/subtensor.py ``unflatten_index_variables``: ints in
    ``idx_list`` are positions into the runtime index inputs."""

    def conv(e):
        if isinstance(e, slice):
            return slice(conv(e.start), conv(e.stop), conv(e.step))
        if e is None:
            return None
        v = index_values[e]
        if isinstance(v, np.ndarray) and v.ndim == 0:
            return v[()]
        return v

    return tuple(conv(e) for e in idx_list)


def _as_index(v):
    if isinstance(v, np.ndarray) and v.ndim == 0:
        return int(v)
    return v


@op("Subtensor")
def _subtensor(p, inputs, node, graph):
    # pytensor/tensor/subtensor.py:912-917 (Subtensor.perform)
    x, *idx = inputs
    cdata = unflatten_index(p["idx_list"], [_as_index(i) for i in idx])
    return [np.asarray(x[cdata])]


@op("IncSubtensor")
def _inc_subtensor(p, inputs, node, graph):
    # pytensor/tensor/subtensor.py:1441+ (IncSubtensor.perform)
    x, y, *idx = inputs
    cdata = unflatten_index(p["idx_list"], [_as_index(i) for i in idx])
    out = x.copy()
    if p["set_instead_of_inc"]:
        out[cdata] = y
    else:
        out[cdata] += y
    return [out]


@op("AdvancedSubtensor")
def _adv_subtensor(p, inputs, node, graph):
    # pytensor/tensor/subtensor.py:1932+ (AdvancedSubtensor.perform): x[indices]
    x, *idx = inputs
    cdata = unflatten_index(p["idx_list"], idx)
    return [np.asarray(x[cdata])]


@op("AdvancedIncSubtensor")
def _adv_inc_subtensor(p, inputs, node, graph):
    # pytensor/tensor/subtensor.py:2275+ (AdvancedIncSubtensor.perform):
    # set → out[idx] = y ; inc → np.add.at(out, idx, y) (duplicates accumulate)
    x, y, *idx = inputs
    cdata = unflatten_index(p["idx_list"], idx)
    out = x.copy()
    if p["set_instead_of_inc"]:
        out[cdata] = y
    elif p.get("ignore_duplicates"):
        out[cdata] += y
    else:
        np.add.at(out, cdata, y)
    return [out]


@op("Alloc")
def _alloc(p, inputs, node, graph):
    # pytensor/tensor/basic.py:1545+ (Alloc.perform): broadcast value to shape
    v, *shape = inputs
    shape = tuple(int(s) for s in shape)
    out = np.empty(shape, dtype=v.dtype)
    out[...] = v
    return [out]


@op("AllocEmpty")
def _alloc_empty(p, inputs, node, graph):
    # pytensor/tensor/basic.py:4197+ (AllocEmpty.perform); contents unspecified,
    # the oracle zero-fills so comparisons of *defined* regions stay meaningful.
    shape = tuple(int(s) for s in inputs)
    return [np.zeros(shape, dtype=p["dtype"])]


@op("MakeVector")
def _make_vector(p, inputs, node, graph):
    # pytensor/tensor/basic.py:1900+ (MakeVector.perform)
    return [np.asarray(inputs, dtype=p["dtype"]).reshape(len(inputs))]


@op("Shape_i")
def _shape_i(p, inputs, node, graph):
    # pytensor/tensor/shape.py:201+ (Shape_i.perform)
    return [np.asarray(np.shape(inputs[0])[p["i"]], dtype="int64")]


@op("Shape")
def _shape(p, inputs, node, graph):
    return [np.asarray(np.shape(inputs[0]), dtype="int64")]


@op("Reshape")
def _reshape(p, inputs, node, graph):
    # pytensor/tensor/shape.py:613+ (Reshape.perform)
    x, shp = inputs
    return [np.reshape(x, tuple(int(s) for s in np.asarray(shp).ravel()))]


@op("SpecifyShape")
def _specify_shape(p, inputs, node, graph):
    x, *shape = inputs
    for d, s in enumerate(shape):
        if s is not None and np.shape(x)[d] != int(s):
            raise AssertionError(f"SpecifyShape: dim {d} of input has shape {np.shape(x)[d]}, expected {int(s)}.")
    return [x]


@op("ExtractDiag")
def _extract_diag(p, inputs, node, graph):
    # pytensor/tensor/basic.py:3636+ (ExtractDiag.perform): np.diagonal
    (x,) = inputs
    return [np.asarray(x.diagonal(p["offset"], p["axis1"], p["axis2"]))]


@op("ScalarFromTensor")
def _scalar_from_tensor(p, inputs, node, graph):
    # pytensor/tensor/basic.py:684+
    return [np.asarray(inputs[0])[()]]


@op("TensorFromScalar")
def _tensor_from_scalar(p, inputs, node, graph):
    # pytensor/tensor/basic.py:627+
    return [np.asarray(inputs[0])]


@op("Join")
def _join(p, inputs, node, graph):
    # pytensor/tensor/basic.py:2405+ (Join.perform): np.concatenate
    return [np.concatenate(inputs, axis=p["axis"])]


@op("CheckAndRaise")
def _check_and_raise(p, inputs, node, graph):
    # pytensor/raise_op.py:26+ (CheckAndRaise.perform)
    x, *conds = inputs
    if not all(np.all(c) for c in conds):
        import builtins

        exc = np.linalg.LinAlgError if p["exc_type"] == "LinAlgError" else getattr(builtins, p["exc_type"], RuntimeError)
        raise exc(p["msg"])
    return [x]


@op("DeepCopyOp")
def _deepcopy(p, inputs, node, graph):
    # pytensor/compile/ops.py:121+
    return [np.array(inputs[0], copy=True)]


@op("ViewOp")
def _view(p, inputs, node, graph):
    # pytensor/compile/ops.py:87+
    return [inputs[0]]


@op("Scan")
def _scan(p, inputs, node, graph):
    """pytensor/scan/op.py:1827+ (``Scan.perform``), restricted to the tap kinds
    the hot path uses: seqs, mit-sot / sit-sot recurrences, nit-sot outputs,
    untraced sit-sot, non-seqs, optional while-condition.  (mit-mot — only
    produced by ``Scan.pullback`` — is not restated.)

    Outer input order (op.py:322-635): n_steps, seqs, mit_mot, mit_sot, sit_sot,
    untraced_sit_sot, nit_sot lengths, non_seqs.  Recurrent buffers hold the
    initial taps first; step t reads buffer[t + tap - min_tap] and writes
    buffer[t - min_tap] (circularly when the buffer is shorter than needed).
    """
    info = p["info"]
    inner: "Graph" = p["inner"]
    n_steps = int(inputs[0])
    k = 1
    seqs = inputs[k : k + info["n_seqs"]]
    k += info["n_seqs"]
    # mit-mot (op.py:2091-2140): inputs read at the `in` taps, the first inner outputs written
    # back at the `out` taps of the same buffer: buf[out_tap + pos], pos = step - min(in taps)
    mm_in = [list(t) for t in info["mit_mot_in_slices"]]
    mm_out = [list(t) for t in info["mit_mot_out_slices"]]
    n_mm = len(mm_in)
    mit_sot_taps = mm_in + [list(t) for t in info["mit_sot_in_slices"]]
    sit_sot_taps = [list(t) for t in info["sit_sot_in_slices"]]
    n_ms, n_ss = len(mit_sot_taps), len(sit_sot_taps)
    rec_bufs = [np.array(b, copy=True) for b in inputs[k : k + n_ms + n_ss]]
    k += n_ms + n_ss
    untraced = list(inputs[k : k + info["n_untraced_sit_sot"]])
    k += info["n_untraced_sit_sot"]
    nit_lens = [int(x) for x in inputs[k : k + info["n_nit_sot"]]]
    k += info["n_nit_sot"]
    non_seqs = list(inputs[k:])
    taps = mit_sot_taps + sit_sot_taps
    mintaps = [-min(t) for t in taps]
    nit_bufs = [None] * info["n_nit_sot"]
    steps_done = 0
    for t in range(n_steps):
        inner_in = [s[t] for s in seqs]
        for buf, tp, mt in zip(rec_bufs, taps, mintaps):
            L = buf.shape[0]
            for tap in tp:
                inner_in.append(buf[(t + mt + tap) % L])
        inner_in += untraced
        inner_in += non_seqs
        outs = [np.array(v, copy=True) for v in run_graph(inner, inner_in)]
        o = 0
        for j, (buf, mt) in enumerate(zip(rec_bufs, mintaps)):
            if j < n_mm:
                for tap in mm_out[j]:
                    buf[(t + mt + tap) % buf.shape[0]] = outs[o]
                    o += 1
                continue
            buf[(t + mt) % buf.shape[0]] = outs[o]
            o += 1
        for j in range(info["n_nit_sot"]):
            if nit_bufs[j] is None:
                nit_bufs[j] = np.zeros((nit_lens[j], *np.shape(outs[o])), dtype=np.asarray(outs[o]).dtype)
            nit_bufs[j][t % nit_lens[j]] = outs[o]
            o += 1
        for j in range(info["n_untraced_sit_sot"]):
            untraced[j] = outs[o]
            o += 1
        steps_done = t + 1
        if info["as_while"] and bool(outs[o]):
            break
    res = []
    for j, (buf, mt) in enumerate(zip(rec_bufs, mintaps)):
        # rotate circular buffers so that the oldest entry comes first (op.py:2087-2130)
        L = buf.shape[0]
        end = (steps_done + mt) % L
        if steps_done + mt > L and end != 0:
            buf = np.concatenate([buf[end:], buf[:end]])
        elif j >= n_mm and L > steps_done + mt and n_steps > 0 and not info["as_while"]:
            # op.py:2280-2286: a buffer longer than the steps taken (truncated back-propagation
            # through time): "Scan is expected to return 0 for all entries for which the gradient
            # is not actually computed"
            buf[steps_done + mt :] = 0
        if info["as_while"]:
            buf = buf[: steps_done + mt]
        res.append(buf)
    for j, buf in enumerate(nit_bufs):
        if buf is None:
            ov = inner.vars[inner.outputs[sum(len(t) for t in mm_out) + (n_ms - n_mm) + n_ss + j]]
            buf = np.zeros((0,) * (ov.ndim + 1), dtype=ov.dtype)
        elif steps_done > nit_lens[j] and steps_done % nit_lens[j]:
            e = steps_done % nit_lens[j]
            buf = np.concatenate([buf[e:], buf[:e]])
        if info["as_while"]:
            buf = buf[:steps_done]
        res.append(buf)
    res += untraced
    return res


# ---------------------------------------------------------------------------
# graph interpreter
# ---------------------------------------------------------------------------


def run_graph(graph, inputs):
    """Evaluate ``graph`` (pytensor_amd.ir.Graph) on host arrays; returns a list."""
    env = {}
    for vid, v in graph.vars.items():
        if v.const is not None:
            env[vid] = v.const
        elif v.kind == "none":
            env[vid] = None  # NoneConst (an unspecified SpecifyShape entry)
    if len(inputs) != len(graph.inputs):
        raise TypeError(f"expected {len(graph.inputs)} inputs, got {len(inputs)}")
    for vid, val in zip(graph.inputs, inputs):
        env[vid] = val if graph.vars[vid].kind != "tensor" else np.asarray(val)
    guard, members = _branch_guards(graph)
    open_branches, taken = set(), {}
    todo = list(range(len(graph.nodes)))[::-1]
    while todo:
        k = todo.pop()
        node = graph.nodes[k]
        if guard[k] is not None and guard[k] not in open_branches:
            continue
        if node.op == "IfElse":
            # pytensor/ifelse.py:300-345 (the lazy thunk): the condition, then only the branch taken
            n_out = len(node.outputs)
            if k not in taken:
                b = 0 if np.asarray(env[node.inputs[0]]).item() != 0 else 1
                taken[k] = b
                open_branches.add((k, b))
                todo.append(k)
                todo.extend(reversed(members.get((k, b), ())))
                continue
            b = taken[k]
            for vid, src in zip(node.outputs, node.inputs[1 + b * n_out : 1 + (b + 1) * n_out]):
                env[vid] = np.array(env[src], copy=True)
            continue
        f = OPS.get(node.op)
        if f is None:
            raise NotImplementedError(f"oracle has no handler for {node.op}")
        outs = f(node.params, [env[i] for i in node.inputs], node, graph)
        for vid, val in zip(node.outputs, outs):
            env[vid] = val
    return [env[o] for o in graph.outputs]


def _branch_guards(graph):
    """nodes whose values reach the outputs only through ONE branch of an IfElse run only when that
    branch is taken (the reference's VM is lazy): guard[k] = (IfElse node, branch), innermost first"""
    nodes = graph.nodes
    guard, members = [None] * len(nodes), {}
    if not any(n.op == "IfElse" for n in nodes):
        return guard, members
    uses = {}
    for k, n in enumerate(nodes):
        for pos, i in enumerate(n.inputs):
            uses.setdefault(i, []).append((k, pos))
    outs = set(graph.outputs)
    for k, n in enumerate(nodes):
        if n.op != "IfElse":
            continue
        n_out = len(n.outputs)
        for b in (0, 1):
            lo, hi = 1 + b * n_out, 1 + (b + 1) * n_out
            inside = set()
            for j in range(k - 1, -1, -1):
                m = nodes[j]
                if any(o in outs for o in m.outputs):
                    continue
                us = [u for o in m.outputs for u in uses.get(o, ())]
                if us and all((uk == k and lo <= up < hi) or uk in inside for uk, up in us):
                    inside.add(j)
            mine = [j for j in sorted(inside) if guard[j] is None]
            for j in mine:
                guard[j] = (k, b)
            members[(k, b)] = mine
    return guard, members


# ---------------------------------------------------------------------------
# fused IR nodes (pytensor_amd/fusion.py) restated from their unfused parts, so that the
# IR passes themselves can be checked on the CPU: oracle(pass(graph)) == oracle(graph)
# ---------------------------------------------------------------------------


def _sum_partial_inputs(p, inputs):
    """Inputs listed in ``partial_inputs`` are split-K slabs (S, *shape) of a ``GemmPartials``
    node (fusion.defer_gemm_finish): their value is the sum over the slab axis."""
    gspec = p.get("gather")
    if gspec:
        # gatherfuse.absorb_gathers: body input `pos` is a table read as table[idx], the index
        # vectors follow the body inputs (AdvancedSubtensor on axis 0: subtensor.py:1932)
        nbody = len(p["scalar"]["in_dtypes"])
        body_in = list(inputs[:nbody])
        for pos, extra in gspec:
            body_in[pos] = np.asarray(body_in[pos])[np.asarray(inputs[nbody + extra])]
        inputs = body_in
    pi = p.get("partial_inputs")
    if not pi:
        return inputs
    return [np.sum(x, axis=0) if k in pi else x for k, x in enumerate(inputs)]


@op("ElemwiseReduce")
def _elemwise_reduce(p, inputs, node, graph):
    inputs = _sum_partial_inputs(p, inputs)
    outs = eval_scalar_body(p["scalar"], inputs)
    shape = np.broadcast(*inputs).shape if inputs else ()
    res = []
    for o, spec, dt in zip(outs, p["reduce"], p["scalar"]["out_dtypes"]):
        full = np.array(np.broadcast_to(o, shape), dtype=dt, order="C")
        if spec is None:
            res.append(full)
        else:
            uf = _REDUCE[spec["op"]]
            res.append(np.asarray(uf.reduce(full.ravel(), dtype=np.dtype(spec["acc_dtype"]))).astype(spec["dtype"]))
    return res


@op("ElemwiseAxisReduce")
def _elemwise_axis_reduce(p, inputs, node, graph):
    # axisfuse.fuse_elemwise_axis_reduce restated from its parts: Elemwise.perform (elemwise.py:755-823) then
    # CAReduce.perform per output (elemwise.py:1493-1511)
    outs = eval_scalar_body(p["scalar"], inputs)
    shape = np.broadcast(*inputs).shape if inputs else ()
    res = []
    for o, spec, dt in zip(outs, p["reduce"], p["scalar"]["out_dtypes"]):
        full = np.array(np.broadcast_to(o, shape), dtype=dt, order="C")
        if spec["op"] == "LogSumExp":
            # axisfuse.fuse_logsumexp: log(sum(exp(.))) over the axes, restated with scipy's stable form
            # (the reference's own LogSumExp perform: pytensor/tensor/special.py:102-120)
            import scipy.special

            res.append(np.asarray(scipy.special.logsumexp(full.astype(spec["acc_dtype"]), axis=tuple(p["axis"]))).astype(spec["dtype"]))
            continue
        sub = {"axis": p["axis"], "scalar_op": spec["op"], "acc_dtype": spec["acc_dtype"], "dtype": spec["dtype"]}
        res.append(_careduce(sub, [full], node, graph)[0])
    return res


@op("GemvChain")
def _gemv_chain(p, inputs, node, graph):
    y1, a1, A, x1, b1, *rest = inputs
    r = _gemv({}, [y1, a1, A, x1, b1], node, graph)[0]
    gather = set(p.get("gather") or [])
    it = iter(rest)
    e_in = []
    for pos in range(len(p["scalar"]["in_dtypes"])):
        if pos == p["r_pos"]:
            e_in.append(r)
        elif pos in gather:
            table, idx = next(it), next(it)
            e_in.append(table[idx])  # AdvancedSubtensor on axis 0
        else:
            e_in.append(next(it))
    outs = _elemwise_reduce({"scalar": p["scalar"], "reduce": p["reduce"]}, e_in, node, graph)
    w = outs[p["w_out"]]
    part = np.dot(A.T, w)[None, :]
    res = ([r] if p["store_r"] else []) + outs + [part]
    if p.get("scatter_out") is not None:
        sidx, base = next(it), next(it)
        nbins = int(np.asarray(base).reshape(-1)[0]) if p.get("scatter_len_input") else base.shape[0]
        acc = np.zeros(nbins, dtype="float64")
        np.add.at(acc, sidx, outs[p["scatter_out"]])  # AdvancedIncSubtensor(inc) on zeros
        res.append(acc[None, :])
    return res


@op("GemvFinish")
def _gemv_finish(p, inputs, node, graph):
    part, y2, a2, b2 = inputs
    s = part.sum(axis=0)
    # (the slabs of a float32 chain are float64: the node's value has the dtype the graph gives its output)
    return [np.asarray((a2 * s if b2 == 0.0 else b2 * y2 + a2 * s), dtype=graph.vars[node.outputs[0]].dtype)]


@op("SeqDot22")
def _seq_dot22(p, inputs, node, graph):
    seq, W = inputs
    return [np.matmul(seq, W)]


@op("CholeskyTrsv")
def _cholesky_trsv(p, inputs, node, graph):
    S, b = inputs
    L = _cholesky({"lower": True}, [S], node, graph)[0]
    x = _solve_tri({"lower": True, "unit_diagonal": False, "b_ndim": 1}, [L, b], node, graph)[0]
    return [L, x]


@op("ARange")
def _arange(p, inputs, node, graph):
    # pytensor/tensor/basic.py:3139 ARange.perform
    start, stop, step = (np.asarray(i).reshape(()) for i in inputs)
    return [np.arange(start, stop, step, dtype=p["dtype"])]


@op("Eye")
def _eye(p, inputs, node, graph):
    # pytensor/tensor/basic.py:1351 Eye.perform
    n, m, k = (int(np.asarray(i).reshape(())) for i in inputs)
    return [np.eye(n, m, k, dtype=p["dtype"])]


@op("CumOp")
def _cumop(p, inputs, node, graph):
    # pytensor/tensor/extra_ops.py:281 CumOp.perform
    (x,) = inputs
    f = np.cumsum if p["mode"] == "add" else np.cumprod
    # accumulate in the declared output dtype (= the input dtype, extra_ops.py:302 make_node):
    # NumPy alone would widen small integers to the platform integer; the reference's C code
    # (extra_ops.py c_code) accumulates in the output type and wraps, and the C linker wins
    return [f(x, axis=p["axis"], dtype=x.dtype)]


@op("Argmax")
def _argmax(p, inputs, node, graph):
    # pytensor/tensor/math.py:188-206 Argmax.perform: kept axes first, reduced axes flattened last
    (x,) = inputs
    axes = list(p["axis"])
    keep = [d for d in range(x.ndim) if d not in axes]
    xt = np.transpose(x, keep + axes)
    kept_shape = xt.shape[: len(keep)]
    r = xt.reshape((*kept_shape, int(np.prod(xt.shape[len(keep) :], dtype="int64"))))
    return [np.asarray(np.argmax(r, axis=-1), dtype="int64")]


@op("Solve")
def _solve_general(p, inputs, node, graph):
    # pytensor/tensor/linalg/solvers/general.py:62-75 (Solve.perform): scipy.linalg.solve; a
    # singular system is NaN-filled (the reference fills a.shape — we fill the solution's shape)
    a, b = inputs
    if a.ndim > 2 or b.ndim > p["b_ndim"]:  # Blockwise batching
        bshape = np.broadcast_shapes(a.shape[:-2], b.shape[: b.ndim - p["b_ndim"]])
        ab = np.broadcast_to(a, (*bshape, *a.shape[-2:]))
        bb = np.broadcast_to(b, (*bshape, *b.shape[b.ndim - p["b_ndim"] :]))
        out = np.empty(bb.shape, dtype=b.dtype)
        for idx in np.ndindex(*bshape):
            out[idx] = _solve_general(p, [ab[idx], bb[idx]], node, graph)[0]
        return [out]
    try:
        return [scipy.linalg.solve(a, b, lower=p["lower"], check_finite=False, assume_a=p["assume_a"])]
    except np.linalg.LinAlgError:
        return [np.full(b.shape, np.nan, dtype=b.dtype)]


@op("Det")
def _det(p, inputs, node, graph):
    # pytensor/tensor/linalg/summary.py:59-65 (Det.perform): np.linalg.det
    return [np.asarray(np.linalg.det(inputs[0]))]


@op("SLogDet")
def _slogdet(p, inputs, node, graph):
    # pytensor/tensor/linalg/summary.py:101-107 (SLogDet.perform): np.linalg.slogdet
    s, l = np.linalg.slogdet(inputs[0])
    return [np.asarray(s), np.asarray(l)]


@op("SortOp")
def _sort_op(p, inputs, node, graph):
    # pytensor/tensor/sort.py:52-55 (SortOp.perform)
    return [np.sort(inputs[0], int(inputs[1]), p["kind"])]


@op("ArgSortOp")
def _argsort_op(p, inputs, node, graph):
    # pytensor/tensor/sort.py:180-186 (ArgSortOp.perform); a stable kind is used whatever the op
    # says so that ties have one defined answer (the one the device kernel gives)
    return [np.asarray(np.argsort(inputs[0], int(inputs[1]), "stable"), dtype=p.get("dtype", "int64"))]


@op("Nonzero")
def _nonzero(p, inputs, node, graph):
    # pytensor/tensor/basic.py Nonzero.perform: np.nonzero, int64 vectors
    return [np.asarray(r, dtype=np.int64) for r in np.nonzero(inputs[0])]


@op("Split")
def _split(p, inputs, node, graph):
    # pytensor/tensor/basic.py:2268-2283 (Split.perform)
    x, splits = inputs
    splits = np.asarray(splits)
    if len(splits) != p["len_splits"]:
        raise ValueError("Length of splits is not equal to n_splits")
    if splits.sum() != x.shape[p["axis"]]:
        raise ValueError(f"Split sizes sum to {splits.sum()}; expected {x.shape[p['axis']]}")
    if (splits < 0).any():
        raise ValueError("Split sizes cannot be negative")
    return list(np.split(x, np.cumsum(splits[:-1]), axis=p["axis"]))


@op("RandomVariable")
def _random_variable(p, inputs, node, graph):
    # RandomVariable.perform (pytensor/tensor/random/op.py) with the hip linker's own stream: the
    # restatement of csrc/random.hip in philox_ref.py (the reference's draws cannot be matched)
    import philox_ref

    gen, size, *params = inputs
    size = None if p["size_is_none"] else [int(v) for v in np.asarray(size).ravel()]
    return list(philox_ref.draw(p["name"], gen, size, params, p["dtype"], p.get("ndims_params"), p.get("method", "cholesky")))


@op("Eigh")
def _eigh(p, inputs, node, graph):
    # pytensor/tensor/linalg/decomposition/eigen.py:177-195 (Eigh.perform, standard problem)
    if len(inputs) == 2:  # generalised problem A v = w B v (perform 179-186)
        w, v = scipy.linalg.eigh(inputs[0], b=inputs[1], lower=p["lower"])
    else:
        w, v = scipy.linalg.eigh(inputs[0], lower=p["lower"])
    return [w, v]


@op("Eigvalsh")
def _eigvalsh(p, inputs, node, graph):
    # pytensor/tensor/linalg/decomposition/eigen.py:363 (Eigvalsh.perform): scipy.linalg.eigvalsh
    ins = [i for i in inputs if i is not None]
    return [scipy.linalg.eigvalsh(ins[0], b=ins[1] if len(ins) == 2 else None, lower=p["lower"])]


@op("QR")
def _qr(p, inputs, node, graph):
    # pytensor/tensor/linalg/decomposition/qr.py:153-221 (QR.perform): LAPACK geqrf, R = its upper
    # triangle (the leading n rows for economic / raw when m >= n), Q from orgqr
    (x,) = inputs
    M, N = x.shape
    geqrf, orgqr = scipy.linalg.get_lapack_funcs(("geqrf", "orgqr"), (x,))
    qr, tau, _w, _info = geqrf(x)
    mode = p["mode"]
    R = np.triu(qr) if (mode not in ("economic", "raw") or M < N) else np.triu(qr[:N, :])
    if mode == "r":
        return [R]
    if mode == "raw":
        return [qr, tau, R]
    if M < N:
        Q = orgqr(qr[:, :M], tau)[0]
    elif mode == "economic":
        Q = orgqr(qr, tau)[0]
    else:
        qqr = np.empty((M, M), dtype=qr.dtype)
        qqr[:, :N] = qr
        Q = orgqr(qqr, tau)[0]
    return [Q, R]


@op("SVD")
def _svd(p, inputs, node, graph):
    # pytensor/tensor/linalg/decomposition/svd.py:85-93 (SVD.perform): np.linalg.svd
    (x,) = inputs
    if p["compute_uv"]:
        return list(np.linalg.svd(x, p["full_matrices"], True))
    return [np.linalg.svd(x, p["full_matrices"], False)]


@op("MatrixPinv")
def _matrix_pinv(p, inputs, node, graph):
    # pytensor/tensor/linalg/inverse.py:32-35 (MatrixPinv.perform): np.linalg.pinv
    return [np.linalg.pinv(inputs[0], hermitian=p["hermitian"])]


@op("Lstsq")
def _lstsq(p, inputs, node, graph):
    # pytensor/tensor/linalg/solvers/lstsq.py:29-34 (Lstsq.perform): np.linalg.lstsq; the outputs are
    # typed dmatrix / dvector / iscalar / dvector by make_node
    x, res, rank, s = np.linalg.lstsq(inputs[0], inputs[1], inputs[2])
    return [np.asarray(x, dtype="float64"), np.asarray(res, dtype="float64"), np.asarray(rank, dtype="int32"), np.asarray(s, dtype="float64")]


@op("TensorInv")
def _tensor_inv(p, inputs, node, graph):
    # pytensor/tensor/linalg/inverse.py:187-190 (TensorInv.perform): np.linalg.tensorinv
    return [np.linalg.tensorinv(inputs[0], p["ind"])]


@op("TensorSolve")
def _tensor_solve(p, inputs, node, graph):
    # pytensor/tensor/linalg/solvers/lstsq.py:58-65 (TensorSolve.perform): np.linalg.tensorsolve
    return [np.linalg.tensorsolve(inputs[0], inputs[1], None if p.get("axes") is None else tuple(p["axes"]))]


@op("LUFactorTridiagonal")
def _lu_factor_tridiagonal(p, inputs, node, graph):
    # pytensor/tensor/linalg/solvers/tridiagonal.py:70-90: LAPACK gttrf (ipiv 1-based, as returned)
    dt = np.result_type(*[np.asarray(i).dtype for i in inputs])
    gttrf = scipy.linalg.get_lapack_funcs("gttrf", dtype=dt)
    dl, d, du, du2, ipiv, _info = gttrf(*[np.asarray(i, dtype=dt) for i in inputs])
    return [dl, d, du, du2, ipiv]


@op("SolveLUFactorTridiagonal")
def _solve_lu_factor_tridiagonal(p, inputs, node, graph):
    # pytensor/tensor/linalg/solvers/tridiagonal.py:170-180: LAPACK gttrs
    dl, d, du, du2, ipiv, b = inputs
    dt = np.result_type(dl.dtype, d.dtype, du.dtype, du2.dtype, b.dtype)
    gttrs = scipy.linalg.get_lapack_funcs("gttrs", dtype=dt)
    x, _info = gttrs(*[np.asarray(v, dtype=dt) for v in (dl, d, du, du2)], ipiv, np.asarray(b, dtype=dt), trans="T" if p["transposed"] else "N")
    return [x]


@op("BlockDiagonal")
def _block_diagonal(p, inputs, node, graph):
    # pytensor/tensor/linalg/constructors.py:73-75: scipy.linalg.block_diag cast to the output dtype
    return [scipy.linalg.block_diag(*inputs).astype(p["dtype"])]


@op("RFFTOp")
def _rfft(p, inputs, node, graph):
    # pytensor/tensor/fft.py:39-48 (RFFTOp.perform)
    a, s = inputs
    s = tuple(int(v) for v in s)
    A = np.fft.rfftn(a, s=s, axes=tuple(range(a.ndim - len(s), a.ndim)))
    out = np.zeros((*A.shape, 2), dtype=a.dtype)
    out[..., 0], out[..., 1] = np.real(A), np.imag(A)
    return [out]


@op("IRFFTOp")
def _irfft(p, inputs, node, graph):
    # pytensor/tensor/fft.py:109-117 (IRFFTOp.perform): numpy's 1/n normalisation removed
    a, s = inputs
    s = np.asarray(s)
    inp = a[..., 0] + 1j * a[..., 1]
    out = np.fft.irfftn(inp, s=tuple(int(v) for v in s), axes=tuple(range(inp.ndim - len(s), inp.ndim)))
    return [(out * s.prod()).astype(a.dtype)]


@op("Convolve2d")
def _convolve2d(p, inputs, node, graph):
    # pytensor/tensor/signal/conv.py:260-263: scipy.signal.convolve (direct sums: the restated kernel)
    import scipy.signal

    return [scipy.signal.convolve(inputs[0], inputs[1], mode="full" if bool(inputs[2]) else "valid", method="direct")]


@op("Choose")
def _choose(p, inputs, node, graph):
    # pytensor/tensor/basic.py (Choose.perform): np.choose
    return [np.choose(inputs[0], inputs[1], mode=p["mode"])]


@op("PermuteRowElements")
def _permute_row_elements(p, inputs, node, graph):
    # pytensor/tensor/basic.py:3500-3560 (_rec_perform): a permutation per row, leading dims broadcast
    x, y = inputs
    nd = max(x.ndim, y.ndim)
    x = x.reshape((1,) * (nd - x.ndim) + x.shape)
    y = y.reshape((1,) * (nd - y.ndim) + y.shape)
    shape = np.broadcast_shapes(x.shape, y.shape)
    xb, yb = np.broadcast_to(x, shape), np.broadcast_to(y, shape)
    out = np.empty(shape, dtype=x.dtype)
    for idx in np.ndindex(*shape[:-1]):
        if p["inverse"]:
            out[idx][yb[idx]] = xb[idx]
        else:
            out[idx] = xb[idx][yb[idx]]
    return [out]


@op("SolveSylvester")
def _solve_sylvester(p, inputs, node, graph):
    # pytensor/tensor/linalg/solvers/linear_control.py:117-165: A X + X B = C through real Schur forms and
    # trsyl — what scipy.linalg.solve_sylvester does
    return [scipy.linalg.solve_sylvester(*inputs)]


@op("Expm")
def _expm(p, inputs, node, graph):
    # pytensor/tensor/linalg/products.py:35-38 (Expm.perform): scipy.linalg.expm
    return [scipy.linalg.expm(inputs[0])]


@op("CpuContiguous")
def _cpu_contiguous(p, inputs, node, graph):
    # pytensor/tensor/extra_ops.py:67-75
    return [np.ascontiguousarray(inputs[0])]


@op("JoinDims")
def _join_dims(p, inputs, node, graph):
    # pytensor/tensor/reshape.py:72-82
    (x,) = inputs
    a, n = p["start_axis"], p["n_axes"]
    return [x.reshape((*x.shape[:a], -1, *x.shape[a + n :]))]


@op("SplitDims")
def _split_dims(p, inputs, node, graph):
    # pytensor/tensor/reshape.py:197-203
    x, shape = inputs
    return [x.reshape((*x.shape[: p["axis"]], *[int(v) for v in np.asarray(shape).ravel()], *x.shape[p["axis"] + 1 :]))]


@op("FillDiagonal")
def _fill_diagonal(p, inputs, node, graph):
    # pytensor/tensor/extra_ops.py:871-886: rectangular matrices accepted, no wrap
    a, val = inputs[0].copy(), inputs[1]
    if a.ndim == 2:
        a.flat[: a.shape[1] * a.shape[1] : a.shape[1] + 1] = val
    else:
        np.fill_diagonal(a, val)
    return [a]


@op("FillDiagonalOffset")
def _fill_diagonal_offset(p, inputs, node, graph):
    # pytensor/tensor/extra_ops.py:971-1004
    a, val, offset = inputs[0].copy(), inputs[1], int(inputs[2])
    height, width = a.shape
    if offset >= 0:
        start, steps = offset, min(min(width, height), width - offset)
    else:
        start, steps = -offset * a.shape[1], min(min(width, height), height + offset)
    step = a.shape[1] + 1
    a.flat[start : start + step * steps : step] = val
    return [a]


@op("Bartlett")
def _bartlett(p, inputs, node, graph):
    # pytensor/tensor/extra_ops.py (Bartlett.perform): np.bartlett
    return [np.bartlett(int(inputs[0]))]


@op("SearchsortedOp")
def _searchsorted(p, inputs, node, graph):
    # pytensor/tensor/extra_ops.py:155-166
    x, v = inputs[:2]
    return [np.searchsorted(x, v, side=p["side"], sorter=inputs[2] if len(inputs) == 3 else None).astype("int64")]


@op("Repeat")
def _repeat(p, inputs, node, graph):
    # pytensor/tensor/extra_ops.py:706-708
    return [np.repeat(inputs[0], repeats=inputs[1], axis=p["axis"])]


@op("UnravelIndex")
def _unravel_index(p, inputs, node, graph):
    # pytensor/tensor/extra_ops.py (UnravelIndex.perform): np.unravel_index, int64 copies
    return [np.array(r, dtype="int64") for r in np.unravel_index(inputs[0], tuple(int(d) for d in inputs[1]), order=p["order"])]


@op("RavelMultiIndex")
def _ravel_multi_index(p, inputs, node, graph):
    # pytensor/tensor/extra_ops.py (RavelMultiIndex.perform): np.ravel_multi_index
    *mi, dims = inputs
    return [np.asarray(np.ravel_multi_index(mi, tuple(int(d) for d in dims), mode=p["mode"], order=p["order"]), "int64")]


@op("Unique")
def _unique(p, inputs, node, graph):
    # pytensor/tensor/extra_ops.py:1227-1240 (old_np_unique: the inverse of the flattened input is 1-d)
    (x,) = inputs
    outs = np.unique(x, return_index=p["return_index"], return_inverse=p["return_inverse"], return_counts=p["return_counts"], axis=p["axis"])
    if not isinstance(outs, tuple):
        return [outs]
    outs = list(outs)
    if p["return_inverse"]:  # npy_2_compat.old_np_unique: 1-d inverse (flat, or along the axis)
        k = 1 + int(p["return_index"])
        outs[k] = outs[k].reshape(-1) if p["axis"] is None else outs[k].reshape((x.shape[p["axis"]],))
    return outs


@op("LU")
def _lu(p, inputs, node, graph):
    # pytensor/tensor/linalg/decomposition/lu.py:77-89: scipy.linalg.lu
    return list(scipy.linalg.lu(inputs[0], permute_l=p["permute_l"], p_indices=p["p_indices"]))


@op("Convolve1d")
def _convolve1d(p, inputs, node, graph):
    # pytensor/tensor/signal/conv.py:124-128: np.convolve, "full" if the third input is true
    return [np.convolve(inputs[0], inputs[1], mode="full" if bool(inputs[2]) else "valid")]


@op("MatrixInverse")
def _matrix_inverse(p, inputs, node, graph):
    # pytensor/tensor/linalg/inverse.py:114-117 (MatrixInverse.perform): np.linalg.inv
    return [np.linalg.inv(inputs[0])]


@op("Softmax")
def _softmax(p, inputs, node, graph):
    # pytensor/tensor/special.py:44-47 (Softmax.build_inner_graph) / 85-87 (LogSoftmax): the
    # inner graph the reference inlines, evaluated as written (float32 sums accumulate in
    # float64 like Sum's acc_dtype, elemwise.py:1383-1417)
    (x,) = inputs
    axis = tuple(p["axis"])
    with np.errstate(all="ignore"):
        xs = x - x.max(axis=axis, keepdims=True)
        e = np.exp(xs)
        acc = np.float64 if x.dtype == np.float32 else x.dtype
        s = e.sum(axis=axis, keepdims=True, dtype=acc).astype(x.dtype)
        return [(xs - np.log(s)) if p["log"] else (e / s)]


@op("GemmPartials")
def _gemm_partials(p, inputs, node, graph):
    # one slab = the whole product; the consumer's "partial_inputs" sum over the slab axis
    A, B = inputs
    return [np.dot(A, B)[None]]


@op("PackB16")
def _pack_b16(p, inputs, node, graph):
    # gemmfuse.fuse_dot_epilogue: the right operand in the device kernel's operand order,
    # Bp[ct][k4][j][q] = B[4*k4+q][16*ct+j] zero padded (include/pthip.h pthip_pack_b16); only
    # the device reads it — DotEpilogue below takes the plain matrix
    (B,) = inputs
    K, N = B.shape
    Kp, Np = (K + 15) // 16 * 16, (N + 15) // 16 * 16
    pad = np.zeros((Kp, Np), dtype=B.dtype)
    pad[:K, :N] = B
    return [np.ascontiguousarray(pad.reshape(Kp // 4, 4, Np // 16, 16).transpose(2, 0, 3, 1)).reshape(-1)]


@op("DotEpilogue")
def _dot_epilogue(p, inputs, node, graph):
    # Dot22 (blas/gemm.py:248-275) per product, then Elemwise.perform (elemwise.py:755-823)
    nb = len(p["scalar"]["in_dtypes"])
    body_in = list(inputs[:nb])
    for j, q in enumerate(p["dot_inputs"]):
        body_in[q] = np.dot(body_in[q], inputs[nb + 2 * j])
    outs = eval_scalar_body(p["scalar"], body_in)
    shape = np.broadcast(*body_in).shape
    return [np.array(np.broadcast_to(o, shape), dtype=graph.vars[vid].dtype, order="C") for o, vid in zip(outs, node.outputs)]


@op("LUFactor")
def _lu_factor(p, inputs, node, graph):
    # pytensor/tensor/linalg/decomposition/lu.py:279-299 (LUFactor.perform): scipy getrf; a zero
    # pivot (info != 0) NaN-fills the factors.  Leading dims are a batch (Blockwise).
    import scipy.linalg

    (A,) = inputs
    n = A.shape[-1]
    LU = np.empty_like(A)
    piv = np.empty(A.shape[:-1], dtype=np.int32)
    if A.size:
        (getrf,) = scipy.linalg.get_lapack_funcs(("getrf",), (A,))
        for idx in np.ndindex(A.shape[:-2]):
            lu, pv, info = getrf(A[idx])
            if info != 0:
                lu[...] = np.nan
            LU[idx], piv[idx] = lu, pv
    return [LU, piv]


@op("PivotToPermutations")
def _pivot_to_permutations(p, inputs, node, graph):
    # lu.py:220-231 (PivotToPermutations.perform)
    (pivots,) = inputs
    out = np.empty(pivots.shape, dtype=np.int64)
    for idx in np.ndindex(pivots.shape[:-1]):
        pv = pivots[idx]
        p_inv = np.arange(len(pv), dtype="int64")
        for i in range(len(pv)):
            p_inv[i], p_inv[pv[i]] = p_inv[pv[i]], p_inv[i]
        out[idx] = p_inv if p["inverse"] else np.argsort(p_inv)
    return [out]


@op("ScatterScalars")
def _scatter_scalars(p, inputs, node, graph):
    # widefuse.collect_scalar_updates: a chain of IncSubtensor nodes (subtensor.py:1441 perform:
    # x[idx] = y / x[idx] += y on a copy), one element each at a constant index, in chain order
    if p.get("base_fill") is not None:  # base = alloc(fill, n), not materialised by the fused node
        out = np.full((int(inputs[0]),), p["base_fill"], dtype=graph.vars[node.outputs[0]].dtype)
    else:
        out = np.array(inputs[0], copy=True)
    for k, is_set, y in zip(p["indices"], p["set"], inputs[1:]):
        if is_set:
            out[k] = y
        else:
            out[k] += y
    return [out]


@op("MultiElemwise")
def _multi_elemwise(p, inputs, node, graph):
    # widefuse.fuse_independent_reductions: independent ElemwiseReduce nodes in one launch; the
    # values are the members' own
    res, pos = [], 0
    for t in p["terms"]:
        ins = inputs[pos : pos + t["n_inputs"]]
        pos += t["n_inputs"]
        res += _elemwise_reduce({"scalar": t["scalar"], "reduce": t["reduce"]}, list(ins), node, graph)
    return res


@op("AllReduce")
def _all_reduce(p, inputs, node, graph):
    # pytensor_amd/collective.py AllReduce.perform: element-wise reduction over the ranks of the
    # job (identity in a single process) — no reference Op exists; semantics = NumPy's
    # sum/prod/max/min over the stacked per-rank arrays
    from pytensor_amd import comm

    return [comm.all_reduce_host(inputs[0], p["op"])]


@op("Tail")
def _tail(p, inputs, node, graph):
    # tailfuse.fuse_tail: the member nodes in their original order (the fusion changes how many
    # launches they cost on the device, not what they compute)
    env = dict(zip(node.inputs, inputs))
    for sub in p["nodes"]:
        ins = [env[i] if i in env else graph.vars[i].const for i in sub.inputs]
        outs = OPS[sub.op](sub.params, ins, sub, graph)
        for vid, val in zip(sub.outputs, outs):
            env[vid] = val
    return [env[o] for o in node.outputs]
