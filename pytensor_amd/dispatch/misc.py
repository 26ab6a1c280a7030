"""Index / layout / signal ops of the widening tier: ``CpuContiguous``, ``JoinDims`` / ``SplitDims``,
``FillDiagonal`` / ``FillDiagonalOffset``, ``Bartlett``, ``SearchsortedOp``, ``Repeat``,
``UnravelIndex`` / ``RavelMultiIndex``, ``Unique``, ``LU``, ``Convolve1d`` / ``Convolve2d``, ``Choose``,
``PermuteRowElements``.

Reference: pytensor/tensor/extra_ops.py (CpuContiguous 47, SearchsortedOp 111, Repeat 639, Bartlett 776,
FillDiagonal 839, FillDiagonalOffset 943, Unique 1189, UnravelIndex 1287, RavelMultiIndex 1365),
tensor/reshape.py (JoinDims 20, SplitDims 151), linalg/decomposition/lu.py:22 ``LU``
(scipy.linalg.lu), signal/conv.py ``Convolve1d`` (np.convolve).  Data-dependent output shapes
(``Repeat``, ``Unique``) cost one host read, like ``Nonzero``.
"""

from __future__ import annotations

import numpy as np

from pytensor_amd import ffi
from pytensor_amd.device import DeviceArray, contiguous_strides, copy_into
from pytensor_amd.dispatch import handler
from pytensor_amd.executor import HostValue


def _cs(shape):
    return contiguous_strides(tuple(shape))


def _reshaped(x: DeviceArray, shape) -> DeviceArray:
    x = x.contiguous()
    return x.view(tuple(shape), _cs(shape))


def _ew(env, body_ops, ins, in_dtypes, out_dtype, shape):
    from pytensor_amd.dispatch.elemwise import launch_elemwise

    body = {"in_dtypes": list(in_dtypes), "out_dtypes": [out_dtype], "body": body_ops, "outs": [["t", len(body_ops) - 1]]}
    (out,), _, _ = launch_elemwise(body, ins, tuple(shape), [out_dtype], None, env)
    return out


def _bview(d: DeviceArray, shape) -> DeviceArray:
    """``d`` broadcast to ``shape`` as a strided view (stride 0 along broadcast axes)"""
    lead = len(shape) - d.ndim
    st = tuple(0 if (s == 1 and t != 1) else q for s, q, t in zip(d.shape, d.strides, shape[lead:]))
    return d.view(tuple(shape), (0,) * lead + st)


def _host_ints(env, v):
    return [int(t) for t in np.asarray(env.to_host(v)).ravel()]


def _as_int64(env, x: DeviceArray) -> DeviceArray:
    if str(x.dtype) == "int64":
        return x.contiguous()
    from pytensor_amd.dispatch.elemwise import _cast

    return _cast(env, x.contiguous(), "int64")


@handler("CpuContiguous")
def cpu_contiguous(node, inputs, env):
    return [env.to_device(inputs[0]).contiguous()]


@handler("JoinDims")
def join_dims(node, inputs, env):
    x = env.to_device(inputs[0])
    a, n = int(node.params["start_axis"]), int(node.params["n_axes"])
    merged = int(np.prod(x.shape[a : a + n], dtype=np.int64)) if n else 1
    return [_reshaped(x, (*x.shape[:a], merged, *x.shape[a + n :]))]


@handler("SplitDims")
def split_dims(node, inputs, env):
    x = env.to_device(inputs[0])
    axis = int(node.params["axis"])
    shape = _host_ints(env, inputs[1])
    if int(np.prod(shape, dtype=np.int64)) != x.shape[axis]:
        raise ValueError(f"cannot reshape array of size {x.size} into shape {(*x.shape[:axis], *shape, *x.shape[axis + 1:])}")
    return [_reshaped(x, (*x.shape[:axis], *shape, *x.shape[axis + 1 :]))]


def _fill_line(env, out: DeviceArray, start: int, step: int, count: int, val):
    """out.flat[start : start + step * count : step] = val (a scalar)"""
    if count <= 0:
        return
    v = env.to_device(val)
    if str(v.dtype) != str(out.dtype):
        from pytensor_amd.dispatch.elemwise import _cast

        v = _cast(env, v.contiguous(), out.dtype)
    copy_into(out.view((count,), (step,), start), v.view((count,), (0,)))


@handler("FillDiagonal")
def fill_diagonal(node, inputs, env):
    a = env.to_device(inputs[0]).contiguous_copy()
    if a.ndim < 2:
        raise ValueError("array must be at least 2-d")
    if a.ndim == 2:
        # extra_ops.py:875-882: a.flat[: w * w : w + 1] = val (rectangular matrices accepted, no wrap)
        h, w = a.shape
        _fill_line(env, a, 0, w + 1, min(h, w), inputs[1])
    else:
        if len(set(a.shape)) != 1:
            raise ValueError("All dimensions of input must be of equal length")
        _fill_line(env, a, 0, sum(_cs(a.shape)), a.shape[0], inputs[1])
    return [a]


@handler("FillDiagonalOffset")
def fill_diagonal_offset(node, inputs, env):
    a = env.to_device(inputs[0]).contiguous_copy()
    offset = int(np.asarray(env.to_host(inputs[2])).item())
    h, w = a.shape
    if offset >= 0:  # extra_ops.py:990-999
        start, steps = offset, min(min(w, h), w - offset)
    else:
        start, steps = -offset * w, min(min(w, h), h + offset)
    # a.flat[start : start + (w + 1) * steps : w + 1] = val, with Python's slice rules (an offset
    # beyond the matrix makes the end negative, which wraps — the reference inherits that)
    line = range(*slice(start, start + (w + 1) * steps, w + 1).indices(a.size))
    if len(line):
        _fill_line(env, a, line[0], w + 1, len(line), inputs[1])
    return [a]


@handler("Bartlett")
def bartlett(node, inputs, env):
    # np.bartlett: n = arange(1 - M, M, 2); where(n <= 0, 1 + n / (M - 1), 1 - n / (M - 1))
    M = int(np.asarray(env.to_host(inputs[0])).item())
    if M < 1:
        return [DeviceArray.empty((0,), "float64")]
    if M == 1:
        return [HostValue(np.ones(1, dtype="float64"))]
    n = DeviceArray.empty((M,), "int64")
    ffi.check(env.lib.pthip_arange(ffi.np_dtype_code(np.dtype("int64")), M, float(1 - M), 2.0, 1 - M, 2, n.ptr))
    d = float(M - 1).hex()
    ops = [{"op": "Cast", "in": [["i", 0]], "dtype": "float64"},
           {"op": "TrueDiv", "in": [["t", 0], ["c", d, "float64"]], "dtype": "float64"},
           {"op": "LE", "in": [["i", 0], ["c", 0, "int64"]], "dtype": "bool"},
           {"op": "Add", "in": [["c", (1.0).hex(), "float64"], ["t", 1]], "dtype": "float64"},
           {"op": "Sub", "in": [["c", (1.0).hex(), "float64"], ["t", 1]], "dtype": "float64"},
           {"op": "Switch", "in": [["t", 2], ["t", 3], ["t", 4]], "dtype": "float64"}]
    return [_ew(env, ops, [n], ["int64"], "float64", (M,))]


@handler("SearchsortedOp")
def searchsorted(node, inputs, env):
    x = env.to_device(inputs[0]).contiguous()
    v = env.to_device(inputs[1]).contiguous()
    sorter = _as_int64(env, env.to_device(inputs[2])) if len(inputs) == 3 else None
    if x.ndim != 1:
        raise ValueError("object too deep for desired array")
    out = DeviceArray.empty(v.shape, "int64")
    if v.size:
        ffi.check(env.lib.pthip_searchsorted(ffi.np_dtype_code(x.dtype), x.shape[0], x.ptr, sorter.ptr if sorter is not None else None,
                                             ffi.np_dtype_code(v.dtype), v.size, v.ptr, int(node.params["side"] == "right"), out.ptr))
    env.keepalive.extend(t for t in (x, v, sorter) if t is not None)
    return [out]


def _take_axis(env, x: DeviceArray, idx: DeviceArray, axis: int) -> DeviceArray:
    """x.take(idx, axis) for an int64 device index vector"""
    order = [axis] + [d for d in range(x.ndim) if d != axis]
    xt = x.view([x.shape[d] for d in order], [x.strides[d] for d in order]).contiguous()
    rest = tuple(xt.shape[1:])
    inner = int(np.prod(rest, dtype=np.int64)) if rest else 1
    out = DeviceArray.empty((idx.size, *rest), x.dtype)
    if out.size:
        ffi.check(env.lib.pthip_take_rows(x.itemsize, idx.size, inner, xt.ptr, xt.shape[0], inner, idx.ptr, out.ptr))
    inv = [order.index(d) for d in range(x.ndim)]
    return out.view([out.shape[d] for d in inv], [out.strides[d] for d in inv])


def _any_outside(env, idx: DeviceArray, n: int) -> bool:
    """any(idx < 0 or idx >= n), decided on the device (one host read of a count)"""
    from pytensor_amd.dispatch.subtensor import nonzero_flat

    if not idx.size:
        return False
    ops = [{"op": "LT", "in": [["i", 0], ["c", 0, "int64"]], "dtype": "bool"}, {"op": "GE", "in": [["i", 0], ["c", int(n), "int64"]], "dtype": "bool"},
           {"op": "OR", "in": [["t", 0], ["t", 1]], "dtype": "bool"}]
    flat = idx.contiguous().view((idx.size,), (1,))
    return nonzero_flat(env, _ew(env, ops, [flat], ["int64"], "bool", (idx.size,))).size > 0


@handler("Repeat")
def repeat(node, inputs, env):
    """np.repeat(x, repeats, axis) (extra_ops.py:706): output position j reads the source position
    searchsorted(cumsum(repeats), j, "right") — cumulative sum, arange and binary search on the
    device; the output length (the last cumulative value) is the one host read a data-dependent
    shape costs."""
    x = env.to_device(inputs[0])
    axis = int(node.params["axis"])
    reps = _as_int64(env, env.to_device(inputs[1]))
    n = x.shape[axis]
    if reps.ndim == 0 or (reps.size == 1 and n != 1):
        reps = _bcast_copy(env, reps.view((1,), (0,)) if reps.ndim == 0 else reps, (n,))
    if tuple(reps.shape) != (n,):
        raise ValueError(f"operands could not be broadcast together with shape ({n},) ({reps.size},)")
    if n == 0:
        return [x.contiguous_copy()]
    if _any_outside(env, reps, 1 << 62):
        raise ValueError("repeats may not contain negative values.")
    ends = DeviceArray.empty((n,), "int64")
    ffi.check(env.lib.pthip_cumulative(ffi.np_dtype_code(np.dtype("int64")), 0, 1, n, 1, reps.contiguous().ptr, ends.ptr))
    total = int(np.asarray(env.to_host(ends.view((1,), (1,), n - 1))).item())
    idx = DeviceArray.empty((total,), "int64")
    if total:
        pos = _iota(env, total)
        ffi.check(env.lib.pthip_searchsorted(ffi.np_dtype_code(np.dtype("int64")), n, ends.ptr, None, ffi.np_dtype_code(np.dtype("int64")),
                                             total, pos.ptr, 1, idx.ptr))
        env.keepalive.extend((ends, pos))
    return [_take_axis(env, x, idx, axis)]


@handler("UnravelIndex")
def unravel_index(node, inputs, env):
    from pytensor_amd.dispatch.subtensor import unravel_flat

    idx = _as_int64(env, env.to_device(inputs[0]))
    dims = _host_ints(env, inputs[1])
    if node.params["order"] == "F":
        comps = unravel_flat(env, idx, dims[::-1])[::-1]
    else:
        comps = unravel_flat(env, idx, dims)
    return [c if c is not idx else idx.contiguous_copy() for c in comps]


@handler("RavelMultiIndex")
def ravel_multi_index(node, inputs, env):
    *multi, dims = inputs
    dims = _host_ints(env, dims)
    if len(multi) != len(dims):
        raise ValueError(f"parameter multi_index must be a sequence of length {len(dims)}")
    devs = [_as_int64(env, env.to_device(m)) for m in multi]
    shape = np.broadcast_shapes(*[tuple(d.shape) for d in devs])
    strides = [1] * len(dims)
    if node.params["order"] == "F":
        for k in range(1, len(dims)):
            strides[k] = strides[k - 1] * dims[k - 1]
    else:
        for k in range(len(dims) - 2, -1, -1):
            strides[k] = strides[k + 1] * dims[k + 1]
    mode = node.params["mode"]
    ops, acc = [], None
    for k in range(len(dims)):
        ref = ["i", k]
        if mode == "wrap":
            ops.append({"op": "Mod", "in": [ref, ["c", max(dims[k], 1), "int64"]], "dtype": "int64"})
            ref = ["t", len(ops) - 1]
        elif mode == "clip":
            ops.append({"op": "Maximum", "in": [ref, ["c", 0, "int64"]], "dtype": "int64"})
            ops.append({"op": "Minimum", "in": [["t", len(ops) - 1], ["c", dims[k] - 1, "int64"]], "dtype": "int64"})
            ref = ["t", len(ops) - 1]
        ops.append({"op": "Mul", "in": [ref, ["c", strides[k], "int64"]], "dtype": "int64"})
        if acc is not None:
            ops.append({"op": "Add", "in": [acc, ["t", len(ops) - 1]], "dtype": "int64"})
        acc = ["t", len(ops) - 1]
    bviews = [_bview(d, shape) for d in devs]
    if mode == "raise":
        for d, n in zip(devs, dims):
            if _any_outside(env, d, n):  # (np.ravel_multi_index checks its input)
                raise ValueError("invalid entry in coordinates array")
    return [_ew(env, ops, bviews, ["int64"] * len(devs), "int64", shape)]


@handler("Unique")
def unique(node, inputs, env):
    """np.unique of the flattened input (extra_ops.py:1229 ``old_np_unique``: the inverse is 1-d):
    sort, flag the first element of every run, compact (csrc/sort.hip, nonzero.hip)."""
    from pytensor_amd.dispatch.subtensor import nonzero_flat

    p = node.params
    if p.get("axis") is not None:
        return _unique_axis(node, inputs, env)
    x = env.to_device(inputs[0]).contiguous()
    n = x.size
    flat = x.view((n,), (1,))
    want = [p["return_index"], p["return_inverse"], p["return_counts"]]
    if n == 0:
        return [flat] + [DeviceArray.empty((0,), "int64") for w in want if w]
    vals = DeviceArray.empty((n,), x.dtype)
    order = DeviceArray.empty((n,), "int64")
    if x.dtype.kind == "b":
        raise NotImplementedError("hip linker: Unique of a bool array")
    ffi.check(env.lib.pthip_sort(ffi.np_dtype_code(x.dtype), 1, n, flat.ptr, vals.ptr, order.ptr))
    dt = str(x.dtype)
    # first[i] = i == 0 or vals[i] differs from vals[i-1]
    first = DeviceArray.empty((n,), "bool")
    ffi.check(env.lib.pthip_memset(first.ptr, 1, 1))
    if n > 1:
        ops = [{"op": "NEQ", "in": [["i", 0], ["i", 1]], "dtype": "bool"}]
        if x.dtype.kind == "f":  # np.unique(equal_nan=True): the NaNs (sorted last) are one value
            ops += [{"op": "IsNan", "in": [["i", 0]], "dtype": "bool"}, {"op": "IsNan", "in": [["i", 1]], "dtype": "bool"},
                    {"op": "AND", "in": [["t", 1], ["t", 2]], "dtype": "bool"}, {"op": "Invert", "in": [["t", 3]], "dtype": "bool"},
                    {"op": "AND", "in": [["t", 0], ["t", 4]], "dtype": "bool"}]
        neq = _ew(env, ops, [vals.view((n - 1,), (1,), 1), vals.view((n - 1,), (1,))], [dt, dt], "bool", (n - 1,))
        copy_into(first.view((n - 1,), (1,), 1), neq)
    starts = nonzero_flat(env, first)  # positions (in sorted order) where a new value starts
    k = starts.size
    uniq = DeviceArray.empty((k,), x.dtype)
    ffi.check(env.lib.pthip_take_rows(x.itemsize, k, 1, vals.ptr, n, 1, starts.ptr, uniq.ptr))
    outs = [uniq]
    if p["return_index"]:
        # stable sort: the first element of a run is the first occurrence
        idx = DeviceArray.empty((k,), "int64")
        ffi.check(env.lib.pthip_take_rows(8, k, 1, order.ptr, n, 1, starts.ptr, idx.ptr))
        outs.append(idx)
    if p["return_inverse"]:
        # rank of every sorted element = (number of run starts up to it) - 1, scattered back by `order`
        from pytensor_amd.dispatch.elemwise import _cast
        from pytensor_amd.dispatch.subtensor import _scatter_rows

        f64 = _cast(env, first, "int64")
        run = DeviceArray.empty((n,), "int64")
        ffi.check(env.lib.pthip_cumulative(ffi.np_dtype_code(np.dtype("int64")), 0, 1, n, 1, f64.ptr, run.ptr))
        rank = _ew(env, [{"op": "Sub", "in": [["i", 0], ["c", 1, "int64"]], "dtype": "int64"}], [run], ["int64"], "int64", (n,))
        inv = DeviceArray.empty((n,), "int64")
        _scatter_rows(env, {"set_instead_of_inc": True}, inv, rank, order)
        outs.append(inv)
    if p["return_counts"]:
        nxt = DeviceArray.empty((k,), "int64")
        if k > 1:
            copy_into(nxt.view((k - 1,), (1,)), starts.view((k - 1,), (1,), 1))
        copy_into(nxt.view((1,), (1,), k - 1), env.to_device(HostValue(np.asarray([n], dtype="int64"))))
        outs.append(_ew(env, [{"op": "Sub", "in": [["i", 0], ["i", 1]], "dtype": "int64"}], [nxt, starts], ["int64", "int64"], "int64", (k,)))
    env.keepalive.extend((vals, order, first))
    return outs


MAX_UNIQUE_WIDTH = 4096  # elements per slice of Unique(axis=...): one stable sort pass per element


def _unique_axis(node, inputs, env):
    """np.unique(x, axis=a): the slices along ``a`` as rows, ordered lexicographically by one stable
    device sort per column (last column first — an LSD radix sort whose digits are whole columns), runs
    of equal rows flagged, compacted like the flat case."""
    from pytensor_amd.dispatch.elemwise import _cast
    from pytensor_amd.dispatch.subtensor import _scatter_rows, nonzero_flat

    p = node.params
    axis = int(p["axis"])
    x = env.to_device(inputs[0])
    if x.dtype.kind == "b":
        raise NotImplementedError("hip linker: Unique of a bool array")
    order_ax = [axis] + [d for d in range(x.ndim) if d != axis]
    xt = x.view([x.shape[d] for d in order_ax], [x.strides[d] for d in order_ax]).contiguous()
    n = xt.shape[0]
    rest = tuple(xt.shape[1:])
    m = int(np.prod(rest, dtype=np.int64)) if rest else 1
    if m > MAX_UNIQUE_WIDTH:
        raise NotImplementedError(f"hip linker: Unique(axis=...) over slices of {m} elements (up to {MAX_UNIQUE_WIDTH})")
    want = [p["return_index"], p["return_inverse"], p["return_counts"]]

    def back(rows: DeviceArray, k: int) -> DeviceArray:
        full = rows.view((k, *rest), _cs((k, *rest)))
        inv = [order_ax.index(d) for d in range(x.ndim)]
        return full.view([full.shape[d] for d in inv], [full.strides[d] for d in inv]).contiguous()

    if n == 0 or m == 0:
        k = 0 if n == 0 else 1
        outs = [back(DeviceArray.empty((k, m), x.dtype), k)]
        if p["return_index"]:
            outs.append(env.to_device(HostValue(np.zeros(k, dtype="int64"))))
        if p["return_inverse"]:
            outs.append(env.to_device(HostValue(np.zeros(n, dtype="int64"))))
        if p["return_counts"]:
            outs.append(env.to_device(HostValue(np.full(k, n, dtype="int64"))))
        return outs
    x2 = xt.view((n, m), (m, 1))
    dt = str(x.dtype)
    code = ffi.np_dtype_code(x.dtype)
    order = _iota(env, n)
    keys, idx = DeviceArray.empty((n,), x.dtype), DeviceArray.empty((n,), "int64")
    for col in range(m - 1, -1, -1):
        colv = x2.view((n,), (m,), col).contiguous()  # this column, in the original row order
        ffi.check(env.lib.pthip_take_rows(x.itemsize, n, 1, colv.ptr, n, 1, order.ptr, keys.ptr))
        ffi.check(env.lib.pthip_sort(code, 1, n, keys.ptr, None, idx.ptr))
        nxt = DeviceArray.empty((n,), "int64")
        ffi.check(env.lib.pthip_take_rows(8, n, 1, order.ptr, n, 1, idx.ptr, nxt.ptr))
        env.keepalive.extend((colv, order))
        order = nxt
    S = DeviceArray.empty((n, m), x.dtype)
    ffi.check(env.lib.pthip_take_rows(x.itemsize, n, m, x2.ptr, n, m, order.ptr, S.ptr))
    first = DeviceArray.empty((n,), "bool")
    ffi.check(env.lib.pthip_memset(first.ptr, 1, 1))
    if n > 1:
        # a row starts a run when any of its elements differs from the row above
        neq = _ew(env, [{"op": "NEQ", "in": [["i", 0], ["i", 1]], "dtype": "bool"}, {"op": "Cast", "in": [["t", 0]], "dtype": "float64"}],
                  [S.view((n - 1, m), (m, 1), m), S.view((n - 1, m), (m, 1))], [dt, dt], "float64", (n - 1, m))
        from pytensor_amd.dispatch.decomp import _column_sums, _t

        diff = _column_sums(env, _t(neq).contiguous())  # (n - 1,): how many elements differ
        flags = _ew(env, [{"op": "GT", "in": [["i", 0], ["c", (0.0).hex(), "float64"]], "dtype": "bool"}], [diff], ["float64"], "bool", (n - 1,))
        copy_into(first.view((n - 1,), (1,), 1), flags)
    starts = nonzero_flat(env, first)
    k = starts.size
    uniq = DeviceArray.empty((k, m), x.dtype)
    ffi.check(env.lib.pthip_take_rows(x.itemsize, k, m, S.ptr, n, m, starts.ptr, uniq.ptr))
    outs = [back(uniq, k)]
    if p["return_index"]:
        first_idx = DeviceArray.empty((k,), "int64")
        ffi.check(env.lib.pthip_take_rows(8, k, 1, order.ptr, n, 1, starts.ptr, first_idx.ptr))
        outs.append(first_idx)
    if p["return_inverse"]:
        f64 = _cast(env, first, "int64")
        run = DeviceArray.empty((n,), "int64")
        ffi.check(env.lib.pthip_cumulative(ffi.np_dtype_code(np.dtype("int64")), 0, 1, n, 1, f64.ptr, run.ptr))
        rank = _ew(env, [{"op": "Sub", "in": [["i", 0], ["c", 1, "int64"]], "dtype": "int64"}], [run], ["int64"], "int64", (n,))
        inv = DeviceArray.empty((n,), "int64")
        _scatter_rows(env, {"set_instead_of_inc": True}, inv, rank, order)
        outs.append(inv)
    if p["return_counts"]:
        nxt = DeviceArray.empty((k,), "int64")
        if k > 1:
            copy_into(nxt.view((k - 1,), (1,)), starts.view((k - 1,), (1,), 1))
        copy_into(nxt.view((1,), (1,), k - 1), env.to_device(HostValue(np.asarray([n], dtype="int64"))))
        outs.append(_ew(env, [{"op": "Sub", "in": [["i", 0], ["i", 1]], "dtype": "int64"}], [nxt, starts], ["int64", "int64"], "int64", (k,)))
    env.keepalive.extend((S, order, first, keys, idx))
    return outs


@handler("LU")
def lu(node, inputs, env):
    """scipy.linalg.lu (lu.py:77-89): A = P L U from the packed getrf factors; ``p_indices`` gives the
    row order p with A = L[p] U, ``permute_l`` the product P L."""
    from pytensor_amd.dispatch.decomp import _triu
    from pytensor_amd.dispatch.lu import getrf_device

    a = env.to_device(inputs[0])
    if a.ndim != 2 or a.shape[0] != a.shape[1]:
        raise NotImplementedError("hip linker: LU of a rectangular or batched matrix (Blockwise loops the items)")
    n = a.shape[0]
    LU, perm, _, _, _ = getrf_device(env, a)
    packed = LU.view((n, n), (n, 1))
    L = _triu(env, packed, n, lower=True, unit=True)
    U = _triu(env, packed, n)
    gather = perm.view((n,), (1,))  # row i of L U is row gather[i] of A
    # scipy's p / P: the inverse of that order (A = (L U)[p])
    inv = DeviceArray.empty((n,), "int64")
    if n:
        from pytensor_amd.dispatch.subtensor import _scatter_rows

        ar = DeviceArray.empty((n,), "int64")
        ffi.check(env.lib.pthip_arange(ffi.np_dtype_code(np.dtype("int64")), n, 0.0, 1.0, 0, 1, ar.ptr))
        _scatter_rows(env, {"set_instead_of_inc": True}, inv, ar, gather.contiguous())
    p = node.params
    if p["permute_l"]:
        return [_take_axis(env, L, inv, 0).contiguous(), U]
    if p["p_indices"]:
        from pytensor_amd.dispatch.elemwise import _cast

        return [_cast(env, inv, "int32"), L, U]
    eye = DeviceArray.empty((n, n), a.dtype)
    if n:
        ffi.check(env.lib.pthip_eye(ffi.np_dtype_code(a.dtype), n, n, 0, eye.ptr))
    return [_take_axis(env, eye, inv, 0).contiguous(), L, U]


@handler("Convolve1d")
def convolve1d(node, inputs, env):
    a, b = (env.to_device(i) for i in inputs[:2])
    full = bool(np.asarray(env.to_host(inputs[2])).item())
    if a.ndim != 1 or b.ndim != 1:
        raise ValueError("object too deep for desired array")
    if a.shape[0] == 0 or b.shape[0] == 0:
        raise ValueError("v cannot be empty" if b.shape[0] == 0 else "a cannot be empty")
    dt = np.result_type(np.dtype(a.dtype), np.dtype(b.dtype))
    kdt = dt if dt.name in ("float64", "float32", "int64") else np.dtype("int64" if dt.kind in "iub" else "float64")
    if dt.kind == "b":
        raise NotImplementedError("hip linker: Convolve1d of bool arrays")
    from pytensor_amd.dispatch.elemwise import _cast

    a, b = (t.contiguous() if np.dtype(t.dtype) == kdt else _cast(env, t.contiguous(), kdt) for t in (a, b))
    if b.shape[0] > a.shape[0]:
        a, b = b, a  # np.convolve swaps so that the first operand is the longer one
    na, nb = a.shape[0], b.shape[0]
    out = DeviceArray.empty((na + nb - 1 if full else na - nb + 1,), kdt)
    ffi.check(env.lib.pthip_convolve1d(ffi.np_dtype_code(kdt), na, a.ptr, nb, b.ptr, int(full), out.ptr))
    env.keepalive.extend((a, b))
    return [out if kdt == dt else _cast(env, out, dt)]


@handler("Convolve2d")
def convolve2d(node, inputs, env):
    # signal/conv.py:260-263: scipy.signal.convolve(in1, in2, "full" | "valid") — direct sums
    a, b = (env.to_device(i) for i in inputs[:2])
    full = bool(np.asarray(env.to_host(inputs[2])).item())
    if a.ndim != 2 or b.ndim != 2:
        raise ValueError("convolve2d inputs must both be 2-D arrays")
    dt = np.result_type(np.dtype(a.dtype), np.dtype(b.dtype))
    if dt.kind == "b":
        raise NotImplementedError("hip linker: Convolve2d of bool arrays")
    kdt = dt if dt.name in ("float64", "float32", "int64") else np.dtype("int64" if dt.kind in "iu" else "float64")
    from pytensor_amd.dispatch.elemwise import _cast

    a, b = (t.contiguous() if np.dtype(t.dtype) == kdt else _cast(env, t.contiguous(), kdt) for t in (a, b))
    (ha, wa), (hb, wb) = a.shape, b.shape
    if 0 in (ha, wa, hb, wb):
        raise ValueError("convolve2d: empty input")
    if full:
        shape = (ha + hb - 1, wa + wb - 1)
    else:
        if not ((ha >= hb and wa >= wb) or (hb >= ha and wb >= wa)):
            raise ValueError("For 'valid' mode, one must be at least as large as the other in every dimension")
        shape = (abs(ha - hb) + 1, abs(wa - wb) + 1)
    out = DeviceArray.empty(shape, kdt)
    ffi.check(env.lib.pthip_convolve2d(ffi.np_dtype_code(kdt), ha, wa, a.ptr, hb, wb, b.ptr, int(full), out.ptr))
    env.keepalive.extend((a, b))
    return [out if kdt == dt else _cast(env, out, dt)]


def _iota(env, n: int) -> DeviceArray:
    out = DeviceArray.empty((n,), "int64")
    if n:
        ffi.check(env.lib.pthip_arange(ffi.np_dtype_code(np.dtype("int64")), n, 0.0, 1.0, 0, 1, out.ptr))
    return out


def _bcast_copy(env, x: DeviceArray, shape) -> DeviceArray:
    if tuple(x.shape) == tuple(shape) and x.is_contiguous():
        return x
    out = DeviceArray.empty(tuple(shape), x.dtype)
    if out.size:
        copy_into(out, _bview(x, shape))
    return out


@handler("Choose")
def choose(node, inputs, env):
    """np.choose(a, choices, mode) with ``choices`` one tensor (K, ...): out[i] = choices[a[i]][i]
    (tensor/basic.py:4135; typed-list choices are not lowered)"""
    a = _as_int64(env, env.to_device(inputs[0]))
    ch = env.to_device(inputs[1])
    if ch.ndim < 1:
        raise ValueError("choices must have at least one dimension")
    K = ch.shape[0]
    shape = tuple(np.broadcast_shapes(tuple(a.shape), tuple(ch.shape[1:])))
    size = int(np.prod(shape, dtype=np.int64)) if shape else 1
    ab = _bcast_copy(env, a, shape)
    cb = _bcast_copy(env, ch, (K, *shape))
    mode = node.params["mode"]
    if mode == "raise" and _any_outside(env, ab, K):
        raise ValueError("invalid entry in choice array")
    ops = []
    ref = ["i", 0]
    if mode == "wrap":
        ops.append({"op": "Mod", "in": [ref, ["c", max(K, 1), "int64"]], "dtype": "int64"})
        ref = ["t", 0]
    elif mode == "clip":
        ops.append({"op": "Maximum", "in": [ref, ["c", 0, "int64"]], "dtype": "int64"})
        ops.append({"op": "Minimum", "in": [["t", 0], ["c", K - 1, "int64"]], "dtype": "int64"})
        ref = ["t", 1]
    ops.append({"op": "Mul", "in": [ref, ["c", size, "int64"]], "dtype": "int64"})
    ops.append({"op": "Add", "in": [["t", len(ops) - 1], ["i", 1]], "dtype": "int64"})
    out = DeviceArray.empty(shape, ch.dtype)
    if size:
        flat = _ew(env, ops, [ab.view((size,), (1,)), _iota(env, size)], ["int64", "int64"], "int64", (size,))
        ffi.check(env.lib.pthip_take_rows(ch.itemsize, size, 1, cb.ptr, K * size, 1, flat.ptr, out.ptr))
        env.keepalive.extend((cb, flat))
    return [out]


@handler("PermuteRowElements")
def permute_row_elements(node, inputs, env):
    """tensor/basic.py:3426: out[..., i] = x[..., y[..., i]] (inverse: out[..., y[..., i]] = x[..., i]),
    the leading dimensions of x and y broadcast against each other"""
    from pytensor_amd.dispatch.subtensor import _scatter_rows

    x = env.to_device(inputs[0])
    y = _as_int64(env, env.to_device(inputs[1]))
    nd = max(x.ndim, y.ndim)
    xs = (1,) * (nd - x.ndim) + tuple(x.shape)
    ys = (1,) * (nd - y.ndim) + tuple(y.shape)
    if xs[-1] != ys[-1]:
        raise ValueError(f"Dimension mismatch: {xs[-1]}, {ys[-1]}")
    shape = tuple(np.broadcast_shapes(xs, ys))
    L = shape[-1]
    size = int(np.prod(shape, dtype=np.int64))
    xb = _bcast_copy(env, x.view(xs, (0,) * (nd - x.ndim) + tuple(x.strides)), shape)
    yb = _bcast_copy(env, y.view(ys, (0,) * (nd - y.ndim) + tuple(y.strides)), shape)
    out = DeviceArray.empty(shape, x.dtype)
    if size == 0:
        return [out]
    # flat index of the addressed element: (position // L) * L + y
    ops = [{"op": "IntDiv", "in": [["i", 1], ["c", L, "int64"]], "dtype": "int64"},
           {"op": "Mul", "in": [["t", 0], ["c", L, "int64"]], "dtype": "int64"},
           {"op": "Add", "in": [["t", 1], ["i", 0]], "dtype": "int64"}]
    flat = _ew(env, ops, [yb.view((size,), (1,)), _iota(env, size)], ["int64", "int64"], "int64", (size,))
    if node.params["inverse"]:
        _scatter_rows(env, {"set_instead_of_inc": True}, out.view((size,), (1,)), xb.view((size,), (1,)), flat)
    else:
        ffi.check(env.lib.pthip_take_rows(x.itemsize, size, 1, xb.ptr, size, 1, flat.ptr, out.ptr))
    env.keepalive.extend((xb, flat))
    return [out]
