"""Indexing ops (bit-exact tier): ``Subtensor``, ``IncSubtensor``, ``AdvancedSubtensor``,
``AdvancedIncSubtensor``.

Reference: pytensor/tensor/subtensor.py — Subtensor 868 (perform 912-917: a NumPy
view), IncSubtensor 1441, AdvancedSubtensor 1932, AdvancedIncSubtensor 2275
(``np.add.at`` semantics for ``inc``).  Basic indexing is descriptor arithmetic on
the host (no kernel); integer-array indexing on one axis maps to the row
gather/scatter kernels.
"""

from __future__ import annotations

import numpy as np

from pytensor_amd import ffi
from pytensor_amd.device import DeviceArray, contiguous_strides, copy_into
from pytensor_amd.dispatch import handler
from pytensor_amd.executor import HostValue


def _resolve(idx_list, index_inputs, env):
    """idx_list with input positions → concrete Python ints / slices (host)."""

    def val(e):
        if e is None:
            return None
        v = index_inputs[e]
        a = env.to_host(v)
        return int(a)

    out = []
    for e in idx_list:
        if isinstance(e, slice):
            out.append(slice(val(e.start), val(e.stop), val(e.step)))
        else:
            out.append(val(e))
    return out


def basic_view(x: DeviceArray, index) -> DeviceArray:
    """NumPy basic indexing as a strided view."""
    if len(index) > x.ndim:
        raise IndexError("too many indices for array")
    off = 0
    shape, strides = [], []
    for d in range(x.ndim):
        n, st = x.shape[d], x.strides[d]
        e = index[d] if d < len(index) else slice(None)
        if isinstance(e, slice):
            start, stop, step = e.indices(n)
            ln = len(range(start, stop, step))
            off += start * st if ln else 0
            shape.append(ln)
            strides.append(st * step)
        else:
            i = int(e)
            if i < -n or i >= n:
                raise IndexError(f"index {i} is out of bounds for axis {d} with size {n}")
            if i < 0:
                i += n
            off += i * st
    return x.view(shape, strides, off)


@handler("Subtensor")
def subtensor(node, inputs, env):
    x, *idx = inputs
    index = _resolve(node.params["idx_list"], idx, env)
    if isinstance(x, HostValue):
        return [HostValue(np.asarray(x.a[tuple(index)]))]
    return [basic_view(x, index)]


def _own_copy(env, x: DeviceArray, pos: int) -> DeviceArray:
    """A buffer this node may overwrite: ``x`` itself when the executor donated it
    (fresh, single consumer — executor._compute_donations), else a copy."""
    if pos in env.donated and x.is_contiguous() and x.offset == 0:
        return x
    out = DeviceArray.empty(x.shape, x.dtype)
    copy_into(out, x)
    return out


def _add_into(env, view: DeviceArray, y: DeviceArray):
    """view[...] += broadcast(y) through a fused add + strided write-back."""
    from pytensor_amd.dispatch.elemwise import launch_elemwise

    dt = str(view.dtype)
    body = {
        "in_dtypes": [dt, str(y.dtype)],
        "out_dtypes": [dt],
        "body": [{"op": "Add", "in": [["i", 0], ["i", 1]], "dtype": dt}],
        "outs": [["t", 0]],
    }
    nd = view.ndim
    yv = y.view((1,) * (nd - y.ndim) + y.shape, (0,) * (nd - y.ndim) + y.strides)
    for d in range(nd):
        if yv.shape[d] != view.shape[d] and yv.shape[d] != 1:
            raise ValueError(f"shape mismatch: value array of shape {y.shape} could not be broadcast to {view.shape}")
    outs, _, _ = launch_elemwise(body, [view, yv], view.shape, [dt], None, env)
    copy_into(view, outs[0])


@handler("IncSubtensor")
def inc_subtensor(node, inputs, env):
    x, y, *idx = inputs
    x, y = env.to_device(x), env.to_device(y)
    index = _resolve(node.params["idx_list"], idx, env)
    out = _own_copy(env, x, 0)
    view = basic_view(out, index)
    if y.ndim > view.ndim:
        raise ValueError("IncSubtensor: value has more dimensions than the indexed region")
    if view.size == 0:
        return [out]
    if node.params["set_instead_of_inc"]:
        if str(y.dtype) != str(out.dtype):
            from pytensor_amd.dispatch.elemwise import _cast

            y = _cast(env, y, out.dtype)
        copy_into(view, y)
    else:
        _add_into(env, view, y)
    return [out]


def _combine_indices(env, ivs, dims):
    """Row-major linear index of pointwise index vectors over the leading ``dims`` (negative
    entries wrap like NumPy; any out-of-range component poisons the row so that the gather /
    scatter kernel raises the device error flag -> IndexError)."""
    from pytensor_amd.dispatch.elemwise import launch_elemwise
    from pytensor_amd.executor import HostValue

    ivs = [_index_on_device(env, iv) for iv in ivs]
    try:
        shape = np.broadcast_shapes(*[iv.shape for iv in ivs])
    except ValueError:
        raise IndexError(f"shape mismatch: indexing arrays could not be broadcast together with shapes {[i.shape for i in ivs]}") from None
    nd = len(shape)
    # NumPy broadcasting of the index arrays: left-pad to the common rank, stride 0 on length-1 dims
    ins = [iv.view((1,) * (nd - iv.ndim) + tuple(iv.shape), (0,) * (nd - iv.ndim) + tuple(iv.strides)) for iv in ivs]
    k = len(ivs)
    ins += [HostValue(np.asarray(int(n), dtype="int64")) for n in dims]
    body = []

    def emit(op, args, dt="int64"):
        body.append({"op": op, "in": args, "dtype": dt})
        return ["t", len(body) - 1]

    zero = ["c", 0, "int64"]
    lin = valid = None
    for d in range(k):
        i, n = ["i", d], ["i", k + d]
        w = emit("Switch", [emit("LT", [i, zero], "bool"), emit("Add", [i, n]), i])
        ok = emit("AND", [emit("GE", [w, zero], "bool"), emit("LT", [w, n], "bool")], "bool")
        valid = ok if valid is None else emit("AND", [valid, ok], "bool")
        lin = w if lin is None else emit("Add", [emit("Mul", [lin, n]), w])
    out = emit("Switch", [valid, lin, ["c", -(1 << 62), "int64"]])
    sb = {"in_dtypes": ["int64"] * (2 * k), "out_dtypes": ["int64"], "body": body, "outs": [out]}
    outs, _, _ = launch_elemwise(sb, ins, tuple(shape), ["int64"], None, env)
    return outs[0]


def nonzero_flat(env, mask) -> DeviceArray:
    """Ascending flat (C-order) int64 indices of the non-zero elements of ``mask``
    (csrc/nonzero.hip).  The length is data dependent: one host read of the count."""
    m = env.to_device(mask)
    if m.dtype.kind != "b":
        from pytensor_amd.dispatch.elemwise import launch_elemwise

        dt = str(m.dtype)
        body = {"in_dtypes": [dt], "out_dtypes": ["bool"],
                "body": [{"op": "NEQ", "in": [["i", 0], ["c", 0 if m.dtype.kind in "iu" else (0.0).hex(), dt]], "dtype": "bool"}],
                "outs": [["t", 0]]}
        outs, _, _ = launch_elemwise(body, [m], m.shape, ["bool"], None, env)
        m = outs[0]
    mc = m.contiguous()
    n = mc.size
    idx = DeviceArray.empty((max(n, 1),), "int64")
    cnt = DeviceArray.empty((1,), "int64")
    ffi.check(env.lib.pthip_nonzero(n, mc.ptr, idx.ptr, cnt.ptr))
    k = int(np.asarray(env.to_host(cnt)).ravel()[0])
    return idx.view((k,), (1,))


def unravel_flat(env, flat: DeviceArray, shape):
    """the per-axis components of C-order flat indices (np.unravel_index), on the device"""
    if len(shape) == 1:
        return [flat]
    from pytensor_amd.dispatch.elemwise import launch_elemwise

    comps, stride = [None] * len(shape), 1
    for d in range(len(shape) - 1, -1, -1):
        body = {"in_dtypes": ["int64"], "out_dtypes": ["int64"],
                "body": [{"op": "IntDiv", "in": [["i", 0], ["c", int(stride), "int64"]], "dtype": "int64"},
                         {"op": "Mod", "in": [["t", 0], ["c", max(int(shape[d]), 1), "int64"]], "dtype": "int64"}],
                "outs": [["t", 1]]}
        if flat.size:
            outs, _, _ = launch_elemwise(body, [flat], flat.shape, ["int64"], None, env)
            comps[d] = outs[0]
        else:
            comps[d] = flat
        stride *= int(shape[d])
    return comps


def _is_bool_index(v):
    return getattr(v, "dtype", None) is not None and np.dtype(v.dtype).kind == "b"


def _expand_bool_masks(env, idx_list, idx, x_shape):
    """NumPy semantics: a boolean index over m axes stands for the m integer vectors of its
    ``nonzero()``.  Returns an (idx_list, index values) pair without boolean entries."""
    if not any((not isinstance(e, slice)) and _is_bool_index(idx[e]) for e in idx_list):
        return idx_list, idx
    new_list, new_idx, d = [], [], 0
    for e in idx_list:
        if isinstance(e, slice):
            # (a slice's start / stop / step are positions in `idx` too: they move with the renumbering —
            #  `x[1:, mask]`, reference test tests/tensor/test_subtensor.py::TestSubtensor::test_boolean)
            comps = []
            for c in (e.start, e.stop, e.step):
                if c is None:
                    comps.append(None)
                else:
                    comps.append(len(new_idx))
                    new_idx.append(idx[c])
            new_list.append(slice(*comps))
            d += 1
            continue
        v = idx[e]
        if _is_bool_index(v):
            mv = env.to_device(v)
            m = mv.ndim
            if m == 0:
                raise NotImplementedError("hip linker: 0-d boolean index")
            if tuple(mv.shape) != tuple(x_shape[d : d + m]):
                raise IndexError(f"boolean index did not match indexed array along axis {d}; size of axis is "
                                 f"{tuple(x_shape[d:d + m])} but size of corresponding boolean axis is {tuple(mv.shape)}")
            for c in unravel_flat(env, nonzero_flat(env, mv), mv.shape):
                new_list.append(len(new_idx))
                new_idx.append(c)
            d += m
        else:
            new_list.append(len(new_idx))
            new_idx.append(v)
            d += 1
    return new_list, new_idx


@handler("Nonzero")
def nonzero(node, inputs, env):
    # Nonzero.perform (pytensor/tensor/basic.py): np.nonzero -> one int64 vector per axis
    x = env.to_device(inputs[0])
    if x.ndim == 0:
        raise ValueError("Nonzero only supports non-scalar arrays.")
    return unravel_flat(env, nonzero_flat(env, x), x.shape)


def _index_on_device(env, iv):
    iv = env.to_device(iv)
    if iv.dtype.kind == "b":
        raise NotImplementedError("hip linker: boolean index outside AdvancedSubtensor/AdvancedIncSubtensor")
    if str(iv.dtype) != "int64":
        from pytensor_amd.dispatch.elemwise import _cast

        iv = _cast(env, iv, "int64")
    return iv.contiguous()


def _simple_axis0(idx_list):
    """``x[iv]`` / ``x[iv, :, ...]`` with full slices — the row gather/scatter kernels apply directly."""
    return (len(idx_list) >= 1 and not isinstance(idx_list[0], slice)
            and all(isinstance(e, slice) and (e.start, e.stop, e.step) == (None, None, None) for e in idx_list[1:]))


def _general_plan(env, x: DeviceArray, idx_list, idx):
    """NumPy advanced indexing in general (subtensor.py:1932 ``x.__getitem__(tuple(indices))``):
    basic slices are applied first as a view; the advanced axes are then moved to the front,
    flattened, and addressed with ONE combined row index (N-d index arrays broadcast against each
    other).  Returns (view with the advanced axes leading, number k of advanced axes, combined
    index of the broadcast shape, position at which NumPy places the broadcast dims — 0 when the
    advanced axes are separated by a slice — and the axis order used)."""
    basic, adv = [], []
    for d, e in enumerate(idx_list):
        if isinstance(e, slice):
            basic.append(slice(*(None if v is None else int(env.to_host(idx[v])) for v in (e.start, e.stop, e.step))))
        else:
            basic.append(slice(None))
            adv.append(d)
    if not adv:
        raise NotImplementedError("hip linker: advanced indexing without an integer index")
    xv = basic_view(x, basic)
    order = adv + [d for d in range(xv.ndim) if d not in adv]
    xt = xv.view([xv.shape[d] for d in order], [xv.strides[d] for d in order])
    k = len(adv)
    ivs = [idx[idx_list[d]] for d in adv]
    if k == 1:
        iv = _index_on_device(env, ivs[0])
    else:
        iv = _combine_indices(env, ivs, xt.shape[:k])
    adjacent = adv == list(range(adv[0], adv[0] + k))
    return xt, k, iv, (adv[0] if adjacent else 0), order


@handler("AdvancedSubtensor")
def advanced_subtensor(node, inputs, env):
    x, *idx = inputs
    x = env.to_device(x)
    idx_list, idx = _expand_bool_masks(env, node.params["idx_list"], idx, x.shape)
    xt, k, iv, place, _ = _general_plan(env, x, idx_list, idx)
    rest = tuple(xt.shape[k:])
    rows = int(np.prod(xt.shape[:k], dtype=np.int64))
    if k > 1:
        # flatten the k leading axes: they must be jointly contiguous
        xt = xt.contiguous()
        xf = xt.view((rows, *rest), contiguous_strides((rows, *rest)))
    else:
        xf = xt
    inner = int(np.prod(rest)) if rest else 1
    # rows must be contiguous runs of `inner` elements
    if inner > 1 and not xf.view(rest, xf.strides[1:]).is_contiguous():
        xf = xf.contiguous()
    n_idx = iv.size
    out = DeviceArray.empty((n_idx, *rest), x.dtype)
    if out.size:
        ffi.check(
            env.lib.pthip_take_rows(x.itemsize, n_idx, inner, xf.ptr, xf.shape[0], xf.strides[0] if xf.shape[0] > 1 else inner, iv.ptr, out.ptr)
        )
    elif n_idx and xf.shape[0] == 0:
        raise IndexError("index out of bounds for axis with size 0")
    res_shape = (*iv.shape, *rest)
    res = out.view(res_shape, contiguous_strides(res_shape))
    if place != 0:
        # NumPy places the broadcast index dims where the (adjacent) advanced axes were
        nb = len(iv.shape)
        order = list(range(nb, nb + place)) + list(range(nb)) + list(range(nb + place, len(res_shape)))
        res = res.view([res.shape[d] for d in order], [res.strides[d] for d in order])
    return [res]


@handler("AdvancedIncSubtensor")
def advanced_inc_subtensor(node, inputs, env):
    p = node.params
    x, y, *idx = inputs
    x, y = env.to_device(x), env.to_device(y)
    idx_list, idx = _expand_bool_masks(env, p["idx_list"], idx, x.shape)
    own = _own_copy(env, x, 0)
    if _simple_axis0(idx_list):
        iv = _index_on_device(env, idx[idx_list[0]])
        if iv.ndim == 1:
            if list(p["idx_list"]) == [0] and not _is_bool_index(inputs[2]):
                # AdvancedIncSubtensor._check_runtime_broadcast_of_vector_index (subtensor.py:2448-2473)
                static = env.graph.vars[node.inputs[1]].shape
                expected = (iv.shape[0], *x.shape[1:])
                for sd, yd, ed in zip(reversed(static), reversed(y.shape), reversed(expected)):
                    if sd != 1 and yd == 1 and ed != 1:
                        raise ValueError(
                            "Runtime broadcasting not allowed. AdvancedIncSubtensor was asked to broadcast the second input (y) "
                            "along a dimension that was not marked as broadcastable. If broadcasting was intended, use "
                            "`specify_broadcastable` on the relevant dimension(s).")
            return [_scatter_rows(env, p, own, y, iv)]
    xt, k, iv, place, _ = _general_plan(env, own, idx_list, idx)
    rest = tuple(xt.shape[k:])
    rows = int(np.prod(xt.shape[:k], dtype=np.int64))
    flat_shape = (rows, *rest)
    direct = xt.is_contiguous()
    work = xt if direct else xt.contiguous_copy()
    flat = work.view(flat_shape, contiguous_strides(flat_shape))
    # y broadcasts against NumPy's result shape; bring it to (index dims..., remaining dims...)
    nb = len(iv.shape)
    res_shape = (*rest[:place], *iv.shape, *rest[place:]) if place else (*iv.shape, *rest)
    if y.ndim > len(res_shape):
        raise ValueError(f"shape mismatch: value array of shape {y.shape} could not be broadcast to indexing result of shape {res_shape}")
    yv = y.view((1,) * (len(res_shape) - y.ndim) + tuple(y.shape), (0,) * (len(res_shape) - y.ndim) + tuple(y.strides))
    for d in range(len(res_shape)):
        if yv.shape[d] not in (1, res_shape[d]):
            raise ValueError(f"shape mismatch: value array of shape {y.shape} could not be broadcast to indexing result of shape {res_shape}")
    yb = yv.view(res_shape, [0 if yv.shape[d] == 1 and res_shape[d] != 1 else yv.strides[d] for d in range(len(res_shape))])
    if place:
        order = list(range(place, place + nb)) + list(range(place)) + list(range(place + nb, len(res_shape)))
        yb = yb.view([yb.shape[d] for d in order], [yb.strides[d] for d in order])
    n_idx = iv.size
    if n_idx and flat.size:
        ydt = yb
        if str(ydt.dtype) != str(own.dtype):
            from pytensor_amd.dispatch.elemwise import _cast

            ydt = _cast(env, ydt.contiguous(), own.dtype)
        yfull = DeviceArray.empty((*iv.shape, *rest), own.dtype)
        copy_into(yfull, ydt)
        y2 = yfull.view((n_idx, *rest), contiguous_strides((n_idx, *rest)))
        _scatter_rows(env, p, flat, y2, iv.view((n_idx,), (1,)))
        if not direct:
            copy_into(xt, work)
    return [own]


def _scatter_rows(env, p, out, y, iv):
    """``out[iv] (+)= y`` on the leading axis of the (owned, contiguous) ``out``."""
    x = out
    inner_shape = x.shape[1:]
    inner = int(np.prod(inner_shape)) if inner_shape else 1
    n_idx = iv.size
    if n_idx == 0 or inner == 0:
        return out
    if not p["set_instead_of_inc"] and str(x.dtype) in ("int8", "int16", "uint8", "uint16"):
        # the device adds 4- and 8-byte words: accumulate in 32 bits and narrow again — the same value modulo 2^8 / 2^16 as
        # NumPy's wrap-around add (reference test tests/tensor/test_extra_ops.py TestBinCount[int8 … uint16]: bincount is
        # an inc_subtensor of ones into zeros)
        from pytensor_amd.dispatch.elemwise import _cast

        wide = "uint32" if str(x.dtype).startswith("u") else "int32"
        acc = _scatter_rows(env, p, _cast(env, out, wide), _cast(env, y, wide), iv)
        copy_into(out, _cast(env, acc, x.dtype))
        return out
    if str(y.dtype) != str(x.dtype):
        from pytensor_amd.dispatch.elemwise import _cast

        y = _cast(env, y, x.dtype)
    # y broadcasts against (n_idx, *inner_shape)
    tgt = (n_idx, *inner_shape)
    yv = y.view((1,) * (len(tgt) - y.ndim) + y.shape, (0,) * (len(tgt) - y.ndim) + y.strides)
    for d in range(len(tgt)):
        if yv.shape[d] not in (1, tgt[d]):
            raise ValueError(f"shape mismatch: value array of shape {y.shape} could not be broadcast to indexing result of shape {tgt}")
    row_bcast = yv.shape[0] == 1 and n_idx != 1
    inner_view = yv.view(yv.shape[1:], yv.strides[1:])
    if inner_view.shape == tuple(inner_shape) and inner_view.is_contiguous() and (row_bcast or inner == 1 or yv.strides[0] == inner or n_idx == 1):
        ysrc = yv
        ys0 = 0 if row_bcast else (yv.strides[0] if n_idx > 1 else inner)
    else:
        ysrc = DeviceArray.empty(tgt, x.dtype)
        copy_into(ysrc, yv)
        ys0 = inner
    lib = env.lib
    ws_bytes = lib.pthip_scatter_rows_workspace(n_idx, x.shape[0], inner)
    ws = DeviceArray.empty((ws_bytes,), "uint8") if ws_bytes else None
    inc = 0 if p["set_instead_of_inc"] else 1
    if inc and p.get("ignore_duplicates"):
        # `out[idx] += y` as NumPy's buffered fancy in-place add (subtensor.py AdvancedIncSubtensor.perform,
        # `ignore_duplicates`): every addressed row becomes its OLD value plus the LAST update aimed at
        # it — not the sum of all of them (np.add.at).  gather, add, then the deterministic
        # last-writer-wins scatter-set.
        from pytensor_amd.dispatch.elemwise import launch_elemwise

        old = DeviceArray.empty(tgt, x.dtype)
        ffi.check(lib.pthip_take_rows(x.itemsize, n_idx, inner, out.ptr, x.shape[0], inner, iv.ptr, old.ptr))
        yfull = ysrc
        if ys0 != inner or tuple(ysrc.shape) != tuple(tgt):
            yfull = DeviceArray.empty(tgt, x.dtype)
            copy_into(yfull, yv.view(tgt, [0 if yv.shape[d] == 1 and tgt[d] != 1 else yv.strides[d] for d in range(len(tgt))]))
        dt = str(x.dtype)
        body = {"in_dtypes": [dt, dt], "out_dtypes": [dt], "body": [{"op": "Add", "in": [["i", 0], ["i", 1]], "dtype": dt}], "outs": [["t", 0]]}
        (ysrc,), _, _ = launch_elemwise(body, [old, yfull], tgt, [dt], None, env)
        ys0, inc = inner, 0
    ffi.check(
        lib.pthip_scatter_rows(
            ffi.np_dtype_code(x.dtype), inc, n_idx, inner, out.ptr, x.shape[0], iv.ptr, ysrc.ptr, ys0,
            ws.ptr if ws is not None else None, ws_bytes,
        )
    )
    return out
