"""``Elemwise`` / ``CAReduce`` / ``DimShuffle`` and the fused ``ElemwiseReduce``.

Reference: pytensor/tensor/elemwise.py — ``Elemwise`` 375 (perform 755-823,
``_check_runtime_broadcast`` 825-840), ``CAReduce`` 1233 (perform 1493-1511,
``_acc_dtype`` 1383-1417), ``DimShuffle`` 41 (view, 186-256).
"""

from __future__ import annotations

import ctypes as C
import hashlib
import json
import os
import struct

import numpy as np

from pytensor_amd import codegen, ffi, kernel_cache
from pytensor_amd.device import DeviceArray
from pytensor_amd.dispatch import handler
from pytensor_amd.executor import HOST_MAX, HostValue

BLOCK = codegen.BLOCK
import os as _os

MAX_GRID = int(_os.environ.get("PTHIP_EW_MAXGRID", 256 * 8))  # ≫256 workgroups, grid-stride beyond (cdna guide G11)
EW_UNROLL = int(_os.environ.get("PTHIP_EW_UNROLL", 2))
# software-pipelined main loop (codegen.flat_kernel_source prefetch=True) for scalar graphs with at
# least this many nodes.  OPT-IN: measured on config #2's 52-op fp64 body (profiles/r3c_c2_prefetch.txt)
# 33.3 / 34.8 us with it (unroll 1 / 2) against 32.6-34.7 without — inside the run-to-run spread; the
# kernel is co-limited by fp64 issue (PMC: 12.9 M VALU wave-instructions = 21 us per SIMD) and HBM
# (25 us), and eight resident waves per SIMD already overlap the two as well as a prefetch does
EW_PREFETCH_MIN_OPS = int(_os.environ.get("PTHIP_EW_PREFETCH_MIN_OPS", 1 << 30))

_body_key_cache = {}


def _body_key(body) -> str:
    k = id(body)
    v = _body_key_cache.get(k)
    if v is None:
        v = hashlib.sha256(json.dumps(body, sort_keys=True).encode()).hexdigest()[:16]
        _body_key_cache[k] = (v, body)
        return v
    return v[0]


# ---------------------------------------------------------------------------
# host evaluation of *shape arithmetic* (tiny integer/bool values only)
# ---------------------------------------------------------------------------

_HOST_OPS = {
    "Add": lambda *a: sum(a[1:], a[0]),
    "Mul": lambda *a: np.prod(np.broadcast_arrays(*a), axis=0) if len(a) > 2 else a[0] * a[1],
    "Sub": lambda a, b: a - b,
    "Neg": lambda a: -a,
    "IntDiv": lambda a, b: a // b,
    "Mod": lambda a, b: a % b,
    "Abs": abs,
    "EQ": lambda a, b: a == b,
    "NEQ": lambda a, b: a != b,
    "LT": lambda a, b: a < b,
    "GT": lambda a, b: a > b,
    "LE": lambda a, b: a <= b,
    "GE": lambda a, b: a >= b,
    "AND": lambda *a: np.bitwise_and.reduce(np.broadcast_arrays(*a)),
    "OR": lambda *a: np.bitwise_or.reduce(np.broadcast_arrays(*a)),
    "Invert": lambda a: ~a,
    "Maximum": np.maximum,
    "Minimum": np.minimum,
    "Switch": np.where,
    "Cast": lambda a: a,
    "Identity": lambda a: a,
    "Sign": np.sign,
    "Sqr": lambda a: a * a,
}


def _host_evaluable(body, inputs):
    if not all(isinstance(i, HostValue) for i in inputs):
        return False
    if any(np.dtype(d).kind not in "iub" for d in body["in_dtypes"] + body["out_dtypes"]):
        return False
    if any(np.dtype(n["dtype"]).kind not in "iub" for n in body["body"]):
        return False
    return all(n["op"] in _HOST_OPS for n in body["body"])


def _host_eval(body, inputs):
    vals = []

    def get(r):
        if r[0] == "i":
            return inputs[r[1]].a
        if r[0] == "t":
            return vals[r[1]]
        return np.asarray(r[1], dtype=r[2])

    for n in body["body"]:
        out = _HOST_OPS[n["op"]](*[get(r) for r in n["in"]])
        vals.append(np.asarray(out).astype(n["dtype"]))
    shape = np.broadcast_shapes(*[i.a.shape for i in inputs]) if inputs else ()
    return [
        HostValue(np.broadcast_to(np.asarray(get(r)).astype(dt), shape).copy())
        for r, dt in zip(body["outs"], body["out_dtypes"])
    ]


# ---------------------------------------------------------------------------
# launch planning
# ---------------------------------------------------------------------------


def _broadcast_shape(node, graph, ins, skip=()):
    """Elemwise output shape; ``skip`` = positions of split-K slab inputs (one extra leading
    dim, they do not take part in broadcasting)."""
    if skip:
        keep = [k for k in range(len(ins)) if k not in skip]
        sel = type("_Sel", (), {"inputs": [node.inputs[k] for k in keep]})
        shape = _broadcast_shape(sel, graph, [ins[k] for k in keep])
        # a slab operand (S, *product shape) is the unfinished product itself: it takes part in the
        # broadcast with its trailing shape (every other operand may be a broadcast row / scalar —
        # reference test tests/tensor/linalg/test_decomposition/test_qr.py::test_qr_grad)
        for k in skip:
            ps = tuple(ins[k].shape[1:])
            nd = max(len(ps), len(shape))
            a, b = (1,) * (nd - len(shape)) + tuple(shape), (1,) * (nd - len(ps)) + ps
            if any(x != y and x != 1 for x, y in zip(a, b)):
                raise ValueError(f"Incompatible Elemwise input shapes {[i.shape for i in ins]}")
            shape = b
        return shape
    nd = max((i.ndim for i in ins), default=0)
    shape = [1] * nd
    # operands are right-aligned (NumPy rule): the IR passes may leave an operand with fewer
    # dimensions than its siblings (a producer folded in through a DimShuffle view, inline.py);
    # the missing leading dimensions are broadcastable by construction
    shp = [(1,) * (nd - i.ndim) + tuple(i.shape) for i in ins]
    for d in range(nd):
        lens = {s[d] for s in shp}
        big = [l for l in lens if l != 1]
        if len(set(big)) > 1:
            raise ValueError(f"Incompatible Elemwise input shapes {[i.shape for i in ins]}")
        shape[d] = big[0] if big else (1 if lens else 1)
        if big and 1 in lens:
            # Elemwise._check_runtime_broadcast (elemwise.py:825-840)
            for vid, arr, s in zip(node.inputs, ins, shp):
                dd = d - (nd - arr.ndim)
                if dd < 0:
                    continue
                static = graph.vars[vid].shape
                if s[d] == 1 and static[dd] != 1:
                    raise ValueError(
                        f"Runtime broadcasting not allowed. One input had a distinct dimension length of 1 along axis {d}, "
                        "but the static type does not mark it as broadcastable"
                    )
    return tuple(shape)


def _collapse(shape, strides_list):
    """Merge adjacent dims that are jointly contiguous for every operand."""
    dims = [(s, [st[k] for st in strides_list]) for k, s in enumerate(shape) if s != 1]
    if not dims:
        return (1,), [[0] for _ in strides_list]
    out = [dims[0]]
    for s, sts in dims[1:]:
        ps, psts = out[-1]
        if all(p == s * c for p, c in zip(psts, sts)):
            out[-1] = (ps * s, sts)
        else:
            out.append((s, sts))
    shp = tuple(d[0] for d in out)
    per_op = [[d[1][k] for d in out] for k in range(len(strides_list))]
    return shp, per_op


def _grid(n_units):
    g = (n_units + BLOCK - 1) // BLOCK
    return max(1, min(g, MAX_GRID))


def _scalar_bits(h, dtype) -> int:
    """The bytes of a host scalar, zero-extended to the 8-byte argument slot."""
    raw = np.asarray(h.a, dtype=dtype).reshape(()).tobytes()
    return int.from_bytes(raw.ljust(8, b"\0"), "little", signed=True)


def _scalar_or_device(env, i):
    return i if isinstance(i, HostValue) and i.a.size == 1 else env.to_device(i)


def _placed(env, node, shape, out_dtypes):
    """Pre-assigned destinations (``env.placement``) for this node's outputs, where they fit."""
    if not env.placement:
        return None
    res = []
    for o, dt in zip(node.outputs, out_dtypes):
        b = env.placement.get(o)
        ok = b is not None and b.shape == tuple(shape) and str(b.dtype) == str(dt) and b.is_contiguous()
        res.append(b if ok else None)
    return res if any(r is not None for r in res) else None


def _take(env, table: DeviceArray, idx: DeviceArray) -> DeviceArray:
    """table[idx] for a 1-d table (the unfused form of a gather input)."""
    out = DeviceArray.empty(idx.shape, table.dtype)
    if out.size:
        tc = table.contiguous()
        ffi.check(env.lib.pthip_take_rows(table.itemsize, idx.size, 1, tc.ptr, tc.shape[0], 1, idx.contiguous().ptr, out.ptr))
    return out


# single-pass reductions: the kernel's last workgroup folds the partials itself (codegen._reduce_epilogue).
# Measured on BASELINE config #2 (160 MB streamed, 2048 workgroups; profiles/r4k_c2_ab.txt, r4i_c2_ab_fence_ticket.txt):
#   two launches (kernel + second stage)            32.6 - 33.4 us per replay  (kernel 30.6 - 31.0)
#   one pass, pairs + ticket, batched polls          40.0 - 40.8
#   one pass, pairs + ticket, polls one by one       44.5
#   one pass, __threadfence() + ticket               376          (an agent-scope fence per workgroup: L2 write-back +
#                                                                  invalidate under 2000 streaming workgroups)
# The second launch costs less than any in-kernel hand-over here: opt-in only (PTHIP_EW_SINGLE_PASS=1).
_SINGLE_PASS = os.environ.get("PTHIP_EW_SINGLE_PASS", "0") == "1"


def launch_elemwise(body, ins, out_shape, out_dtypes, reduce_spec, env, partial=(), out_bufs=None, gather=None, finals_out=None):
    """Launch the fused kernel.  Returns (stored outputs or None per output,
    partial buffers or None per output, grid).

    ``partial``: positions of inputs that are unfinished split-K slabs ``(S, *out_shape)``
    (``GemmPartials``): summed inside the kernel, in slab order."""
    lib = env.lib
    n = int(np.prod(out_shape)) if out_shape else 1
    nout = len(out_dtypes)
    reduce_spec = reduce_spec or [None] * nout
    outs = [
        None if reduce_spec[k] else (out_bufs[k] if out_bufs and out_bufs[k] is not None else DeviceArray.empty(out_shape, out_dtypes[k]))
        for k in range(nout)
    ]
    if n == 0:
        return outs, [None] * nout, 0
    nd = len(out_shape)
    partial = set(partial)
    gather = gather or {}
    # right-align operands that arrive with fewer dimensions than the output (see _broadcast_shape)
    ins = [a.view((1,) * (nd - a.ndim) + tuple(a.shape), (0,) * (nd - a.ndim) + tuple(a.strides))
           if (isinstance(a, DeviceArray) and k not in partial and k not in gather and a.ndim < nd) else a for k, a in enumerate(ins)]
    modes = []
    flat = not partial
    for k, a in enumerate(ins) if flat else ():
        if k in gather:
            modes.append("G")  # table[idx[i]] read inside the loop (gatherfuse.py)
        elif isinstance(a, HostValue):
            modes.append("C")  # host-known scalar: by value, no upload node in a captured plan
        elif a.size == 1:
            modes.append("S")
        elif a.shape == tuple(out_shape) and a.is_contiguous():
            modes.append("V")
        else:
            flat = False
            break
    if gather:
        for k, idx in gather.items():
            t = ins[k]
            if not (flat and isinstance(t, DeviceArray) and t.ndim == 1 and t.is_contiguous() and t.shape[0] > 0
                    and idx.shape == tuple(out_shape) and idx.is_contiguous() and str(idx.dtype) == "int64"):
                flat = False
        if not flat:
            # layouts the in-loop gather does not cover: gather first, then the ordinary kernel
            ins = [_take(env, env.to_device(a), gather[k]) if k in gather else a for k, a in enumerate(ins)]
            return launch_elemwise(body, ins, out_shape, out_dtypes, reduce_spec, env, partial, out_bufs, finals_out=finals_out)
    byvalue = set()
    if not flat:
        # host-known scalars travel by value here too (no upload node per replay)
        byvalue = {k for k, a in enumerate(ins) if isinstance(a, HostValue) and a.a.size == 1}
        ins = [a if k in byvalue else env.to_device(a) for k, a in enumerate(ins)]
    bkey = _body_key(body)
    rkey = "".join("-" if r is None else r["op"][0] + r["acc_dtype"][0] + r["acc_dtype"][-1] for r in reduce_spec)
    rs = [None if r is None else (r["op"], r["acc_dtype"]) for r in reduce_spec]
    if flat:
        dts = [body["in_dtypes"][k] for k, m in enumerate(modes) if m == "V"] + [
            d for k, d in enumerate(out_dtypes) if reduce_spec[k] is None
        ]
        vec = codegen._vec_width(dts) if dts else 1
        if vec > 1:
            ptrs = [a.ptr for a, m in zip(ins, modes) if m == "V"] + [o.ptr for o in outs if o is not None]
            if any(p % 16 for p in ptrs) or n < vec or "G" in modes:
                vec = 1
        unroll = EW_UNROLL
        prefetch = vec > 1 and "V" in modes and len(body["body"]) >= EW_PREFETCH_MIN_OPS
        units = (n // vec + unroll - 1) // unroll if vec > 1 else n
        grid = _grid(max(units, 1))
        # the caller wants finished values (nothing defers the second stage to a Tail kernel) and every output is
        # a reduction: the kernel finishes them itself
        finish = None
        if finals_out is not None and _SINGLE_PASS and 1 < grid <= codegen.FINISH_MAX_PER_THREAD * BLOCK and all(r is not None for r in reduce_spec) and isinstance(reduce_spec[0], dict):
            finish = [r["dtype"] for r in reduce_spec]
        name = f"ew_{bkey}_{''.join(modes)}_v{vec}_{rkey}".replace("-", "x") + (f"_pf{unroll}" if prefetch else "") + ("_1p" + "".join(np.dtype(d).char for d in finish) if finish else "")
        src = codegen.flat_kernel_source(name, body, "".join(modes), vec, rs, unroll, prefetch=prefetch, finish=finish)
        fn = kernel_cache.get_function(src, name)
        args = [n]
        for k, (a, m) in enumerate(zip(ins, modes)):
            if m == "C":
                args.append(_scalar_bits(a, body["in_dtypes"][k]))
            elif m == "G":
                args += [a.ptr, gather[k].ptr, a.shape[0]]
            else:
                args.append(a.ptr)
    else:
        sshape = tuple(out_shape)
        strides = []
        for k, a in enumerate(ins):
            if k in byvalue:
                continue
            if k in partial:
                # slabs (S, *shape): rows of shape[-1] contiguous elements, uniformly spaced
                # (a column block of wider slabs is fine: merge_sibling_gemms)
                ok = a.shape[1:] == sshape and (sshape[-1] == 1 or a.strides[-1] == 1)
                for d in range(1, nd - 1):
                    ok = ok and (a.shape[d] == 1 or a.strides[d] == a.strides[d + 1] * a.shape[d + 1])
                if not ok:
                    raise ValueError(f"split-K slabs of shape {a.shape}/strides {a.strides} do not match the Elemwise shape {sshape}")
                continue
            st = tuple(0 if a.shape[d] == 1 and sshape[d] != 1 else a.strides[d] for d in range(nd))
            strides.append(st)
        cshape, cstr = _collapse(sshape, strides + [_cstrides(sshape)])
        cstr = cstr[:-1]
        ndc = len(cshape)
        if ndc > codegen.MAX_ND:
            # rare: more than MAX_ND dimensions that do not merge (seven alternating broadcast / full dimensions: golden
            # logsumexp_degenerate).  Every operand that is not already the full shape in C order is expanded into one
            # (copy_into broadcasts and peels dimensions beyond its own limit); the retry is then the flat kernel.
            # (Until round 5 this made operands contiguous WITHOUT expanding them: a broadcast operand came back with the
            #  same collapsed rank and the call recursed until the interpreter gave up.)
            from pytensor_amd.device import copy_into

            if partial:
                raise NotImplementedError(f"Elemwise over split-K slabs with {ndc} non-mergeable dimensions")
            full, changed = [], False
            for k, a in enumerate(ins):
                if k in byvalue or isinstance(a, HostValue) or a.size == 1 or (tuple(a.shape) == sshape and a.is_contiguous()):
                    full.append(a)
                    continue
                t = DeviceArray.empty(sshape, a.dtype)
                copy_into(t, a)
                full.append(t)
                changed = True
            if not changed:
                raise RuntimeError(f"Elemwise: {ndc} non-mergeable dimensions with every operand already expanded")
            return launch_elemwise(body, full, out_shape, out_dtypes, reduce_spec, env, partial, out_bufs, finals_out=finals_out)
        if not partial and _TILE:
            return _launch_tiled(body, ins, byvalue, cshape, cstr, outs, out_dtypes, reduce_spec, rs, bkey, rkey, env)
        pkey = ("_p" + "".join(str(k) + "." for k in sorted(partial))) if partial else ""
        pkey += ("_c" + "".join(str(k) + "." for k in sorted(byvalue))) if byvalue else ""
        name = f"ewnd_{bkey}_d{ndc}_{rkey}{pkey}".replace("-", "x").replace(".", "_")
        src = codegen.nd_kernel_source(name, body, ndc, rs, partial, byvalue)
        fn = kernel_cache.get_function(src, name)
        grid = _grid(n)
        args = [n] + list(cshape)
        it = iter(cstr)
        for k, a in enumerate(ins):
            if k in byvalue:
                args.append(_scalar_bits(a, body["in_dtypes"][k]))
                continue
            args.append(a.ptr)
            if k in partial:
                pn = sshape[-1] if nd else 1
                pld = a.strides[-2] if nd >= 2 and n > pn else pn
                args += [a.shape[0], a.strides[0] if a.shape[0] > 1 else 0, pn, pld]
            else:
                args += list(next(it))
    if flat and finish:
        # pair arrays (16 bytes per workgroup and reduced output) instead of partials; never cleared from here: the
        # kernel's last workgroup zeroes every pair it consumed, other users of pool memory leave pairs under other
        # magic constants, and arbitrary bits validate with probability 2^-64
        parts = [DeviceArray.empty((2 * grid,), "uint64") for _ in range(nout)]
    else:
        parts = alloc_partials(reduce_spec, grid)
    for k in range(nout):
        args.append(outs[k].ptr if reduce_spec[k] is None else parts[k].ptr)
    if flat and "G" in modes:
        args.append(env.lib.pthip_status_ptr())  # device error flag: out-of-range index
    if flat and finish:
        fins = [DeviceArray.empty((), dt) for dt in finish]
        slot = C.c_void_p()
        ffi.check(lib.pthip_ticket_slot(C.byref(slot)))
        args += [f.ptr for f in fins] + [slot.value, env.lib.pthip_status_ptr()]
        finals_out[:] = fins
        env.keepalive.append(parts)
    buf = struct.pack(f"<{len(args)}q", *args)
    env.timed(name, lambda: ffi.check(lib.pthip_launch(fn, grid, 1, 1, BLOCK, 1, 1, 0, buf, len(buf))))
    return outs, parts, grid


# the tiled N-d loop (codegen_tile.py); PTHIP_EW_TILE=0 restores the element-per-thread loop with a division per dimension
_TILE = os.environ.get("PTHIP_EW_TILE", "1") != "0"
_TILE_RPT = int(os.environ.get("PTHIP_EW_TILE_RPT", 4))
_LDS_BUDGET = 64 * 1024


def _pow2ceil(x):
    p = 1
    while p < x:
        p *= 2
    return p


def tile_plan(cshape, cstr, dev_dtypes, out_dtypes_stored, ptrs, out_ptrs):
    """Host-side planning of one tiled launch (codegen_tile.tile_kernel_source).

    ``cshape`` / ``cstr``: the collapsed iteration space and, per device operand, its element strides over it
    (0 on broadcast dimensions).  Returns ``dict(jr, batch, cls, V, TX, RPT, lds_rows)``: the tile's row
    dimension, the batch dimensions, the operand classes, the pack width and the tile shape."""
    nd = len(cshape)
    D = cshape[-1]
    # ---- the tile's row dimension: where a transposed operand is contiguous, else the last outer dimension ----
    jr = nd - 2 if nd > 1 else None
    tdim = None
    if nd > 1:
        weight = {}
        for st, dt in zip(cstr, dev_dtypes):
            if st[-1] in (0, 1):
                continue
            for j in range(nd - 1):
                if st[j] == 1 and cshape[j] >= 16:
                    weight[j] = weight.get(j, 0) + np.dtype(dt).itemsize
        if weight:
            tdim = max(weight, key=lambda j: (weight[j], j))
            jr = tdim
    R = cshape[jr] if jr is not None else 1
    batch = [j for j in range(nd - 1) if j != jr]
    # ---- operand classes ----
    cls = []
    for st in cstr:
        si = st[-1]
        sr = st[jr] if jr is not None else 0
        if not any(st):
            cls.append("S")
        elif si == 1:
            cls.append("R" if (sr == 0 and R > 1) else "V")
        elif si == 0:
            cls.append("B")
        elif tdim is not None and sr == 1:
            cls.append("T")
        else:
            cls.append("G")
    # ---- pack width: 16 bytes of the widest streamed type, as far as every pack stays aligned ----
    wide = [np.dtype(dt).itemsize for dt, c in zip(dev_dtypes, cls) if c in "VR"] + [np.dtype(dt).itemsize for dt in out_dtypes_stored]
    V = max(1, min(4, 16 // max(wide))) if wide else 1

    def aligned(v):
        if D % v:
            return False
        for st, dt, c, p in zip(cstr, dev_dtypes, cls, ptrs):
            if c in "VR" and (p % (v * np.dtype(dt).itemsize) or any(s_ % v for s_ in st[:-1])):
                return False
        return all(p % (v * np.dtype(dt).itemsize) == 0 for dt, p in zip(out_dtypes_stored, out_ptrs))

    while V > 1 and not aligned(V):
        V //= 2
    # ---- tile shape ----
    lds_rows = 0
    if "T" in cls:
        TX = 64 // V
        TY = BLOCK // TX
        tb = lambda tr: sum(64 * (tr + 1) * np.dtype(dt).itemsize for dt, c in zip(dev_dtypes, cls) if c == "T")
        lds_rows = 64 if tb(64) <= _LDS_BUDGET else 32
        while tb(lds_rows) > _LDS_BUDGET:
            # more transposed operands than LDS holds: the last ones read at their strides
            k = max(i for i, c in enumerate(cls) if c == "T")
            cls[k] = "G"
        if "T" not in cls:
            lds_rows = 0
        RPT = max(1, (lds_rows or 64) // TY)
    if "T" not in cls:
        TX = min(BLOCK, _pow2ceil(-(-D // V)))
        TY = BLOCK // TX
        RPT = max(1, min(max(_TILE_RPT, 8 // V), -(-R // TY)))
    # ---- 16-byte loads along the contiguous axis of the transposed operands (the tile's row dimension) ----
    tvec = 1
    if "T" in cls and lds_rows and os.environ.get("PTHIP_EW_TVEC", "1") != "0":
        sizes = {np.dtype(dt).itemsize for dt, c in zip(dev_dtypes, cls) if c == "T"}
        if len(sizes) == 1 and next(iter(sizes)) in (4, 8):
            tv = 16 // next(iter(sizes))
            ok = R % tv == 0 and lds_rows % tv == 0
            for st, c, p_ in zip(cstr, cls, ptrs):
                if c == "T":
                    ok = ok and p_ % 16 == 0 and all(s_ % tv == 0 for j, s_ in enumerate(st) if j != jr)
            if ok:
                tvec = tv
    return {"jr": jr, "batch": batch, "cls": "".join(cls), "V": V, "TX": TX, "RPT": RPT, "lds_rows": lds_rows, "R": R, "D": D, "tvec": tvec}


def _launch_tiled(body, ins, byvalue, cshape, cstr, outs, out_dtypes, reduce_spec, rs, bkey, rkey, env):
    from pytensor_amd import codegen_tile

    lib = env.lib
    nout = len(out_dtypes)
    dev = [k for k in range(len(ins)) if k not in byvalue]
    stored = [k for k in range(nout) if reduce_spec[k] is None]
    plan = tile_plan(cshape, cstr, [body["in_dtypes"][k] for k in dev], [out_dtypes[k] for k in stored], [ins[k].ptr for k in dev], [outs[k].ptr for k in stored])
    jr, batch, V, TX, RPT = plan["jr"], plan["batch"], plan["V"], plan["TX"], plan["RPT"]
    nb = len(batch)
    cls, it = [], iter(plan["cls"])
    for k in range(len(ins)):
        cls.append("C" if k in byvalue else next(it))
    cls = "".join(cls)
    TY = BLOCK // TX
    TC, TR = TX * V, TY * RPT
    R, D = plan["R"], plan["D"]
    nrb, ncb = -(-R // TR), -(-D // TC)
    grid = nrb * ncb * int(np.prod([cshape[j] for j in batch])) if batch else nrb * ncb
    tvec = plan.get("tvec", 1)
    name = f"ewt_{bkey}_{cls}_b{nb}_v{V}_x{TX}_r{RPT}_{rkey}".replace("-", "x") + (f"_tv{tvec}" if tvec > 1 else "")
    src = codegen_tile.tile_kernel_source(name, body, cls, nb, V, TX, RPT, rs, plan["lds_rows"], tvec)
    fn = kernel_cache.get_function(src, name)
    ocs = _cstrides(cshape)
    args = [R, D, nrb, ncb] + [cshape[j] for j in batch] + [ocs[jr] if jr is not None else 0] + [ocs[j] for j in batch]
    its = iter(cstr)
    for k, a in enumerate(ins):
        if k in byvalue:
            args.append(_scalar_bits(a, body["in_dtypes"][k]))
            continue
        st = next(its)
        args += [a.ptr] + [st[j] for j in batch] + [st[jr] if jr is not None else 0, st[-1]]
    parts = alloc_partials(reduce_spec, grid)
    for k in range(nout):
        args.append(outs[k].ptr if reduce_spec[k] is None else parts[k].ptr)
    buf = struct.pack(f"<{len(args)}q", *args)
    env.timed(name, lambda: ffi.check(lib.pthip_launch(fn, grid, 1, 1, BLOCK, 1, 1, 0, buf, len(buf))))
    return outs, parts, grid


# ---------------------------------------------------------------------------
# N-d reductions over an axis tuple (codegen_tile.tile_reduce_source)
# ---------------------------------------------------------------------------

_ND_REDUCE = os.environ.get("PTHIP_ND_REDUCE", "1") != "0"
# splits of an N-d reduction folded by the kernel's last workgroup per output tile instead of a second launch: OPT-IN.
# Measured on the 21 CAReduce cases of tools/bench_hotpath.py (256^3 fp64, cold): nothing gained — 22-26 us + a 4.5 us second
# stage became 25-35 us (the closing fold is a serial poll -> fold -> store at the very end of the launch, and every
# workgroup pays a write-through pair store and an atomic), and 136 us where a tile's elements x splits exceed a few
# thousand pairs (sum over axes (0, 1): 128 columns x 512 splits for ONE workgroup).  profiles/r8_nd_onepass.txt.
_ND_ONEPASS = os.environ.get("PTHIP_ND_ONEPASS", "0") == "1"
_ND_REDUCE_WGS = int(os.environ.get("PTHIP_ND_REDUCE_WGS", 1024))
_RED_RPT = int(os.environ.get("PTHIP_RED_RPT", 0))  # rows per thread of the reduction tiles (0: as the elementwise tiles)
_LSE_INFLIGHT = int(os.environ.get("PTHIP_LSE_INFLIGHT", 8))  # (8, 16, 32 measured the same: profiles/r5t_lse_inflight.txt)  # workgroups wanted before the reduced range is split


def reduce_plan(shape, axes, strides, dev_dtypes, ptrs, out_shape):
    """Host-side planning of one N-d reduction launch.

    ``shape``: the iteration space; ``axes``: its reduced dimensions; ``strides``: per device operand, element
    strides over ``shape`` (0 on broadcast dimensions); the output is C-contiguous over the kept dimensions.
    Dimensions of one role (kept / reduced) are sorted by the dominant operand's stride and merged where every
    operand (and, for kept ones, the output) is jointly contiguous — a sum over a transposed view collapses
    back to the layout in memory.  Returns ``None`` when the shape does not fit the tile (more than 3 batch
    dimensions of one role, > 2^31 tile visits)."""
    nd = len(shape)
    axes = set(axes)
    ost, accu = [0] * nd, 1
    for d in reversed(range(nd)):
        if d not in axes:
            ost[d] = accu
            accu *= shape[d]
    n_out = accu
    weights = [sum(1 for d in range(nd) if st[d] != 0 and shape[d] > 1) * 1000 + np.dtype(dt).itemsize for st, dt in zip(strides, dev_dtypes)]
    dom = int(np.argmax(weights)) if weights else 0
    dims = [{"n": shape[d], "red": d in axes, "st": [st[d] for st in strides], "ost": ost[d]} for d in range(nd) if shape[d] != 1]
    merged = []
    for role in (False, True):
        group = sorted((x for x in dims if x["red"] == role), key=lambda x: -abs(x["st"][dom]))
        out = []
        for x in group:
            if out:
                p = out[-1]
                if all(a == b * x["n"] for a, b in zip(p["st"], x["st"])) and (role or p["ost"] == x["ost"] * x["n"]):
                    p["n"] *= x["n"]
                    p["st"], p["ost"] = x["st"], x["ost"]
                    continue
            out.append(dict(x))
        merged += out
    if not any(x["red"] for x in merged):
        return None
    # the inner dimension: where the dominant operand is contiguous
    unit = [x for x in merged if x["st"][dom] == 1] or [x for x in merged if x["st"][dom] == -1]
    inner = unit[0] if unit else min(merged, key=lambda x: (abs(x["st"][dom]) == 0, abs(x["st"][dom])))
    rest = [x for x in merged if x is not inner]
    if inner["red"]:
        kept = [x for x in rest if not x["red"]]
        cand = kept or [x for x in rest if x["red"]]
        row = min(cand, key=lambda x: (x["st"][dom] == 0, abs(x["st"][dom]))) if kept else (max(cand, key=lambda x: x["n"]) if cand else None)
    else:
        row = max((x for x in rest if x["red"]), key=lambda x: x["n"])
    rest = [x for x in rest if x is not row]
    kb = [x for x in rest if not x["red"]]
    rd = [x for x in rest if x["red"]]
    if len(kb) > 3 or len(rd) > 3:
        return None
    R = row["n"] if row is not None else 1
    D = inner["n"]
    row_kept = row is not None and not row["red"]
    inner_kept = not inner["red"]
    cls = []
    for k in range(len(strides)):
        si = inner["st"][k]
        sr = row["st"][k] if row is not None else 0
        if not any(x["st"][k] for x in merged):
            cls.append("S")
        elif si == 1:
            cls.append("R" if (sr == 0 and R > 1) else "V")
        elif si == 0:
            cls.append("B")
        else:
            cls.append("G")
    wide = [np.dtype(dt).itemsize for dt, c in zip(dev_dtypes, cls) if c in "VR"]
    V = max(1, min(4, 16 // max(wide))) if wide else 1

    def aligned(v):
        if D % v:
            return False
        for k, (dt, c, p) in enumerate(zip(dev_dtypes, cls, ptrs)):
            if c in "VR" and (p % (v * np.dtype(dt).itemsize) or any(x["st"][k] % v for x in merged if x is not inner)):
                return False
        return True

    while V > 1 and not aligned(V):
        V //= 2
    cols = -(-D // V)
    if inner_kept:
        # (the thread rows split the reduced rows inside the workgroup: >= 4 of them, combined through LDS)
        TX = min(64, _pow2ceil(cols))
    else:
        TX = BLOCK if cols >= BLOCK else min(64, _pow2ceil(cols))
    TY = BLOCK // TX
    RPT = max(1, min(_RED_RPT or max(_TILE_RPT, 8 // V), -(-R // TY)))
    TC, TR = TX * V, TY * RPT
    nrb, ncb = -(-R // TR), -(-D // TC)
    n_nat = int(np.prod([x["n"] for x in kb], dtype=np.int64)) * (nrb if row_kept else 1) * (ncb if inner_kept else 1)
    iters = int(np.prod([x["n"] for x in rd], dtype=np.int64)) * (1 if row_kept else nrb) * (1 if inner_kept else ncb)
    if iters >= 2**31 or n_nat >= 2**30:
        return None
    return {"row": row, "inner": inner, "kb": kb, "rd": rd, "R": R, "D": D, "row_kept": row_kept, "inner_kept": inner_kept, "cls": "".join(cls), "V": V, "TX": TX,
            "RPT": RPT, "nrb": nrb, "ncb": ncb, "n_nat": n_nat, "iters": iters, "n_out": n_out}


def launch_axis_reduce(env, body, ins, shape, axes, specs, out_shape):
    """``out[k] = reduce_{axes} body_k(ins)`` in one pass over the operands (+ a small second launch when the reduced
    range had to be split for parallelism).  ``specs[k] = (op, acc_dtype, out_dtype)``.  Returns the outputs, or
    ``None`` when the shape is outside what the tile covers (the caller then takes the run-by-run path)."""
    from pytensor_amd import codegen_tile

    byvalue = {k for k, a in enumerate(ins) if isinstance(a, HostValue)}
    dev = [k for k in range(len(ins)) if k not in byvalue]
    nd = len(shape)
    strides = []
    for k in dev:
        a = ins[k]
        shp = (1,) * (nd - a.ndim) + tuple(a.shape)
        st = (0,) * (nd - a.ndim) + tuple(a.strides)
        strides.append(tuple(0 if shp[d] == 1 and shape[d] != 1 else st[d] for d in range(nd)))
    plan = reduce_plan(shape, axes, strides, [body["in_dtypes"][k] for k in dev], [ins[k].ptr for k in dev], out_shape)
    if plan is None:
        return None
    kb, rd, row, inner = plan["kb"], plan["rd"], plan["row"], plan["inner"]
    n_out, iters, n_nat = plan["n_out"], plan["iters"], plan["n_nat"]
    accsz = max(np.dtype(sp[1]).itemsize for sp in specs)
    in_bytes = int(np.prod(shape, dtype=np.int64)) * max(np.dtype(body["in_dtypes"][k]).itemsize for k in dev)
    nsplit = max(1, min(iters, -(-_ND_REDUCE_WGS // n_nat)))
    nsplit = max(1, min(nsplit, in_bytes // max(1, 16 * n_out * accsz * len(specs))))
    # partial layout: few outputs -> [n_out][split] (the second launch reads each output's splits contiguously, 64 lanes
    # per output; at most 512 splits so that it is ONE launch); many outputs -> [split][n_out] (a thread per output)
    split_fastest = n_out <= 8192
    if split_fastest:
        nsplit = min(nsplit, 512)
    chunk = -(-iters // nsplit)
    nsplit = -(-iters // chunk)
    final = nsplit == 1
    ps_split, ps_out = (0, 1) if final else ((1, nsplit) if split_fastest else (n_out, 1))
    cls, it = [], iter(plan["cls"])
    for k in range(len(ins)):
        cls.append("C" if k in byvalue else next(it))
    cls = "".join(cls)
    V, TX, RPT = plan["V"], plan["TX"], plan["RPT"]
    kouts = [(op, acc, (odt if final else acc)) for op, acc, odt in specs]
    # ONE pass (round 6): with several splits and no log-sum-exp state the kernel's last workgroup per output tile folds
    # the splits itself (codegen_tile.tile_reduce_source `finish`) — no second launch
    nel = (TX * V) if (plan["inner_kept"] and not plan["row_kept"]) else (((BLOCK // TX) * RPT) if plan["row_kept"] else 1)
    onepass = (not final and _ND_ONEPASS and nsplit <= 512 and n_nat <= 8192 and nel * nsplit <= 4096 and not any(op == "LogSumExp" for op, _, _ in specs)
               and all(np.dtype(acc).itemsize <= 8 for _, acc, _ in specs))
    finish = [odt for _, _, odt in specs] if onepass else None
    okey = "_".join(f"{op[:2]}{_dtag(acc)}{_dtag(sd)}" for op, acc, sd in kouts) + ("_1p" + "".join(_dtag(d) for d in finish) if finish else "")
    # tile visits whose loads are in flight together: 8 / RPT
    ui = max(1, min((_LSE_INFLIGHT if any(op == "LogSumExp" for op, _, _ in specs) else 8) // RPT, chunk))
    name = f"rnd_{_body_key(body)}_{cls}_k{len(kb)}r{len(rd)}_{'K' if plan['row_kept'] else 'R'}{'K' if plan['inner_kept'] else 'R'}_v{V}_x{TX}_r{RPT}_u{ui}_{okey}"
    src = codegen_tile.tile_reduce_source(name, body, cls, len(kb), len(rd), plan["row_kept"], plan["inner_kept"], V, TX, RPT, kouts, ui, finish=finish)
    fn = kernel_cache.get_function(src, name)
    args = [plan["R"], plan["D"], plan["nrb"], plan["ncb"], iters, chunk, ps_split] + [x["n"] for x in kb] + [x["n"] for x in rd]
    args += [(row["ost"] if plan["row_kept"] else 0) * ps_out, (inner["ost"] if plan["inner_kept"] else 0) * ps_out] + [x["ost"] * ps_out for x in kb]
    j = 0
    for k, a in enumerate(ins):
        if k in byvalue:
            args.append(_scalar_bits(a, body["in_dtypes"][k]))
            continue
        args += [a.ptr] + [x["st"][j] for x in kb] + [x["st"][j] for x in rd] + [row["st"][j] if row is not None else 0, inner["st"][j]]
        j += 1
    outs = [DeviceArray.empty(out_shape, odt) for _, _, odt in specs]
    if onepass:
        # self-validating pairs (16 bytes per partial) instead of plain partials; never cleared from here (the last
        # workgroup zeroes what it consumed; arbitrary bits validate with probability 2^-64: see launch_elemwise)
        dsts = [DeviceArray.empty((2 * nsplit * n_out,), "uint64") for _ in specs]
    else:
        dsts = outs if final else [DeviceArray.empty((nsplit * n_out,), acc) for _, acc, _ in specs]
    args += [d.ptr for d in dsts]
    if onepass:
        tk = C.c_void_p()
        ffi.check(env.lib.pthip_ticket_slots(int(n_nat), C.byref(tk)))
        pt_L = 1
        while pt_L < min(64, nsplit):
            pt_L *= 2
        args += [o.ptr for o in outs]
        args += [(row["ost"] if plan["row_kept"] else 0), (inner["ost"] if plan["inner_kept"] else 0)] + [x["ost"] for x in kb]
        args += [tk.value, nsplit, n_nat, pt_L, env.lib.pthip_status_ptr()]
        env.keepalive.append(dsts)
    grid = n_nat * nsplit
    buf = struct.pack(f"<{len(args)}q", *args)
    env.timed(name, lambda: ffi.check(env.lib.pthip_launch(fn, grid, 1, 1, BLOCK, 1, 1, 0, buf, len(buf))))
    if onepass:
        return outs
    if not final and any(op == "LogSumExp" for op, _, _ in specs):
        # every split stored its own log-sum-exp: the splits fold by the same reduction (a second, small launch)
        res = []
        for part, (op, acc, odt) in zip(dsts, specs):
            pv = part.view((n_out, nsplit), (nsplit, 1)) if split_fastest else part.view((nsplit, n_out), (n_out, 1))
            r2 = launch_axis_reduce(env, _identity_body(acc), [pv], pv.shape, [1 if split_fastest else 0], [(op, acc, odt)], (n_out,))
            if r2 is None:
                raise RuntimeError("log-sum-exp: the second stage does not fit the tile")
            res.append(r2[0].view(out_shape, _cstrides(out_shape)))
        return res
    if not final:
        if split_fastest:
            outs = [device_reduce(env, op, part, n_out, nsplit, 1, nsplit, 1, 0, acc, odt, out_shape) for part, (op, acc, odt) in zip(dsts, specs)]
        else:
            outs = [device_reduce(env, op, part, 1, nsplit, n_out, 0, n_out, 1, acc, odt, out_shape) for part, (op, acc, odt) in zip(dsts, specs)]
    return outs


class _StreamEnv:
    """what a launch needs of an ``Env`` when there is no graph around it (device.copy_into)"""

    kernel_timer = None
    placement = None

    def __init__(self):
        self.lib = ffi.lib()
        self.keepalive = []

    def timed(self, name, launch):
        return launch()


def tiled_copy(dst: DeviceArray, src: DeviceArray) -> bool:
    """``dst[...] = src`` (same shape, ``dst`` contiguous, ``src`` any strides, 0 on broadcast dimensions) through
    the tiled N-d loop.  False when the dtype or the shape is outside what it covers."""
    dt = str(dst.dtype)
    if dt not in codegen.CTYPE or not _TILE:
        return False
    shape = tuple(dst.shape)
    cshape, cstr = _collapse(shape, [tuple(src.strides), _cstrides(shape)])
    if len(cshape) > codegen.MAX_ND:
        return False
    body = _identity_body(dt)
    _launch_tiled(body, [src], set(), cshape, cstr[:-1], [dst], [dt], [None], [None], _body_key(body), "x", _StreamEnv())
    return True


def _dtag(dt):
    d = np.dtype(dt)
    return f"{d.kind}{d.itemsize}"


def _identity_body(dtype):
    key = str(dtype)
    b = _IDENT_BODIES.get(key)
    if b is None:
        b = _IDENT_BODIES[key] = {"in_dtypes": [key], "out_dtypes": [key], "body": [{"op": "Identity", "in": [["i", 0]], "dtype": key}], "outs": [["t", 0]]}
    return b


_IDENT_BODIES = {}


def _homogeneous(spec):
    red = [r for r in spec if r is not None]
    return len(red) > 1 and all((r["op"], r["acc_dtype"], r["dtype"]) == (red[0]["op"], red[0]["acc_dtype"], red[0]["dtype"]) for r in red)


def alloc_partials(spec, grid):
    """One [n_reduced][grid] slab when every reduced output shares (op, acc, dtype), so that
    ONE second-stage launch finishes them all; separate buffers otherwise."""
    parts = [None] * len(spec)
    red = [k for k, r in enumerate(spec) if r is not None]
    if not red:
        return parts
    if _homogeneous(spec):
        slab = DeviceArray.empty((len(red), grid), spec[red[0]]["acc_dtype"])
        for j, k in enumerate(red):
            parts[k] = slab.view((grid,), (1,), j * grid)
    else:
        for k in red:
            parts[k] = DeviceArray.empty((grid,), spec[k]["acc_dtype"])
    return parts


def finish_partials(env, spec, parts, grid, defer=()):
    """Second stage: per-workgroup partials -> final 0-d values (fixed order, deterministic).
    ``defer``: output positions whose only consumer is a ``Tail`` node — left as
    :class:`~pytensor_amd.executor.DeferredReduce` (no launch here)."""
    res = [None] * len(spec)
    red = [k for k, r in enumerate(spec) if r is not None]
    if not red:
        return res
    if defer and grid > 1:
        from pytensor_amd.executor import DeferredReduce

        rest = [None if (k in defer or r is None) else r for k, r in enumerate(spec)]
        for k in red:
            if k in defer:
                res[k] = DeferredReduce(parts[k], grid, spec[k])
        if any(r is not None for r in rest):
            # (the non-deferred ones keep their own second stage; a shared slab is then read per output)
            for k, r in enumerate(rest):
                if r is not None:
                    res[k] = device_reduce(env, r["op"], parts[k], 1, grid, 1, 0, 1, 0, r["acc_dtype"], r["dtype"], ())
        return res
    if grid == 1:
        # a single workgroup already produced the final value: no second-stage launch
        for k in red:
            r = spec[k]
            if r["acc_dtype"] == r["dtype"]:
                res[k] = parts[k].view((), (), 0)
            else:
                res[k] = device_reduce(env, r["op"], parts[k], 1, 1, 1, 0, 1, 0, r["acc_dtype"], r["dtype"], ())
        return res
    if _homogeneous(spec):
        r = spec[red[0]]
        slab = parts[red[0]]  # first view starts at the slab base
        out = device_reduce(env, r["op"], slab, len(red), grid, 1, grid, 1, 0, r["acc_dtype"], r["dtype"], (len(red),))
        for j, k in enumerate(red):
            res[k] = out.view((), (), j)
    else:
        for k in red:
            r = spec[k]
            res[k] = device_reduce(env, r["op"], parts[k], 1, grid, 1, 0, 1, 0, r["acc_dtype"], r["dtype"], ())
    return res


def _cstrides(shape):
    st = []
    acc = 1
    for s in reversed(shape):
        st.append(acc)
        acc *= max(int(s), 1)
    return tuple(reversed(st))


def device_reduce(env, op, x: DeviceArray, A, R, B, sA, sR, sB, acc_dtype, out_dtype, out_shape):
    lib = env.lib
    acc_code = ffi.np_dtype_code(acc_dtype)
    out = DeviceArray.empty(out_shape, out_dtype)
    ws_bytes = lib.pthip_reduce_workspace(acc_code, A, R, B)
    ws = DeviceArray.empty((ws_bytes,), "uint8") if ws_bytes else None
    ffi.check(
        lib.pthip_reduce(
            ffi.REDUCE_CODE[op],
            ffi.np_dtype_code(x.dtype),
            acc_code,
            ffi.np_dtype_code(out_dtype),
            x.ptr,
            out.ptr,
            A,
            R,
            B,
            sA,
            sR,
            sB,
            ws.ptr if ws is not None else None,
            ws_bytes,
        )
    )
    return out


# ---------------------------------------------------------------------------
# handlers
# ---------------------------------------------------------------------------


def _prepare(node, inputs, env):
    """Handler inputs → (kernel inputs, iteration shape, split-K slab positions, gather map).

    ``params["gather"] = [[pos, extra], …]`` (gatherfuse.py): body input ``pos`` is a 1-d table
    read as ``table[idx]`` with ``idx = inputs[n_body + extra]``; for the shape rules the
    gathered operand has the index's shape."""
    body = node.params["scalar"]
    nbody = len(body["in_dtypes"])
    ins = [_scalar_or_device(env, i) for i in inputs[:nbody]]
    pi = tuple(node.params.get("partial_inputs") or ())
    gather = {}
    shape_ins = list(ins)
    for pos, extra in node.params.get("gather") or []:
        from pytensor_amd.dispatch.subtensor import _index_on_device

        idx = _index_on_device(env, inputs[nbody + extra])
        gather[pos] = idx
        ins[pos] = env.to_device(ins[pos])
        shape_ins[pos] = idx
    shape = _broadcast_shape(node, env.graph, shape_ins, pi)
    return ins, shape, pi, gather


@handler("Elemwise")
def elemwise(node, inputs, env):
    body = node.params["scalar"]
    g = env.graph
    if not node.params.get("gather") and _host_evaluable(body, inputs) and all(i.a.size <= HOST_MAX for i in inputs):
        return _host_eval(body, inputs)
    ins, shape, pi, gather = _prepare(node, inputs, env)
    outs, _, _ = launch_elemwise(body, ins, shape, body["out_dtypes"], None, env, pi, _placed(env, node, shape, body["out_dtypes"]), gather)
    return outs


@handler("ElemwiseReduce")
def elemwise_reduce(node, inputs, env):
    body = node.params["scalar"]
    spec = node.params["reduce"]
    g = env.graph
    ins, shape, pi, gather = _prepare(node, inputs, env)
    defer = node.params.get("defer_reduce") or ()
    done = None if defer else []  # (filled when the kernel finished its reductions itself: single pass)
    outs, parts, grid = launch_elemwise(body, ins, shape, body["out_dtypes"], spec, env, pi, _placed(env, node, shape, body["out_dtypes"]), gather, finals_out=done)
    if done:
        finals = done
    else:
        finals = finish_partials(env, spec, parts, grid, defer) if grid else [None] * len(spec)
    res = []
    for k, r in enumerate(spec):
        if r is None:
            res.append(outs[k])
        elif grid == 0:
            # empty input: identity of the reduction (elemwise.py:1607-1625)
            ident = {"Add": 0, "Mul": 1}.get(r["op"])
            if ident is None:
                raise ValueError(f"zero-size array to reduction operation {r['op'].lower()} which has no identity")
            res.append(env.to_device(HostValue(np.asarray(ident, dtype=r["dtype"]))))
        else:
            res.append(finals[k])
    return res


@handler("ElemwiseAxisReduce")
def elemwise_axis_reduce(node, inputs, env):
    """axisfuse.fuse_elemwise_axis_reduce: ``CAReduce_axes(Elemwise(inputs))`` per output, one pass over the inputs."""
    p = node.params
    body = p["scalar"]
    ins = [_scalar_or_device(env, i) for i in inputs]
    shape = _broadcast_shape(node, env.graph, ins)
    axes = [int(a) for a in p["axis"]]
    out_shape = tuple(s for d, s in enumerate(shape) if d not in axes)
    specs = []
    for r, dt in zip(p["reduce"], body["out_dtypes"]):
        acc = dt if r["op"] in ("Maximum", "Minimum") else r["acc_dtype"]  # (elemwise.py:1383-1417: no widening for max / min)
        specs.append((r["op"], acc, r["dtype"]))
    n = int(np.prod(shape)) if shape else 1
    res = None
    if n and all(shape[a] for a in axes) and len(specs) == 1 and specs[0][0] == "LogSumExp" and not body["body"] and list(body["outs"][0]) == ["i", 0] \
            and isinstance(ins[0], DeviceArray) and tuple(ins[0].shape) == tuple(shape):
        # the reduced expression is X itself (the reference's logsumexp benchmark graph after axisfuse.fuse_logsumexp)
        from pytensor_amd.dispatch.extra import logsumexp_direct

        r = logsumexp_direct(env, ins[0], axes, specs[0][2])
        if r is not None:
            res = [r]
    if res is None and n and all(shape[a] for a in axes):
        res = launch_axis_reduce(env, body, ins, tuple(shape), axes, specs, out_shape)
    if res is None:
        # shapes outside the tile (or empty): the unfused pair
        outs, _, _ = launch_elemwise(body, ins, shape, body["out_dtypes"], None, env)
        res = []
        for o, r, (_, acc, odt) in zip(outs, p["reduce"], specs):
            if r["op"] == "LogSumExp":
                res.append(_logsumexp_axis_by_axis(env, o, axes, acc, odt, out_shape))
                continue
            sub = type("_N", (), {"params": {"axis": axes, "scalar_op": r["op"], "acc_dtype": r["acc_dtype"], "dtype": r["dtype"]}})
            res.append(careduce(sub, [o], env)[0])
    return [r.view(out_shape, _cstrides(out_shape)) for r in res]


def _logsumexp_axis_by_axis(env, x, axes, acc, odt, out_shape):
    """log-sum-exp of a materialised tensor where the one-pass tile does not apply (found by the layout fuzz cases: a
    reduced axis of extent 1; also > 5 dimensions that do not merge, empty tensors).  The reduction is associative:
    one axis at a time (each a 3-d problem (A, R, B) of a contiguous tensor, which always fits the tile), last axis
    first.  Reference semantics (the graph was max + log(sum(exp(x - max))), math.py logsumexp): over extent-1 axes the
    result is x itself; an EMPTY reduced axis raises what the reference's max does."""
    if any(x.shape[a] == 0 for a in axes) and all(s for s in out_shape):
        raise ValueError("zero-size array to reduction operation maximum which has no identity")
    if x.size == 0:
        return DeviceArray.empty(out_shape, odt)
    cur = x if x.is_contiguous() else x.contiguous()
    for a in sorted(axes, reverse=True):
        shp = tuple(cur.shape)
        if shp[a] == 1:
            cur = cur.view(shp[:a] + shp[a + 1:], _cstrides(shp[:a] + shp[a + 1:]))
            continue
        A = int(np.prod(shp[:a], dtype=np.int64))
        B = int(np.prod(shp[a + 1:], dtype=np.int64))
        v3 = cur.view((A, shp[a], B), (shp[a] * B, B, 1))
        r = launch_axis_reduce(env, _identity_body(str(cur.dtype)), [v3], (A, shp[a], B), [1], [("LogSumExp", acc, acc)], (A, B))
        if r is None:
            raise RuntimeError("log-sum-exp: a contiguous (A, R, B) problem does not fit the tile")
        nshape = shp[:a] + shp[a + 1:]
        cur = r[0].view(nshape, _cstrides(nshape))
    if str(cur.dtype) != odt:
        cur = _cast(env, cur, odt)
    return cur.view(out_shape, _cstrides(out_shape))


@handler("CAReduce")
def careduce(node, inputs, env):
    p = node.params
    (x,) = inputs
    if isinstance(x, HostValue):
        x = env.to_device(x)
    axes = sorted(int(a) for a in p["axis"])
    op = p["scalar_op"]
    acc, outdt = p["acc_dtype"], p["dtype"]
    if op in ("AND", "OR", "XOR") or op in ("Maximum", "Minimum", "ScalarMaximum", "ScalarMinimum"):
        acc = str(x.dtype)  # accumulate in the input dtype (elemwise.py:1383-1417)
    keep = [d for d in range(x.ndim) if d not in axes]
    out_shape = tuple(x.shape[d] for d in keep)
    if not axes:
        return [x if str(x.dtype) == outdt else _cast(env, x, outdt)]
    if op == "MulWithoutZeros":
        return [_prod_without_zeros(x, axes, acc, outdt, env)]
    if any(x.shape[d] == 0 for d in axes) and op not in ("Add", "Mul", "AND", "OR", "XOR"):
        raise ValueError(f"zero-size array to reduction operation {op.lower()} which has no identity")
    if _ND_REDUCE and x.size and not (outdt == "bool" and acc != "bool"):
        # one pass whatever the layout: kept / reduced dimensions sorted by stride, merged, tiled
        rop = {"Maximum": "OR", "ScalarMaximum": "OR", "Minimum": "AND", "ScalarMinimum": "AND"}.get(op, op) if str(x.dtype) == "bool" else op
        res = launch_axis_reduce(env, _identity_body(x.dtype), [x], tuple(x.shape), axes, [(rop, acc, outdt)], out_shape)
        if res is not None:
            return [res[0].view(out_shape, _cstrides(out_shape))]
    # split the reduced axes into runs of adjacent dims; reduce the last run first
    runs = []
    for a in axes:
        if runs and runs[-1][-1] == a - 1:
            runs[-1].append(a)
        else:
            runs.append([a])
    cur = x
    cur_dims = list(range(x.ndim))  # original dim ids still present in `cur`
    for ri, run in enumerate(reversed(runs)):
        last = ri == len(runs) - 1
        pos = [cur_dims.index(d) for d in run]
        lo, hi = pos[0], pos[-1]
        groupA = list(range(0, lo))
        groupR = list(range(lo, hi + 1))
        groupB = list(range(hi + 1, cur.ndim))
        merged = _merge3(cur, groupA, groupR, groupB)
        if merged is None:
            cur = cur.contiguous()
            merged = _merge3(cur, groupA, groupR, groupB)
        (A, sA), (R, sR), (B, sB) = merged
        oshape = tuple(cur.shape[k] for k in groupA + groupB)
        odt = outdt if last else acc
        if last and odt == "bool" and acc != "bool":
            odt = acc
        cur = device_reduce(env, op, cur, A, R, B, sA, sR, sB, acc, odt, oshape)
        cur_dims = [d for d in cur_dims if d not in run]
    if str(cur.dtype) != outdt:
        cur = _cast(env, cur, outdt)
    return [cur.view(out_shape, _cstrides(out_shape))]


def _merge_group(arr, dims):
    """(extent, stride) of a group of adjacent dims if it is a single strided run."""
    dims = [d for d in dims if arr.shape[d] != 1]
    if not dims:
        return 1, 0
    ext, st = arr.shape[dims[-1]], arr.strides[dims[-1]]
    for d in reversed(dims[:-1]):
        if arr.strides[d] != ext * st:
            return None
        ext *= arr.shape[d]
    return ext, st


def _merge3(arr, gA, gR, gB):
    if arr.size == 0:
        ext = lambda g: int(np.prod([arr.shape[d] for d in g])) if g else 1
        return (ext(gA), 0), (ext(gR), 0), (ext(gB), 0)
    res = [_merge_group(arr, g) for g in (gA, gR, gB)]
    if any(r is None for r in res):
        return None
    return res


def _prod_without_zeros(x, axes, acc, outdt, env):
    """``ProdWithoutZeros`` (tensor/math.py:3786-3825: CAReduce over ``MulWithoutZeros`` — y if x == 0, x if y == 0, else
    x * y, identity 0 — what ``Prod.grad`` uses when its input may hold zeros): the product of the NON-ZERO entries
    along the axes, and 0 where there is none (a row of zeros, an empty extent).  Three launches out of existing
    parts: zeros -> 1 in the accumulator dtype, a ``Mul`` reduction, an ``OR`` reduction of (x != 0) as the mask."""
    from pytensor_amd.ir import Node

    xdt = str(x.dtype)
    zero = ["c", 0 if np.dtype(xdt).kind in "iub" else float(0).hex(), xdt]
    one = ["c", 1 if np.dtype(acc).kind in "iub" else float(1).hex(), acc]
    sel = {"in_dtypes": [xdt], "out_dtypes": [acc, "bool"], "outs": [["t", 2], ["t", 3]],
           "body": [{"op": "EQ", "in": [["i", 0], zero], "dtype": "bool"}, {"op": "Cast", "in": [["i", 0]], "dtype": acc},
                    {"op": "Switch", "in": [["t", 0], one, ["t", 1]], "dtype": acc}, {"op": "Invert", "in": [["t", 0]], "dtype": "bool"}]}
    (filled, nonzero), _, _ = launch_elemwise(sel, [x], tuple(x.shape), [acc, "bool"], None, env)
    (prod,) = careduce(Node("CAReduce", {"scalar_op": "Mul", "axis": list(axes), "acc_dtype": acc, "dtype": acc}, [0], [1]), [filled], env)
    (some,) = careduce(Node("CAReduce", {"scalar_op": "OR", "axis": list(axes), "acc_dtype": "bool", "dtype": "bool"}, [0], [1]), [nonzero], env)
    zo = ["c", 0 if np.dtype(outdt).kind in "iub" else float(0).hex(), outdt]
    fin = {"in_dtypes": [acc, "bool"], "out_dtypes": [outdt], "outs": [["t", 1]],
           "body": [{"op": "Cast", "in": [["i", 0]], "dtype": outdt}, {"op": "Switch", "in": [["i", 1], ["t", 0], zo], "dtype": outdt}]}
    (out,), _, _ = launch_elemwise(fin, [prod, some], tuple(prod.shape), [outdt], None, env)
    return out


def _cast(env, x, dtype):
    body = {
        "in_dtypes": [str(x.dtype)],
        "out_dtypes": [str(dtype)],
        "body": [{"op": "Cast", "in": [["i", 0]], "dtype": str(dtype)}],
        "outs": [["t", 0]],
    }
    outs, _, _ = launch_elemwise(body, [x], x.shape, [str(dtype)], None, env)
    return outs[0]


@handler("DimShuffle")
def dimshuffle(node, inputs, env):
    (x,) = inputs
    order = node.params["new_order"]
    if isinstance(x, HostValue):
        a = x.a
        keep = [o for o in order if o != "x"]
        drop = [d for d in range(a.ndim) if d not in keep]
        t = a.transpose(keep + drop).reshape([a.shape[d] for d in keep])
        idx = tuple(None if o == "x" else slice(None) for o in order)
        return [HostValue(t[idx])]
    keep = [o for o in order if o != "x"]
    for d in range(x.ndim):
        if d not in keep and x.shape[d] != 1:
            raise ValueError(f"DimShuffle: cannot drop dimension {d} of length {x.shape[d]}")
    shape, strides = [], []
    for o in order:
        if o == "x":
            shape.append(1)
            strides.append(0)
        else:
            shape.append(x.shape[o])
            strides.append(x.strides[o])
    return [x.view(shape, strides)]
