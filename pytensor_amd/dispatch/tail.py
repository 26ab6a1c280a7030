"""``Tail`` — the small nodes at the end of an evaluation in two launches (tailfuse.py).

The member nodes keep their reference meaning (``GemvFinish`` = second stage + epilogue of
pytensor/tensor/blas/gemv.py:64-108, ``Elemwise`` / ``ElemwiseReduce`` = elemwise.py:375 /
1233, ``DimShuffle`` = a view, elemwise.py:41); this handler only decides *how many launches*
they cost.  Extents are checked at run time: when every operand is a scalar or a short vector,
``pthip_multi_finish`` shrinks the large partial slabs and ONE generated single-workgroup kernel
does the rest (``codegen.tail_chain_source``); otherwise the members run through their own
handlers, exactly as before the fusion.
"""

from __future__ import annotations

import ctypes as C
import os
import struct

import numpy as np

from pytensor_amd import codegen, ffi, kernel_cache
from pytensor_amd.device import DeviceArray
from pytensor_amd.dispatch import HANDLERS, handler
from pytensor_amd.dispatch.elemwise import _body_key, _scalar_bits
from pytensor_amd.executor import DeferredReduce, HostValue

MAX_LEN = 4096  # elements per vector the single workgroup loops over
MAX_LDS = 48 << 10
SHRINK_MIN = 8192  # slabs below this many elements are summed by the tail kernel directly
SHRINK_ROWS = 16
_FLOAT = ("float64", "float32")


class _Infeasible(Exception):
    pass


class _Val:
    """What the planner knows about a value: where it lives and its shape."""

    __slots__ = ("ref", "shape", "dtype", "host", "dev")

    def __init__(self, ref, shape, dtype, host=None, dev=None):
        self.ref, self.shape, self.dtype, self.host, self.dev = ref, tuple(shape), str(dtype), host, dev

    @property
    def size(self):
        n = 1
        for s in self.shape:
            n *= s
        return n


def _vec_len(shape):
    """length of a value that is a scalar or a vector up to size-1 dims; None otherwise"""
    big = [s for s in shape if s != 1]
    if len(big) > 1:
        return None
    return big[0] if big else 1


class _Planner:
    def __init__(self, env):
        self.env = env
        self.ext, self.ext_args = [], []  # structural descriptors / runtime arguments
        self.ext_len, self.step_n = [], []  # actual extents (size classes of the preloading kernel)
        self.slots, self.slot_len = [], []
        self.steps, self.step_args = [], []
        self.keep = []
        self.shrink = []  # (op code, slab, nparts, M, S, out buffer)

    def add_ext(self, v) -> _Val:
        if isinstance(v, HostValue):
            if v.a.size != 1:
                raise _Infeasible("host array operand")
            dt = str(v.a.dtype)
            if dt not in codegen.CTYPE:
                raise _Infeasible(dt)
            self.ext.append({"kind": "C", "dtype": dt})
            self.ext_len.append(1)
            self.ext_args.append([("q", _scalar_bits(v, dt))])
            return _Val(("e", len(self.ext) - 1), v.a.shape, dt, host=v)
        if isinstance(v, DeferredReduce):
            return _Val(None, (), v.spec["dtype"], dev=v)  # consumed through an `rsum` step
        if not isinstance(v, DeviceArray):
            raise _Infeasible(type(v).__name__)
        dt = str(v.dtype)
        if dt not in codegen.CTYPE:
            raise _Infeasible(dt)
        n = _vec_len(v.shape)
        if n is None or n > MAX_LEN:
            return _Val(None, v.shape, dt, dev=v)  # only usable as a partial slab
        self.ext_len.append(n)
        if n == 1:
            self.ext.append({"kind": "S", "dtype": dt})
            self.ext_args.append([("q", v.ptr)])
        else:
            stride = [st for s, st in zip(v.shape, v.strides) if s != 1][0]
            self.ext.append({"kind": "V", "dtype": dt})
            self.ext_args.append([("q", v.ptr), ("q", stride)])
        self.keep.append(v)
        return _Val(("e", len(self.ext) - 1), v.shape, dt, dev=v)

    def add_slab(self, arr: DeviceArray) -> tuple:
        self.ext.append({"kind": "P", "dtype": str(arr.dtype)})
        self.ext_len.append(arr.size)
        self.ext_args.append([("q", arr.ptr)])
        self.keep.append(arr)
        return ("e", len(self.ext) - 1)

    def new_slot(self, shape, dtype) -> _Val:
        n = 1
        for s in shape:
            n *= s
        if n > MAX_LEN or _vec_len(shape) is None:
            raise _Infeasible("large intermediate")
        self.slots.append({"dtype": str(dtype), "scalar": n == 1})
        self.slot_len.append(n)
        return _Val(("l", len(self.slots) - 1), shape, dtype)

    def materialise(self, val: _Val) -> _Val:
        """A deferred reduction becomes an LDS scalar through an `rsum` step (once)."""
        if val.ref is not None:
            return val
        d = val.dev
        if not isinstance(d, DeferredReduce):
            raise _Infeasible("large operand")
        if d.spec["op"] not in codegen.REDUCE_OPS or d.spec["acc_dtype"] not in codegen.CTYPE:
            raise _Infeasible("reduction op")
        out = self.new_slot((), d.spec["dtype"])
        src = self.add_slab(d.parts)
        self.steps.append({"op": "rsum", "src": src, "red": d.spec["op"], "acc_dtype": d.spec["acc_dtype"], "dtype": d.spec["dtype"],
                           "out": out.ref[1]})
        self.step_args.append([("q", d.grid)])
        self.step_n.append(int(d.grid))
        val.ref, val.shape = out.ref, ()
        return val


def _plan(node, inputs, env):
    P = _Planner(env)
    vals = {vid: P.add_ext(v) for vid, v in zip(node.inputs, inputs)}
    g = env.graph

    def const_scalar(vid):
        v = vals.get(vid)
        if v is None:
            c = g.vars[vid].const
            if c is None:
                raise _Infeasible("unbound operand")
            v = vals[vid] = P.add_ext(HostValue(np.asarray(c)) if np.asarray(c).size == 1 else env.exe._const(vid, env))
        return v

    for sub in node.params["nodes"]:
        for vid in sub.inputs:
            if vid not in vals:
                const_scalar(vid)
        if sub.op == "DimShuffle":
            x = vals[sub.inputs[0]]
            if x.ref is None and not isinstance(x.dev, DeferredReduce):
                raise _Infeasible("view of a large operand")
            x = P.materialise(x)
            order = sub.params["new_order"]
            shape = tuple(1 if o == "x" else x.shape[o] for o in order)
            if _vec_len(shape) is None:
                raise _Infeasible("view shape")
            vals[sub.outputs[0]] = _Val(x.ref, shape, x.dtype, host=x.host, dev=x.dev)
        elif sub.op == "GemvFinish":
            part, y, a, b = (vals[v] for v in sub.inputs)
            slab = part.dev
            if not isinstance(slab, DeviceArray) or slab.ndim != 2 or not slab.is_contiguous() or str(slab.dtype) not in _FLOAT:
                raise _Infeasible("partial slab layout")
            if a.host is None or b.host is None:
                raise _Infeasible("device alpha/beta")
            if str(g.vars[sub.outputs[0]].dtype) != str(slab.dtype):
                raise _Infeasible("finish in another precision than its output")  # (float32 graph, float64 slabs: the member's own handler casts)
            alpha, beta = float(a.host.a.reshape(-1)[0]), float(b.host.a.reshape(-1)[0])
            rows, M = slab.shape
            if M > MAX_LEN:
                raise _Infeasible("long finish")
            if rows * M >= SHRINK_MIN and rows > SHRINK_ROWS:
                S = SHRINK_ROWS
                small = DeviceArray.empty((S, M), slab.dtype)
                P.shrink.append((ffi.REDUCE_CODE["Add"], slab, rows, M, S, small))
                slab, rows = small, S
            src = P.add_slab(slab)
            out = P.new_slot((M,), slab.dtype)
            st = {"op": "finish", "src": src, "dtype": str(slab.dtype), "y": None, "ymode": "V", "out": out.ref[1]}
            if beta != 0.0:
                y = P.materialise(y)
                n = _vec_len(y.shape)
                if n not in (1, M):
                    raise ValueError(f"Shape mismatch: y.shape[0] != A.shape[0] ({y.shape}, {M})")
                st["y"], st["ymode"] = y.ref, ("V" if n == M and M > 1 else "S")
            P.steps.append(st)
            P.step_args.append([("q", rows), ("q", M), ("d", alpha), ("d", beta)])
            P.step_n.append((int(rows), int(M)))
            vals[sub.outputs[0]] = out
        elif sub.op == "ScatterScalars":
            # widefuse.collect_scalar_updates: out = copy(base); out[k_j] (=|+=) y_j in chain order
            fill = sub.params.get("base_fill")
            out_dt = str(g.vars[sub.outputs[0]].dtype)
            if fill is not None:
                # base = alloc(fill, n): the length is host shape arithmetic, the value a constant
                ln = vals[sub.inputs[0]]
                if ln.host is None:
                    raise _Infeasible("scatter length on the device")
                n = int(np.asarray(ln.host.a).reshape(()))
                base = P.add_ext(HostValue(np.asarray(fill, dtype=out_dt)))
                base = _Val(base.ref, (n,), out_dt, host=base.host)
                if n < 1 or n > MAX_LEN:
                    raise _Infeasible("scatter base")
            else:
                base = P.materialise(vals[sub.inputs[0]])
                n = _vec_len(base.shape)
                if n is None or n > MAX_LEN or len(base.shape) != 1:
                    raise _Infeasible("scatter base")
            ys = [P.materialise(vals[v]) for v in sub.inputs[1:]]
            if any(y.size != 1 for y in ys) or base.dtype not in codegen.CTYPE:
                raise _Infeasible("scatter operand")
            idx = []
            for k in sub.params["indices"]:
                if k < -n or k >= n:
                    raise IndexError(f"index {k} is out of bounds for axis 0 with size {n}")
                idx.append(k + n if k < 0 else k)
            out = P.new_slot((n,), base.dtype)
            P.steps.append({"op": "scatter", "base": base.ref, "bmode": "V" if (n > 1 and fill is None) else "S", "ys": [y.ref for y in ys],
                            "indices": idx, "set": [bool(b) for b in sub.params["set"]], "out": out.ref[1], "dtype": base.dtype})
            P.step_args.append([("q", n)])
            P.step_n.append(int(n))
            vals[sub.outputs[0]] = out
        else:  # Elemwise / ElemwiseReduce
            body = sub.params["scalar"]
            if not codegen.supported(body):
                raise _Infeasible("scalar op")
            ins = [P.materialise(vals[v]) for v in sub.inputs]
            nd = max((len(v.shape) for v in ins), default=0)
            shape = [1] * nd
            for v in ins:
                sh = (1,) * (nd - len(v.shape)) + v.shape
                for d, s in enumerate(sh):
                    if s != 1:
                        if shape[d] not in (1, s):
                            raise ValueError(f"Runtime broadcasting not allowed: operand shapes {[i.shape for i in ins]}")
                        shape[d] = s
            n = _vec_len(shape)
            if n is None or n > MAX_LEN:
                raise _Infeasible("large elementwise")
            modes = "".join("C" if (v.ref[0] == "e" and P.ext[v.ref[1]]["kind"] == "C") else ("V" if (v.size == n and n > 1) else "S") for v in ins)
            for pos, (v, m) in enumerate(zip(ins, modes)):
                # the reference forbids broadcasting a runtime length-1 dim that is not statically 1
                # (elemwise.py:825-840); the member's own handler raises the exact message: let it
                if m == "S" and n > 1 and any(s is None for s in g.vars[sub.inputs[pos]].shape):
                    raise _Infeasible("runtime broadcast check")
            spec = sub.params.get("reduce") or [None] * len(body["out_dtypes"])
            outs, red = [], []
            for q, (r, dt) in enumerate(zip(spec, body["out_dtypes"])):
                if r is None:
                    o = P.new_slot(tuple(shape), dt)
                    red.append(None)
                else:
                    if r["op"] not in codegen.REDUCE_OPS:
                        raise _Infeasible("reduction op")
                    o = P.new_slot((), r["dtype"])
                    red.append((r["op"], r["acc_dtype"], r["dtype"]))
                outs.append(o)
                vals[sub.outputs[q]] = o
            P.steps.append({"op": "ew", "body": body, "ins": [v.ref for v in ins], "modes": modes, "outs": [o.ref[1] for o in outs], "reduce": red})
            P.step_args.append([("q", n)])
            P.step_n.append(int(n))
    results = []
    for vid in node.outputs:
        v = vals[vid]
        if v.ref is None or v.ref[0] != "l":
            raise _Infeasible("pass-through output")
        results.append(v)
    return P, results


# measured on config #4 (profiles/r4b_c4_ab.txt): the fused launch is 28 us where multi_finish + gap + tail are 18 — every
# one of the ~256 shrink workgroups carries the chain kernel's registers and LDS, and the ticket's agent-scope release /
# acquire fences write back and invalidate L2.  Kept as an opt-in (PTHIP_TAIL_FUSE_SHRINK=1), default off.
_FUSE_SHRINK = os.environ.get("PTHIP_TAIL_FUSE_SHRINK", "0") == "1"


def _run_fused(node, P, results, env, inputs=()):
    lib = env.lib
    # launch 1: every large slab -> <= 16 rows.  When the slabs fit ONE task table of one dtype, that work
    # becomes the prologue of the chain's own launch instead (last-workgroup ticket, codegen._tail_prologue)
    by_dt = {}
    for t in P.shrink:
        by_dt.setdefault(str(t[1].dtype), []).append(t)
    fuse_shrink = None
    if _FUSE_SHRINK and len(by_dt) == 1 and 1 <= len(P.shrink) <= codegen.TAIL_SHRINK_MAX_TASKS:
        fuse_shrink = {"dtype": next(iter(by_dt))}
    # the plan's device-side join (plan.py _DEV_JOIN; include/pthip.h pthip_join_signal): the chain launch waits for the
    # other stream's signal word and puts it back
    status0 = getattr(env, "tail_status", None) or (0, 0, 0)
    join_ptr = getattr(env, "tail_join", 0) if (status0[0] and fuse_shrink is None) else 0
    # only the chain launch can wait (and tell the host when it gave up): a slab that comes from the other stream's
    # segment (A) keeps the event (slabs are the operands too long for the chain kernel itself, Plan.add_ext)
    if join_ptr and P.shrink:
        seg = getattr(env.exe, "segments", None)
        prod = {o: k for k, n in enumerate(env.graph.nodes) for o in n.outputs}
        for vid, val in zip(node.inputs, inputs):
            k = prod.get(vid)
            if k is None or not (seg is None or seg[k] == 0):
                continue
            if isinstance(val, DeferredReduce):
                join_ptr = 0  # unfinished partials of a segment-A kernel: the shrink launch reads them, and it cannot wait
            elif isinstance(val, DeviceArray):
                n = _vec_len(val.shape)
                if n is None or n > MAX_LEN:
                    join_ptr = 0
    # launch 2: the chain
    # one-element slots first, at static 16-byte cells; vector slots behind them at run-time offsets
    off = 16 * sum(1 for s in P.slots if s["scalar"])
    offs = []
    for s, n in zip(P.slots, P.slot_len):
        if s["scalar"]:
            continue
        offs.append(off)
        off += (max(n, 1) * np.dtype(s["dtype"]).itemsize + 15) // 16 * 16
    if off > MAX_LDS:
        raise _Infeasible("LDS")
    outs, out_args = [], []
    for vid, v in zip(node.outputs, results):
        dst = env.placement.get(vid) if env.placement else None
        if not (dst is not None and dst.shape == v.shape and str(dst.dtype) == v.dtype and dst.is_contiguous()):
            dst = DeviceArray.empty(v.shape, v.dtype)
        outs.append(dst)
        out_args += [("q", dst.ptr), ("q", v.size)]
    spec = {"ext": P.ext, "slots": P.slots, "steps": P.steps, "outs": [v.ref[1] for v in results]}
    key = repr([[(e["kind"], e["dtype"]) for e in P.ext], [(s["dtype"], s["scalar"]) for s in P.slots],
                [{k: (_body_key(v) if k == "body" else v) for k, v in st.items()} for st in P.steps], spec["outs"]])
    # short operands (the usual case: K- and G-vectors, <= 16-row slabs): the form that requests
    # every global operand before the first step; size classes are part of the kernel identity
    sizes = codegen.tail_preload_sizes(spec, P.ext_len, P.step_n) if os.environ.get("PTHIP_TAIL_PRELOAD", "1") != "0" else None
    args = [a for e in P.ext_args for a in e] + [("q", o) for o in offs] + [a for s in P.step_args for a in s] + out_args
    status = getattr(env, "tail_status", None) or (0, 0, 0)
    join = join_ptr
    args += [("q", status[0]), ("q", status[1]), ("q", status[2] if len(status) > 2 else 0), ("q", join)]
    if join:
        env.tail_join_used = True
    buf = struct.pack("<" + "".join(a[0] for a in args), *[a[1] for a in args])
    if len(buf) > 4000:
        # BEFORE the shrink launch: a chain that does not fit is run as two (`_run_split`), each half shrinking its own
        # slabs — a shrink launched here first would be recorded into the plan and replayed for nothing (north_star's
        # 48-term graph carried one such 4.7 us launch per evaluation until round 6)
        raise _Infeasible("kernel argument block")
    if fuse_shrink is None:
        for dt, ts in by_dt.items():
            for c0 in range(0, len(ts), 16):
                chunk = ts[c0 : c0 + 16]
                n = len(chunk)
                ffi.check(lib.pthip_multi_finish(
                    ffi.np_dtype_code(dt), n, (C.c_int * n)(*[t[0] for t in chunk]), (C.c_void_p * n)(*[t[1].ptr for t in chunk]),
                    (C.c_int64 * n)(*[t[2] for t in chunk]), (C.c_int64 * n)(*[t[3] for t in chunk]), (C.c_int * n)(*[t[4] for t in chunk]),
                    (C.c_void_p * n)(*[t[5].ptr for t in chunk])))
    grid = 1
    if fuse_shrink is not None:
        tasks, grid = codegen.tail_shrink_pack([(t[0], t[1].ptr, t[2], t[3], t[4], t[5].ptr) for t in P.shrink])
        if len(buf) + len(tasks) + 8 > 4000:
            # the task table does not fit the 4 KB argument block next to the chain's own arguments: two launches
            fuse_shrink = None
            t = P.shrink
            ffi.check(lib.pthip_multi_finish(
                ffi.np_dtype_code(str(t[0][1].dtype)), len(t), (C.c_int * len(t))(*[x[0] for x in t]), (C.c_void_p * len(t))(*[x[1].ptr for x in t]),
                (C.c_int64 * len(t))(*[x[2] for x in t]), (C.c_int64 * len(t))(*[x[3] for x in t]), (C.c_int * len(t))(*[x[4] for x in t]),
                (C.c_void_p * len(t))(*[x[5].ptr for x in t])))
            grid = 1
        else:
            slot = C.c_void_p()
            ffi.check(lib.pthip_ticket_slot(C.byref(slot)))
            buf += tasks + struct.pack("<Q", slot.value)
    name = "tail_" + codegen.source_key(key + repr(sizes) + repr(fuse_shrink))[:16]
    src = codegen.tail_chain_source(name, spec, sizes, shrink=fuse_shrink)
    fn = kernel_cache.get_function(src, name)
    if len(buf) > 4000:
        raise _Infeasible("kernel argument block")  # (4 KB kernarg limit: run the members instead)
    kt = env.kernel_timer
    tok = kt.begin() if kt is not None else None
    ffi.check(lib.pthip_launch(fn, grid, 1, 1, codegen.TAIL_BLOCK, 1, 1, max(off, 16), buf, len(buf)))
    if kt is not None:
        kt.end(name, tok)
    if status[1]:
        env.tail_status_done = True
    if len(status) > 2 and status[2]:
        env.tail_done_word = True  # this launch stores the completion word; the plan polls it if nothing follows
        env.tail_launch_mark = int(lib.pthip_launch_count())
    return outs


def _run_members(node, inputs, env):
    """The members one by one through their own handlers (what the graph was before the fusion)."""
    vals = dict(zip(node.inputs, inputs))
    env.donated = frozenset()  # (positions refer to the Tail node, not to its members)
    for sub in node.params["nodes"]:
        ins = []
        for vid in sub.inputs:
            v = vals.get(vid)
            if v is None:
                v = env.exe._const(vid, env)
            if isinstance(v, DeferredReduce):
                v = v.force(env)
                vals[vid] = v
            ins.append(v)
        outs = HANDLERS[sub.op](sub, ins, env)
        for o, val in zip(sub.outputs, outs):
            vals[o] = val
    return [vals[o] for o in node.outputs]


def _run_split(node, inputs, env):
    """A chain too long for ONE launch (the 4 KB kernel-argument block, the LDS budget: a model with dozens of
    likelihood terms hands the tail hundreds of scalars) as two chains, recursively: members [0, h) and [h, n), the
    values crossing the cut become outputs of the first and inputs of the second.  Only the LAST launch carries the
    plan's status / completion word; none of them joins the other stream on the device (the plan keeps its event)."""
    from pytensor_amd.ir import Node

    members = node.params["nodes"]
    h = len(members) // 2
    first, second = members[:h], members[h:]
    made1 = {o for m in first for o in m.outputs}
    need2 = {i for m in second for i in m.inputs}
    outs1 = [v for v in dict.fromkeys(o for m in first for o in m.outputs) if v in need2 or v in node.outputs]
    need1 = {i for m in first for i in m.inputs}
    vals = dict(zip(node.inputs, inputs))
    ins1 = [v for v in node.inputs if v in need1]
    status, join = getattr(env, "tail_status", None), getattr(env, "tail_join", 0)
    env.tail_status, env.tail_join = None, 0
    try:
        r1 = tail(Node("Tail", {"nodes": first}, ins1, outs1), [vals[v] for v in ins1], env)
    finally:
        env.tail_status = status
    vals.update(zip(outs1, r1))
    made2 = {o for m in second for o in m.outputs}
    ins2 = [v for v in dict.fromkeys([*node.inputs, *outs1]) if v in need2]
    outs2 = [v for v in dict.fromkeys(node.outputs) if v in made2]
    try:
        r2 = tail(Node("Tail", {"nodes": second}, ins2, outs2), [vals[v] for v in ins2], env)
    finally:
        env.tail_join = join
    vals.update(zip(outs2, r2))
    return [vals[o] for o in node.outputs]


@handler("Tail")
def tail(node, inputs, env):
    try:
        P, results = _plan(node, inputs, env)
        return _run_fused(node, P, results, env, inputs)
    except _Infeasible as e:
        if str(e) in ("kernel argument block", "LDS") and len(node.params["nodes"]) >= 8:
            return _run_split(node, inputs, env)
        return _run_members(node, inputs, env)
