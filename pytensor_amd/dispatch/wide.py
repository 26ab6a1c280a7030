"""``MultiElemwise`` / ``ScatterScalars`` — wide graphs (widefuse.py): many independent terms.

``MultiElemwise``: the member ``ElemwiseReduce`` nodes (reference: ``Elemwise`` +
``CAReduce``, pytensor/tensor/elemwise.py:375, 1233) in ONE launch when every member is a flat,
contiguous, fully reduced loop — grid (terms, workgroups per term) (codegen.multi_flat_source); each term
gets workgroups in proportion to its work, few enough that the per-output partials (handed to the ``Tail``
kernel unfinished) are at most 64 values.  Anything else runs the members one by one through their
own handler.  ``ScatterScalars`` outside a ``Tail`` is its member ``IncSubtensor`` chain.
"""

from __future__ import annotations

import ctypes as C
import struct

from pytensor_amd import codegen, ffi, kernel_cache
from pytensor_amd.device import DeviceArray
from pytensor_amd.dispatch import HANDLERS, handler
from pytensor_amd.dispatch.elemwise import (
    BLOCK, EW_UNROLL, _body_key, _scalar_bits, _scalar_or_device, alloc_partials, finish_partials,
)
from pytensor_amd.executor import HostValue

MAX_ARG_BYTES = 3900
import os

TOTAL_GROUPS = int(os.environ.get("PTHIP_WIDE_GROUPS", 2048))  # workgroups of one launch, shared among the terms
TERM_CAP = int(os.environ.get("PTHIP_WIDE_CAP", 64))
# OPT-IN: every term finishes its own reductions (its last workgroup folds the term's <= 64 pairs; codegen.multi_flat_source
# `finish`), the Tail behind it receives finished scalars instead of 144 partial arrays to fold in single waves.  Measured on
# north_star's 48-term graph (profiles/r9_wide200_self_finish.md): the two tail launches (18 + 30 us) become one of 51 us —
# the folds were never what a tail kernel's time is made of — and the many-term kernel goes from 140 to 145 us: 0.387 ms
# against 0.385.  Kept for graphs whose terms have no Tail behind them; tests/test_gpu_wide_partials.py runs it.
WIDE_FINISH = os.environ.get("PTHIP_WIDE_FINISH", "0") == "1"
WIDE_UNROLL = int(os.environ.get("PTHIP_WIDE_UNROLL", 0))  # packs per loop iteration of a term (0: the Elemwise default)
WIDE_PREFETCH_MIN = float(os.environ.get("PTHIP_WIDE_PREFETCH_MIN", "inf"))  # software-pipelined loop for terms costing at least this  # per term (the partials a Tail kernel folds in one pass)


# rough VALU instructions per element of a scalar op on gfx950 (fp64): what the split of a launch among its terms is
# weighted by (an estimate only has to rank the families; measured on the ISA: pt_exp 24, pt_log1p ~60 — the device library's
# is ~125 — a division by a varying denominator ~12, pt_tanh ~45)
_L1P = float(os.environ.get("PTHIP_WIDE_COST_LOG1P", 60))
_OP_COST = {"Exp": 24, "Log": 50, "Log1p": _L1P, "Log2": 35, "Log10": 35, "Expm1": 40, "Sigmoid": 40, "Softplus": 80, "Tanh": 85, "Pow": 100, "TrueDiv": 12,
            "Sqrt": 12, "Erf": 60, "Erfc": 80, "GammaLn": 150, "Psi": 150, "Sin": 60, "Cos": 60, "Log1mexp": 70, "ScalarLoop": 400}


def _term_cost(body, modes=None) -> float:
    """Estimated VALU instructions per ELEMENT.  Scalar ops none of whose operands vary with the element (``exp(-log_sigma)``
    of a broadcast parameter) are hoisted out of the loop by the compiler and cost nothing; ``sigmoid`` and ``softplus`` of
    one operand share their exp and a division, divisions by one denominator share a reciprocal when every output is
    summed (codegen.emit_body) — the estimate has to rank the families the way the generated code costs them, or the
    cheap families finish early and the launch ends on a few workgroups of the dear one."""
    nodes = body["body"]
    if modes is None or any(n["op"] in ("ScalarLoop", "LoopOut") for n in nodes):
        return 8.0 + sum(_OP_COST.get(op, 1.5) for op in codegen.body_ops(body))
    varies = []
    cost = 8.0
    seen_den, seen_sig = set(), {}
    for n in nodes:
        v = any((r[0] == "i" and modes[r[1]] in "VG") or (r[0] == "t" and varies[r[1]]) for r in n["in"])
        varies.append(v)
        if not v:
            continue
        op = n["op"]
        if op == "TrueDiv" and n["in"][1][0] in ("i", "t"):
            den = (n["in"][1][0], n["in"][1][1])
            cost += 3.0 if den in seen_den else 28.0
            seen_den.add(den)
        elif op in ("Sigmoid", "Softplus") and n["in"][0][0] in ("i", "t"):
            arg = (n["in"][0][0], n["in"][0][1])
            other = seen_sig.get(arg)
            if other is not None and other != op:
                cost += _L1P if op == "Softplus" else 4.0  # the shared exp and reciprocal are paid: what is left is log1p / a select
            else:
                cost += 24.0 + (28.0 if op == "Sigmoid" else 44.0)
            seen_sig[arg] = op
        else:
            cost += _OP_COST.get(op, 1.5)
    return cost


def _run_members(node, inputs, env):
    vals = dict(zip(node.inputs, inputs))
    saved = env.donated
    env.donated = frozenset()
    try:
        for sub in node.params["nodes"]:
            ins = [vals[v] if v in vals else env.exe._const(v, env) for v in sub.inputs]
            for o, val in zip(sub.outputs, HANDLERS[sub.op](sub, ins, env)):
                vals[o] = val
    finally:
        env.donated = saved
    return [vals[o] for o in node.outputs]


@handler("ScatterScalars")
def scatter_scalars(node, inputs, env):
    # (with `base_fill` the first member is the Alloc the fused node absorbed: its fill value is a
    #  constant of the graph, its length this node's first input)
    return _run_members(node, inputs, env)


def _flat_term(term, ins):
    """(modes, n, vec) when the term is a flat loop over one contiguous extent, else None"""
    body = term["scalar"]
    n, modes = None, []
    for a in ins:
        if isinstance(a, HostValue):
            if a.a.size != 1:
                return None
            modes.append("C")
        elif a.size == 1:
            modes.append("S")
        else:
            if not a.is_contiguous() or (n is not None and a.size != n):
                return None
            n = a.size
            modes.append("V")
    if n is None or n == 0:
        return None
    dts = [body["in_dtypes"][k] for k, m in enumerate(modes) if m == "V"]
    vec = codegen._vec_width(dts) if dts else 1
    if vec > 1 and (any(a.ptr % 16 for a, m in zip(ins, modes) if m == "V") or n < vec):
        vec = 1
    return "".join(modes), n, vec


@handler("MultiElemwise")
def multi_elemwise(node, inputs, env):
    terms = node.params["terms"]
    defer = set(node.params.get("defer_reduce") or ())
    per_term, pos = [], 0
    for t in terms:
        ins = [_scalar_or_device(env, i) for i in inputs[pos : pos + t["n_inputs"]]]
        pos += t["n_inputs"]
        ft = _flat_term(t, ins)
        if ft is None or not codegen.supported(t["scalar"]):
            return _run_members(node, inputs, env)
        per_term.append((t, ins, *ft))
    nt = len(per_term)
    # the same workgroup count for every term (gridDim.x), bounded so that a term's partials fit
    # one pass of the tail kernel; at least enough to give each term a few waves per XCD
    # Workgroups per term in proportion to its work (elements x estimated instructions per element): with the same
    # count for every term the launch lasts as long as its most expensive family — north_star's 48-term graph: 221 us
    # at a mean occupancy of 7 waves per CU, the logistic terms (exp + log1p per element) still running when the
    # normal ones were long done (profiles/r5h_wide_multi_pmc.md).  At most 64 per term (the partials a Tail kernel
    # folds in one pass), at least 4.
    work = [n * _term_cost(t["scalar"], modes) for t, _, modes, n, _ in per_term]
    tot = float(sum(work)) or 1.0
    groups = []
    for (t, ins, modes, n, vec), w in zip(per_term, work):
        units = (n // vec + EW_UNROLL - 1) // EW_UNROLL if vec > 1 else n
        cap = max(1, (units + BLOCK - 1) // BLOCK)
        groups.append(max(1, min(cap, TERM_CAP, max(4, int(round(TOTAL_GROUPS * w / tot))))))
    gx = max(groups)
    finish = WIDE_FINISH and gx <= BLOCK and all(
        t["reduce"] and all(r is not None and r["acc_dtype"] == "float64" and r["dtype"] == "float64" for r in t["reduce"]) for t, *_ in per_term)
    specs, args, out_pos, results, o0 = [], [], 0, [], 0
    gen_terms = []
    blk, blk_words = None, 0
    if finish:
        lay = [codegen.multi_finish_layout(len(t["reduce"]), gt) for (t, *_), gt in zip(per_term, groups)]
        # pairs are never cleared from here (dispatch/elemwise.py: the last workgroup zeroes what it consumed; stale bits
        # validate with probability 2^-64)
        blk = DeviceArray.empty((sum(w for _, w in lay),), "float64")
        env.keepalive.append(blk)
    for ti, ((t, ins, modes, n, vec), gt) in enumerate(zip(per_term, groups)):
        body, spec = t["scalar"], t["reduce"]
        rs = [(r["op"], r["acc_dtype"]) for r in spec]
        gen_terms.append({"body": body, "modes": modes, "vec": vec, "rs": rs, "unroll": WIDE_UNROLL or EW_UNROLL, "groups": gt,
                          "prefetch": vec > 1 and _term_cost(body, modes) >= WIDE_PREFETCH_MIN})
        args.append(n)
        for k, (a, m) in enumerate(zip(ins, modes)):
            args.append(_scalar_bits(a, body["in_dtypes"][k]) if m == "C" else a.ptr)
        if finish:
            fin0, words = lay[ti]
            args.append(blk.ptr + 8 * blk_words)
            results += [blk.view((), (), blk_words + fin0 + k) for k in range(len(spec))]
            blk_words += words
        else:
            parts = alloc_partials(spec, gt)
            args += [p.ptr for p in parts]
            specs.append((spec, parts, gt))
    if finish:
        first = C.c_void_p()
        ffi.check(env.lib.pthip_ticket_slots(nt, C.byref(first)))
        args += [first.value, env.lib.pthip_status_ptr()]
    buf = struct.pack(f"<{len(args)}q", *args)
    if len(buf) > MAX_ARG_BYTES:
        return _run_members(node, inputs, env)
    key = codegen.source_key(repr([(_body_key(g["body"]), g["modes"], g["vec"], g["rs"], g["groups"], g["unroll"], g["prefetch"]) for g in gen_terms]) + repr(finish))
    name = f"multi_{key[:16]}_t{nt}" + ("_1p" if finish else "")
    src = codegen.multi_flat_source(name, gen_terms, finish=finish)
    fn = kernel_cache.get_function(src, name)
    env.timed(name, lambda: ffi.check(env.lib.pthip_launch(fn, nt, gx, 1, BLOCK, 1, 1, 0, buf, len(buf))))
    if finish:
        return results
    for spec, parts, gt in specs:
        d = {k - o0 for k in defer if o0 <= k < o0 + len(spec)}
        results += finish_partials(env, spec, parts, gt, d)
        o0 += len(spec)
    return results
