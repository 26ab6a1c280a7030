"""Elementwise-producer fusion that the reference's ``FusionOptimizer`` leaves on the table.

The reference fuses ``Elemwise`` chains into ``Composite`` ops
(pytensor/tensor/rewriting/elemwise.py:529 ``FusionOptimizer``) but stops at

* a 0-d/size-1 ``Elemwise`` with several clients reached through ``DimShuffle``
  (``exp(log_sigma)`` used by a vector graph and by the scalar epilogue),
* an ``AdvancedSubtensor`` between two ``Elemwise`` nodes (``(mu + sigma*z)[idx]``),
* a ``CAReduce`` of a variable that an ``Elemwise`` node reads anyway.

On the CPU those cost nothing; on MI355X each one is a dependent kernel launch
(≈4.5 µs inside a replayed hipGraph, more than the work of the whole node).  The passes
here remove them on the portable IR, before ``fusion.fuse_elemwise_reduce`` /
``fusion.fuse_gemv_chain`` look at the graph:

``push_gather_through_elemwise``   ``f(u, s…)[idx]  →  f(u[idx], s…)``
``inline_elemwise_producers``      producer ``Elemwise`` folded into its consumer's scalar graph
``merge_sibling_reductions``       ``CAReduce(x)`` → one more reduced output of a node reading ``x``
``dead_code_elimination``          drops what became unused
"""

from __future__ import annotations

from pytensor_amd.ir import Graph, Node, Var

MAX_INPUTS = 24  # inputs of a merged scalar graph (kernel argument block stays small)
MAX_SCALAR_BODY = 16  # scalar ops a size-1 producer may have to be *duplicated* into clients

_EW = ("Elemwise", "ElemwiseReduce")
_FUSABLE_REDUCE = {"Add", "Mul", "Maximum", "Minimum", "ScalarMaximum", "ScalarMinimum"}
_CANON = {"ScalarMaximum": "Maximum", "ScalarMinimum": "Minimum"}


def _index(g: Graph):
    producer, consumers = {}, {}
    for k, n in enumerate(g.nodes):
        for o in n.outputs:
            producer[o] = k
        for i in n.inputs:
            consumers.setdefault(i, []).append(k)
    return producer, consumers


def _copy(g: Graph, nodes) -> Graph:
    out = Graph(name=g.name)
    out.vars = dict(g.vars)
    out.inputs, out.outputs = list(g.inputs), list(g.outputs)
    out.nodes = list(nodes)
    return out


def _scalar_like(var: Var) -> bool:
    return var.kind == "tensor" and all(s == 1 for s in var.shape)


def dead_code_elimination(g: Graph) -> Graph:
    """Nodes none of whose outputs are (transitively) needed by a graph output.  Every op of
    the IR is pure (a ``CheckAndRaise`` whose value is unused is unreachable in the reference
    graph as well), so this is the reference's own notion of a live ``Apply``."""
    live = set(g.outputs)
    keep = []
    for n in reversed(g.nodes):
        if any(o in live for o in n.outputs):
            keep.append(n)
            live.update(n.inputs)
    if len(keep) == len(g.nodes):
        return g
    return _copy(g, list(reversed(keep)))


# ---------------------------------------------------------------------------
# scalar-graph surgery
# ---------------------------------------------------------------------------


def _inline_at(cb: dict, q: int, pb: dict) -> dict:
    """Scalar graph of the consumer ``cb`` with its input ``q`` replaced by the (single-output)
    producer graph ``pb``.  Consumer inputs keep their order minus ``q``; the producer's
    inputs are appended."""
    n_p = len(pb["body"])
    cmap, j = {}, 0
    for pos in range(len(cb["in_dtypes"])):
        if pos != q:
            cmap[pos] = j
            j += 1
    base = j

    def pref(r):
        return ["i", base + r[1]] if r[0] == "i" else list(r)

    pout = pref(pb["outs"][0])

    def cref(r):
        if r[0] == "i":
            return list(pout) if r[1] == q else ["i", cmap[r[1]]]
        if r[0] == "t":
            return ["t", r[1] + n_p]
        return list(r)

    # ({**b}: loop nodes carry their inner body along)
    body = [{**b, "in": [pref(r) for r in b["in"]]} for b in pb["body"]]
    body += [{**b, "in": [cref(r) for r in b["in"]]} for b in cb["body"]]
    return {
        "in_dtypes": [d for pos, d in enumerate(cb["in_dtypes"]) if pos != q] + list(pb["in_dtypes"]),
        "out_dtypes": list(cb["out_dtypes"]),
        "body": body,
        "outs": [cref(r) for r in cb["outs"]],
    }


def _dedupe_inputs(node: Node) -> Node:
    """One input position per distinct variable."""
    first, remap, keep = {}, {}, []
    for pos, v in enumerate(node.inputs):
        if v in first:
            remap[pos] = first[v]
        else:
            first[v] = remap[pos] = len(keep)
            keep.append(pos)
    if len(keep) == len(node.inputs):
        return node
    b = node.params["scalar"]

    def ref(r):
        return ["i", remap[r[1]]] if r[0] == "i" else list(r)

    nb = {
        "in_dtypes": [b["in_dtypes"][p] for p in keep],
        "out_dtypes": list(b["out_dtypes"]),
        "body": [{**e, "in": [ref(r) for r in e["in"]]} for e in b["body"]],
        "outs": [ref(r) for r in b["outs"]],
    }
    params = dict(node.params)
    params["scalar"] = nb
    return Node(node.op, params, [node.inputs[p] for p in keep], list(node.outputs))


# ---------------------------------------------------------------------------
# Elemwise producer -> Elemwise consumer
# ---------------------------------------------------------------------------


def inline_elemwise_producers(g: Graph) -> Graph:
    """Fold single-output ``Elemwise`` producers into the scalar graph of the ``Elemwise`` /
    ``ElemwiseReduce`` nodes that read them:

    * any producer whose value has exactly one client (plain vertical fusion);
    * a size-1 producer with any number of clients, also through ``DimShuffle`` views of the
      size-1 value — the few scalar flops are recomputed per client (loop-invariant in the
      generated kernels) instead of being a launch of their own.
    """
    while True:
        producer, consumers = _index(g)
        out_set = set(g.outputs)
        target = None
        for kc, nc in enumerate(g.nodes):
            if nc.op not in _EW:
                continue
            nd_c = g.vars[nc.outputs[0]].ndim
            for q, v in enumerate(nc.inputs):
                # look through size-1 DimShuffle views
                src, through = v, False
                while True:
                    kp = producer.get(src)
                    if kp is None:
                        break
                    pn = g.nodes[kp]
                    if pn.op == "DimShuffle" and _scalar_like(g.vars[pn.inputs[0]]) and _scalar_like(g.vars[src]):
                        src, through = pn.inputs[0], True
                        continue
                    break
                kp = producer.get(src)
                if kp is None:
                    continue
                P = g.nodes[kp]
                if P.op != "Elemwise":
                    continue
                pb = P.params["scalar"]
                if len(P.outputs) != 1:
                    # a multi-output producer is folded only as a size-1 value (below): the
                    # consumer gets the producer's scalar graph with the one output it reads
                    # (e.g. (-s, exp(-s)) of a scale parameter: two outputs, several clients)
                    if not all(_scalar_like(g.vars[o]) for o in P.outputs):
                        continue
                    j = P.outputs.index(src)
                    pb = {"in_dtypes": pb["in_dtypes"], "out_dtypes": [pb["out_dtypes"][j]], "body": pb["body"], "outs": [pb["outs"][j]]}
                if len(nc.inputs) - 1 + len(P.inputs) > MAX_INPUTS:
                    continue
                p_scalar = _scalar_like(g.vars[P.outputs[0]]) and all(_scalar_like(g.vars[u]) for u in P.inputs)
                single = (not through) and consumers.get(v, []).count(kc) == len(consumers.get(v, [])) and v not in out_set
                if single and g.vars[v].ndim == nd_c and len(P.outputs) == 1:
                    target = (kc, q, kp, False, pb)
                elif p_scalar and len(pb["body"]) <= MAX_SCALAR_BODY:
                    target = (kc, q, kp, True, pb)
                if target:
                    break
            if target:
                break
        if target is None:
            break
        kc, q, kp, expand, pbody = target
        nc, P = g.nodes[kc], g.nodes[kp]
        nd_c = g.vars[nc.outputs[0]].ndim
        new_vars = {}
        pre = []
        p_inputs = []
        for u in P.inputs:
            uv = g.vars[u]
            if uv.ndim == nd_c:
                p_inputs.append(u)
                continue
            # size-1 value of another rank: a broadcast-only view of the consumer's rank
            vid = max(max(g.vars), max(new_vars, default=0)) + 1
            new_vars[vid] = Var(vid, uv.dtype, (1,) * nd_c, "tensor", None, None)
            pre.append(Node("DimShuffle", {"new_order": ["x"] * nd_c}, [u], [vid]))
            p_inputs.append(vid)
        params = dict(nc.params)
        params["scalar"] = _inline_at(nc.params["scalar"], q, pbody)
        merged = Node(nc.op, params, [i for pos, i in enumerate(nc.inputs) if pos != q] + p_inputs, list(nc.outputs))
        merged = _dedupe_inputs(merged)
        nodes = list(g.nodes[:kc]) + pre + [merged] + list(g.nodes[kc + 1 :])
        g = _copy(g, nodes)
        g.vars.update(new_vars)
        g = dead_code_elimination(g)
    return g


# ---------------------------------------------------------------------------
# gather of an elementwise result  ->  elementwise of gathers
# ---------------------------------------------------------------------------


def push_gather_through_elemwise(g: Graph) -> Graph:
    """``Elemwise(u, s…)[idx]`` → ``Elemwise(u[idx], s…)`` for 1-d ``u`` and size-1 ``s``
    (reference ops: ``AdvancedSubtensor1``-style take, subtensor.py:1932, of an
    ``Elemwise``): indexing commutes with a pointwise map, and the gather of a graph
    *input* is something ``fuse_gemv_chain`` can read inside its one pass."""
    while True:
        producer, consumers = _index(g)
        out_set = set(g.outputs)
        hit = None
        for ks, S in enumerate(g.nodes):
            if S.op != "AdvancedSubtensor" or S.params.get("idx_list") != [0] or len(S.inputs) != 2:
                continue
            t, idx = S.inputs
            kp = producer.get(t)
            if kp is None or t in out_set or consumers.get(t, []) != [ks]:
                continue
            P = g.nodes[kp]
            if P.op != "Elemwise" or len(P.outputs) != 1 or g.vars[t].ndim != 1 or g.vars[idx].ndim != 1:
                continue
            vec = [u for u in P.inputs if not _scalar_like(g.vars[u])]
            if not (1 <= len(set(vec)) <= 2) or any(g.vars[u].ndim != 1 for u in P.inputs):
                continue
            hit = (ks, kp)
            break
        if hit is None:
            return g
        ks, kp = hit
        S, P = g.nodes[ks], g.nodes[kp]
        t, idx = S.inputs
        new_vars, gathers, gathered = {}, [], {}
        for u in P.inputs:
            if _scalar_like(g.vars[u]) or u in gathered:
                continue
            vid = max(max(g.vars), max(new_vars, default=0)) + 1
            new_vars[vid] = Var(vid, g.vars[u].dtype, tuple(g.vars[idx].shape), "tensor", None, None)
            gathers.append(Node("AdvancedSubtensor", dict(S.params), [u, idx], [vid]))
            gathered[u] = vid
        moved = Node("Elemwise", P.params, [gathered.get(u, u) for u in P.inputs], list(S.outputs))
        nodes = []
        for k, n in enumerate(g.nodes):
            if k == kp:
                continue
            if k == ks:
                nodes += gathers + [moved]
            else:
                nodes.append(n)
        g = _copy(g, nodes)
        g.vars.update(new_vars)


# ---------------------------------------------------------------------------
# CAReduce(x) next to an Elemwise that already reads x
# ---------------------------------------------------------------------------


def merge_sibling_reductions(g: Graph) -> Graph:
    """A full ``CAReduce`` of ``x`` becomes one more (reduced) output of an ``Elemwise`` /
    ``ElemwiseReduce`` node that reads ``x`` at full shape: ``x`` is streamed once and the
    separate reduction launch disappears.  (Run after ``fuse_elemwise_reduce``.)"""
    from pytensor_amd.fusion import _stable_toposort

    while True:
        producer, consumers = _index(g)
        hit = None
        for kc, c in enumerate(g.nodes):
            if c.op != "CAReduce" or c.params["scalar_op"] not in _FUSABLE_REDUCE:
                continue
            x = c.inputs[0]
            xv = g.vars[x]
            if xv.ndim == 0 or sorted(c.params["axis"]) != list(range(xv.ndim)):
                continue
            if xv.dtype not in ("float64", "float32", "int64", "int32"):
                continue
            for ke in consumers.get(x, []):
                e = g.nodes[ke]
                if e.op not in _EW or ke == kc:
                    continue
                spec = e.params.get("reduce") or [None] * len(e.outputs)
                # the node's iteration shape must be x's shape
                it_shape = None
                for o, s in zip(e.outputs, spec):
                    if s is None:
                        it_shape = g.vars[o].shape
                if it_shape is None:
                    it_shape = tuple(
                        next((g.vars[i].shape[d] for i in e.inputs if g.vars[i].shape[d] != 1), 1) for d in range(xv.ndim)
                    )
                if len(it_shape) != xv.ndim or any(a == 1 and b != 1 for a, b in zip(xv.shape, it_shape)):
                    continue
                if len(e.inputs) >= MAX_INPUTS or _depends_on(g, producer, e, c.outputs[0]):
                    continue
                hit = (kc, ke)
                break
            if hit:
                break
        if hit is None:
            return g
        kc, ke = hit
        c, e = g.nodes[kc], g.nodes[ke]
        x = c.inputs[0]
        b = e.params["scalar"]
        spec = list(e.params.get("reduce") or [None] * len(e.outputs))
        nb = {
            "in_dtypes": list(b["in_dtypes"]),
            "out_dtypes": list(b["out_dtypes"]) + [g.vars[x].dtype],
            "body": list(b["body"]),
            "outs": [list(r) for r in b["outs"]] + [["i", e.inputs.index(x)]],
        }
        spec.append(
            {
                "op": _CANON.get(c.params["scalar_op"], c.params["scalar_op"]),
                "acc_dtype": c.params["acc_dtype"],
                "dtype": c.params["dtype"],
            }
        )
        merged = Node("ElemwiseReduce", {"scalar": nb, "reduce": spec}, list(e.inputs), list(e.outputs) + [c.outputs[0]])
        nodes = [merged if k == ke else n for k, n in enumerate(g.nodes) if k != kc]
        g = _copy(g, _stable_toposort(nodes))


def _depends_on(g: Graph, producer, node: Node, var: int) -> bool:
    seen, stack = set(), list(node.inputs)
    while stack:
        v = stack.pop()
        if v == var:
            return True
        if v in seen:
            continue
        seen.add(v)
        k = producer.get(v)
        if k is not None:
            stack.extend(g.nodes[k].inputs)
    return False
