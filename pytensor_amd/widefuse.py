"""Passes for *wide* graphs: many independent terms, each a streaming kernel plus scalar bookkeeping.

A PyMC model with dozens of likelihood terms (north_star: "≈200 fused Elemwise"; SURVEY Appendix B
asks for a wide synthetic model next to the single hierarchical-normal) lowers to one fused
``ElemwiseReduce`` per term — each far too small to fill 256 CUs (8 MB at N = 1e6: 5-12 µs, a third
of what HBM delivers to one long stream) — and to hundreds of scalar nodes that assemble the
gradient vectors (``grad[k] += g_k``: one ``IncSubtensor`` per parameter and term).  Measured before
these passes (tools/bench_wide.py, 40 terms): 0.66 ms per evaluation at N = 1e5 and 0.89 ms at
N = 1e6 for 32 / 320 MB of data — ≈330 launches of ≈2 µs each.

``collect_scalar_updates``        a chain ``inc_subtensor(...inc_subtensor(base[i0], y0)...[ik], yk)`` with
                                  constant scalar indices → ONE ``ScatterScalars`` node (the tail
                                  kernel executes it out of LDS; alone it is the member chain).
``fuse_independent_reductions``   mutually independent ``ElemwiseReduce`` nodes → ONE
                                  ``MultiElemwise`` launch (``blockIdx.y`` = term): a single stream
                                  over all terms' data, per-term partials handed to the tail.

Reference ops: ``IncSubtensor`` (tensor/subtensor.py:1441), ``Elemwise``/``CAReduce``
(tensor/elemwise.py:375, 1233); the fusions change how many launches the nodes cost, not what they
compute (oracle: np_graph ``ScatterScalars`` / ``MultiElemwise`` run the members).
"""

from __future__ import annotations

import numpy as np

from pytensor_amd.inline import _copy, _index
from pytensor_amd.ir import Graph, Node

MIN_CHAIN = 3  # shorter chains are left to their own launches
MIN_TERMS = 3
MAX_TERMS_PER_LAUNCH = 48  # kernel-argument budget (≈4 KB): ~8 eight-byte arguments per term


def _const_int(g: Graph, vid):
    c = g.vars[vid].const
    if c is None:
        return None
    a = np.asarray(c)
    if a.size != 1 or a.dtype.kind not in "iu":
        return None
    return int(a.reshape(()))


def _scalar_index_update(g: Graph, n: Node):
    """``IncSubtensor`` of ONE element of a vector at a constant index → (x, y, index, is_set) or None"""
    if n.op != "IncSubtensor" or len(n.inputs) != 3:
        return None
    idx = n.params["idx_list"]
    if len(idx) != 1 or isinstance(idx[0], slice):
        return None
    x, y, iv = n.inputs
    if g.vars[x].ndim != 1 or g.vars[y].ndim != 0:
        return None
    k = _const_int(g, iv)
    if k is None:
        return None
    return x, y, k, bool(n.params["set_instead_of_inc"])


def collect_scalar_updates(g: Graph) -> Graph:
    producer, consumers = _index(g)
    outs = set(g.outputs)
    used = set()
    chains = []  # (indices of the member nodes in order, base var, [(y, k, is_set)])
    # walk from the LAST update of a chain backwards
    for kn in range(len(g.nodes) - 1, -1, -1):
        if kn in used:
            continue
        u = _scalar_index_update(g, g.nodes[kn])
        if u is None:
            continue
        members, steps = [kn], [u[1:]]
        x = u[0]
        while True:
            kp = producer.get(x)
            if kp is None or kp in used or x in outs or len(consumers.get(x, [])) != 1:
                break
            up = _scalar_index_update(g, g.nodes[kp])
            if up is None:
                break
            members.append(kp)
            steps.append(up[1:])
            x = up[0]
        if len(members) < MIN_CHAIN:
            continue
        members.reverse()
        steps.reverse()
        used.update(members)
        chains.append((members, x, steps))
    if not chains:
        return g
    replace = {}  # index of the last member -> new node ; other members dropped
    drop = set()
    for members, base, steps in chains:
        sub = [g.nodes[k] for k in members]
        params = {"indices": [int(k) for _, k, _ in steps], "set": [bool(s) for _, _, s in steps], "nodes": sub, "base_fill": None}
        first = base
        # a base that is `alloc(constant, n)` read by nothing else (the zeros a gradient accumulates
        # into) is not materialised: the node takes the length and fills on the fly
        kb = producer.get(base)
        if kb is not None and g.nodes[kb].op == "Alloc" and len(g.nodes[kb].inputs) == 2 and consumers.get(base) == [members[0]] and base not in outs:
            A = g.nodes[kb]
            c = g.vars[A.inputs[0]].const
            if c is not None and np.asarray(c).size == 1 and g.vars[base].ndim == 1:
                params["base_fill"] = float(np.asarray(c).reshape(()))
                params["nodes"] = [A] + sub
                first = A.inputs[1]
                drop.add(kb)
        node = Node("ScatterScalars", params, [first] + [y for y, _, _ in steps], list(sub[-1].outputs))
        replace[members[-1]] = node
        drop.update(members[:-1])
    nodes = []
    for k, n in enumerate(g.nodes):
        if k in drop:
            continue
        nodes.append(replace.get(k, n))
    return _copy(g, nodes)


def _fusable_term(g: Graph, n: Node) -> bool:
    if n.op != "ElemwiseReduce":
        return False
    p = n.params
    if p.get("partial_inputs") or p.get("gather") or p.get("defer_reduce"):
        return False
    if any(b["op"] in ("ScalarLoop", "LoopOut") for b in p["scalar"]["body"]):
        return False
    # every output fully reduced (nothing stored): the streaming-sum shape of a likelihood term
    return all(r is not None for r in p["reduce"])


def fuse_independent_reductions(g: Graph) -> Graph:
    cand = [k for k, n in enumerate(g.nodes) if _fusable_term(g, n)]
    if len(cand) < MIN_TERMS:
        return g
    # a candidate that (transitively) depends on another candidate stays out: the members of the
    # group must be mutually independent
    producer, _ = _index(g)
    tainted = {}  # var -> True if it depends on a candidate's output

    def dep(v):
        return tainted.get(v, False)

    members = []
    for k, n in enumerate(g.nodes):
        d = any(dep(i) for i in n.inputs)
        if k in cand and not d:
            members.append(k)
            for o in n.outputs:
                tainted[o] = True
        else:
            for o in n.outputs:
                tainted[o] = d
    if len(members) < MIN_TERMS:
        return g
    mset = set(members)
    # schedule: everything that does not depend on a member first (original order), then the fused
    # launch(es), then the rest (original order) — valid because members are mutually independent
    before, after = [], []
    tainted = {}
    for k, n in enumerate(g.nodes):
        if k in mset:
            for o in n.outputs:
                tainted[o] = True
            continue
        d = any(tainted.get(i, False) for i in n.inputs)
        for o in n.outputs:
            tainted[o] = d
        (after if d else before).append(n)
    fused = []
    for c0 in range(0, len(members), MAX_TERMS_PER_LAUNCH):
        chunk = [g.nodes[k] for k in members[c0 : c0 + MAX_TERMS_PER_LAUNCH]]
        if len(chunk) < 2:
            fused += chunk
            continue
        terms, ins, outs, reduce = [], [], [], []
        for n in chunk:
            terms.append({"scalar": n.params["scalar"], "reduce": n.params["reduce"], "n_inputs": len(n.inputs), "n_outputs": len(n.outputs)})
            ins += list(n.inputs)
            outs += list(n.outputs)
            reduce += list(n.params["reduce"])
        fused.append(Node("MultiElemwise", {"terms": terms, "reduce": reduce, "nodes": chunk}, ins, outs))
    return _copy(g, before + fused + after)
