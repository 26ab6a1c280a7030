"""GEMM ↔ Elemwise boundary fusions for launch-bound graphs (the inner graph of a ``Scan``).

A GRU/LSTM step is a handful of skinny GEMMs (``(B×H)@(H×H)``, B ≪ 128) separated by
elementwise gates.  On MI355X each of those GEMMs needs split-K to occupy the chip, and the
split-K *finish* (sum of the slabs + the Gemm's alpha/beta epilogue) was a launch of its own:
three of the nine dependent launches of a GRU step, ≈6 µs each, doing what the next kernel
could do while it reads its input anyway.

``defer_gemm_finish``   ``Gemm/Dot22 → Elemwise``: the GEMM leaves its raw slabs
                        (``GemmPartials``), the consumer's generated kernel sums them in slab
                        order and applies ``b*y + a*(·)`` inside its scalar graph.

Both rewrites recurse into ``Scan`` inner graphs (reference ops: ``Gemm`` blas/gemm.py:76,
``Dot22`` blas/gemm.py:248, ``Elemwise`` elemwise.py:375).
"""

from __future__ import annotations

import numpy as np

from pytensor_amd.inline import MAX_INPUTS, _copy, _index, _inline_at
from pytensor_amd.ir import Graph, Node, Var

_EW = ("Elemwise", "ElemwiseReduce")


def _map_scan_inner(g: Graph, f) -> Graph:
    """Apply ``f`` to the inner graph of every Scan node (functionally)."""
    nodes, changed = [], False
    for n in g.nodes:
        if n.op == "Scan":
            inner = n.params["inner"]
            new_inner = f(inner)
            if new_inner is not inner:
                params = dict(n.params)
                params["inner"] = new_inner
                n = Node(n.op, params, list(n.inputs), list(n.outputs))
                changed = True
        nodes.append(n)
    return _copy(g, nodes) if changed else g


def _fresh(g: Graph, new_vars: dict, dtype, shape, const=None, name=None) -> int:
    vid = max(max(g.vars), max(new_vars, default=0)) + 1
    new_vars[vid] = Var(vid, dtype, tuple(shape), "tensor", const, name)
    return vid


def defer_gemm_finish(g: Graph) -> Graph:
    g = _map_scan_inner(g, defer_gemm_finish)
    while True:
        producer, consumers = _index(g)
        out_set = set(g.outputs)
        hit = None
        for kg, G in enumerate(g.nodes):
            if G.op not in ("Gemm", "Dot22"):
                continue
            out = G.outputs[0]
            ov = g.vars[out]
            if ov.ndim != 2 or ov.dtype not in ("float32", "float64") or out in out_set:
                continue
            if any(s == 1 for s in ov.shape):
                continue  # could be broadcast inside the consumer
            cons = consumers.get(out, [])
            if len(cons) != 1:
                continue
            kc = cons[0]
            E = g.nodes[kc]
            if E.op not in _EW or E.inputs.count(out) != 1 or g.vars[E.outputs[0]].ndim != 2:
                continue
            if E.op == "ElemwiseReduce" and all(s is not None for s in E.params["reduce"]):
                pass  # fully reduced consumers are fine too
            if G.op == "Gemm":
                y, a, A, B, b = G.inputs
                bv = g.vars[b]
                if bv.const is None or g.vars[a].ndim != 0 or bv.ndim != 0:
                    continue
                if g.vars[y].ndim != 2:
                    continue
            if len(E.inputs) + 3 > MAX_INPUTS:
                continue
            hit = (kg, kc)
            break
        if hit is None:
            return g
        kg, kc = hit
        G, E = g.nodes[kg], g.nodes[kc]
        out = G.outputs[0]
        dt = g.vars[out].dtype
        q = E.inputs.index(out)
        new_vars, pre = {}, []
        pv = _fresh(g, new_vars, dt, (None, None, None), name="gemm_partials")
        if G.op == "Dot22":
            A, B = G.inputs
            extra = [pv]
            pb = {"in_dtypes": [dt], "out_dtypes": [dt], "body": [], "outs": [["i", 0]]}
        else:
            y, a, A, B, b = G.inputs
            a2 = _fresh(g, new_vars, g.vars[a].dtype, (1, 1))
            pre.append(Node("DimShuffle", {"new_order": ["x", "x"]}, [a], [a2]))
            if not np.any(np.asarray(g.vars[b].const)):
                # beta == 0: y is not read (gemm.py:183-216; it may hold anything)
                extra = [pv, a2]
                pb = {
                    "in_dtypes": [dt, g.vars[a].dtype], "out_dtypes": [dt],
                    "body": [{"op": "Mul", "in": [["i", 1], ["i", 0]], "dtype": dt}], "outs": [["t", 0]],
                }
            else:
                b2 = _fresh(g, new_vars, g.vars[b].dtype, (1, 1))
                pre.append(Node("DimShuffle", {"new_order": ["x", "x"]}, [b], [b2]))
                extra = [pv, a2, b2, y]
                pb = {
                    "in_dtypes": [dt, g.vars[a].dtype, g.vars[b].dtype, g.vars[y].dtype], "out_dtypes": [dt],
                    "body": [
                        {"op": "Mul", "in": [["i", 1], ["i", 0]], "dtype": dt},
                        {"op": "Mul", "in": [["i", 2], ["i", 3]], "dtype": dt},
                        {"op": "Add", "in": [["t", 1], ["t", 0]], "dtype": dt},
                    ],
                    "outs": [["t", 2]],
                }
        params = dict(E.params)
        params["scalar"] = _inline_at(E.params["scalar"], q, pb)
        n_kept = len(E.inputs) - 1
        old_pi = [p - (p > q) for p in (E.params.get("partial_inputs") or [])]
        params["partial_inputs"] = old_pi + [n_kept]  # pv is the first appended input
        merged = Node(E.op, params, [i for pos, i in enumerate(E.inputs) if pos != q] + extra, list(E.outputs))
        partials = Node("GemmPartials", {}, [A, B], [pv])
        nodes = []
        for k, n in enumerate(g.nodes):
            if k == kg:
                nodes.append(partials)
            elif k == kc:
                nodes += pre + [merged]
            else:
                nodes.append(n)
        g = _copy(g, nodes)
        g.vars.update(new_vars)


def merge_sibling_gemms(g: Graph) -> Graph:
    """``h @ U_r`` and ``h @ U_z`` of one Scan step → ``h @ [U_r | U_z]``.

    Two ``GemmPartials`` nodes of a Scan's inner graph that share the left operand and whose
    right operands are loop constants (non-sequences) become one product against the
    concatenated weights; the concatenation (``Join`` on axis 1) is hoisted into the outer
    graph — once per evaluation, not per step — and travels as one more non-sequence.  The
    consumers read column blocks of the shared slabs (strided views, no copy).  One launch
    instead of two per step; the merged GEMM is still far too small to notice the wider N."""
    out_nodes, new_vars, changed = [], {}, False
    for n in g.nodes:
        if n.op != "Scan":
            out_nodes.append(n)
            continue
        res = _merge_in_scan(g, n, new_vars)
        if res is None:
            out_nodes.append(n)
        else:
            pre, scan = res
            out_nodes += pre + [scan]
            changed = True
    if not changed:
        return g
    g2 = _copy(g, out_nodes)
    g2.vars.update(new_vars)
    return merge_sibling_gemms(g2)  # further pairs / other Scan nodes


def _merge_in_scan(g: Graph, scan: Node, new_vars: dict):
    info = scan.params["info"]
    inner: Graph = scan.params["inner"]
    nns = info["n_non_seqs"]
    if nns < 2:
        return None
    n_in = len(inner.inputs)
    non_seq_pos = {v: p for p, v in enumerate(inner.inputs) if p >= n_in - nns}
    gp = [(k, m) for k, m in enumerate(inner.nodes) if m.op == "GemmPartials"]
    pair = None
    for a in range(len(gp)):
        for b in range(a + 1, len(gp)):
            (k1, P1), (k2, P2) = gp[a], gp[b]
            if P1.inputs[0] != P2.inputs[0] or P1.inputs[1] == P2.inputs[1]:
                continue
            if P1.inputs[1] not in non_seq_pos or P2.inputs[1] not in non_seq_pos:
                continue
            v1, v2 = inner.vars[P1.inputs[1]], inner.vars[P2.inputs[1]]
            if v1.dtype != v2.dtype or v1.ndim != 2 or v2.ndim != 2:
                continue
            pair = (k1, k2)
            break
        if pair:
            break
    if pair is None:
        return None
    k1, k2 = pair
    P1, P2 = inner.nodes[k1], inner.nodes[k2]
    A, B1, B2 = P1.inputs[0], P1.inputs[1], P2.inputs[1]
    dt = inner.vars[B1].dtype
    # ---- outer graph: Bcat = Join(axis=1)(B1, B2), one more non-sequence of the Scan ----
    o1 = scan.inputs[len(scan.inputs) - n_in + non_seq_pos[B1]]
    o2 = scan.inputs[len(scan.inputs) - n_in + non_seq_pos[B2]]
    bcat_o = _fresh(g, new_vars, dt, (None, None), name="scan_gemm_weights_cat")
    join = Node("Join", {"axis": 1}, [o1, o2], [bcat_o])
    # ---- inner graph ----
    ivars = {}
    bcat_i = _fresh(inner, ivars, dt, (None, None), name="weights_cat")
    pvc = _fresh(inner, ivars, dt, (None, None, None), name="gemm_partials_cat")
    n1 = _fresh(inner, ivars, "int64", ())
    full = slice(None, None, None)
    new_nodes = []
    for k, m in enumerate(inner.nodes):
        if k == k1:
            new_nodes.append(Node("GemmPartials", {}, [A, bcat_i], [pvc]))
            new_nodes.append(Node("Shape_i", {"i": 1}, [B1], [n1]))
            new_nodes.append(Node("Subtensor", {"idx_list": [full, full, slice(None, 0, None)]}, [pvc, n1], [P1.outputs[0]]))
            new_nodes.append(Node("Subtensor", {"idx_list": [full, full, slice(0, None, None)]}, [pvc, n1], [P2.outputs[0]]))
        elif k == k2:
            continue
        else:
            new_nodes.append(m)
    new_inner = _copy(inner, new_nodes)
    new_inner.vars.update(ivars)
    new_inner.inputs = list(inner.inputs) + [bcat_i]
    new_info = dict(info)
    new_info["n_non_seqs"] = nns + 1
    params = dict(scan.params)
    params["inner"] = new_inner
    params["info"] = new_info
    return [join], Node("Scan", params, list(scan.inputs) + [bcat_o], list(scan.outputs))


def fuse_dot_epilogue(g: Graph) -> Graph:
    """``GemmPartials(A, W) → Elemwise`` inside a Scan step → one ``DotEpilogue`` launch.

    A recurrent product ``h @ U`` has M = batch rows (64 in BASELINE config #5): split-K slabs
    through HBM plus a generic N-d gate kernel cost 11 µs + 5-8 µs per pair on MI355X.  When the
    right operand is a loop constant (a non-sequence of the Scan) it is repacked once per
    evaluation (``PackB16``, hoisted into the outer graph like ``merge_sibling_gemms`` hoists
    its ``Join``) and the product, its slab sum and the consuming ``Composite`` become one
    generated kernel (codegen.dot_epilogue_source): a GRU step is two dependent launches.
    Several products feeding one ``Elemwise`` (``rh@U_h`` and ``h@U_z`` of the update gate) are
    accumulated in the same launch.  Shapes the kernel does not cover are decided at run time
    by the handler (dispatch/blas.py: plain GEMM + elementwise kernel).
    Reference ops: ``Dot22``/``Gemm`` blas/gemm.py:76,248; ``Elemwise`` elemwise.py:375;
    loop semantics scan/op.py:1827."""
    out_nodes, new_vars, changed = [], {}, False
    for n in g.nodes:
        res = _dot_epilogue_in_scan(g, n, new_vars) if n.op == "Scan" else None
        if res is None:
            out_nodes.append(n)
        else:
            pre, scan = res
            out_nodes += pre + [scan]
            changed = True
    if not changed:
        return g
    g2 = _copy(g, out_nodes)
    g2.vars.update(new_vars)
    return g2


def _dot_epilogue_in_scan(g: Graph, scan: Node, new_vars: dict):
    info = scan.params["info"]
    inner: Graph = scan.params["inner"]
    nns = info["n_non_seqs"]
    n_in = len(inner.inputs)
    non_seq_pos = {v: p for p, v in enumerate(inner.inputs) if p >= n_in - nns}
    producer, consumers = _index(inner)
    out_set = set(inner.outputs)
    hits = {}  # index of the Elemwise -> [(pos, A, B, index of the GemmPartials)]
    for ke, E in enumerate(inner.nodes):
        pi = E.params.get("partial_inputs") if E.op == "Elemwise" else None
        if not pi or E.params.get("gather") or inner.vars[E.outputs[0]].ndim != 2:
            continue
        dots = []
        for q in sorted(pi):
            v = E.inputs[q]
            kp = producer.get(v)
            P = inner.nodes[kp] if kp is not None else None
            if P is None or P.op != "GemmPartials" or consumers.get(v) != [ke] or v in out_set:
                break
            A, B = P.inputs
            va, vb = inner.vars[A], inner.vars[B]
            if B not in non_seq_pos or va.ndim != 2 or vb.ndim != 2:
                break
            if va.dtype != vb.dtype or va.dtype not in ("float32", "float64"):
                break
            if inner.vars[v].dtype != va.dtype:
                break
            dots.append((q, A, B, kp))
        else:
            hits[ke] = dots
    if not hits:
        return None
    ivars, pre, packed = {}, [], {}  # packed: inner B -> inner packed var
    new_inner_inputs, new_outer_inputs = [], []
    for dots in hits.values():
        for _, _, B, _ in dots:
            if B in packed:
                continue
            dt = inner.vars[B].dtype
            packed[B] = _fresh(inner, ivars, dt, (None,), name="weights_packed16")
            ob = scan.inputs[len(scan.inputs) - n_in + non_seq_pos[B]]
            op_ = _fresh(g, new_vars, dt, (None,), name="scan_weights_packed16")
            pre.append(Node("PackB16", {}, [ob], [op_]))
            new_inner_inputs.append(packed[B])
            new_outer_inputs.append(op_)
    dead = {kp for dots in hits.values() for _, _, _, kp in dots}
    nodes = []
    for k, m in enumerate(inner.nodes):
        if k in dead:
            continue
        if k in hits:
            dots = hits[k]
            ins = list(m.inputs)
            extra = []
            for q, A, B, _ in dots:
                ins[q] = A
                extra += [B, packed[B]]
            params = {"scalar": m.params["scalar"], "dot_inputs": [q for q, *_ in dots]}
            m = Node("DotEpilogue", params, ins + extra, list(m.outputs))
        nodes.append(m)
    nodes = _hoist_shared_left_operand(inner, nodes, ivars)
    new_inner = _copy(inner, nodes)
    new_inner.vars.update(ivars)
    new_inner.inputs = list(inner.inputs) + new_inner_inputs
    new_info = dict(info)
    new_info["n_non_seqs"] = nns + len(new_inner_inputs)
    params = dict(scan.params)
    params["inner"] = new_inner
    params["info"] = new_info
    return pre, Node("Scan", params, list(scan.inputs) + new_outer_inputs, list(scan.outputs))


def _hoist_shared_left_operand(inner: Graph, nodes, ivars):
    """``h @ U_z`` of a later ``DotEpilogue`` moves into the earlier one that already streams ``h``
    (``h @ U_r``): the generated kernel loads the shared left operand once and runs two accumulator
    chains (codegen.dot_epilogue_source ``share``); the moved product leaves as an extra raw output
    and enters its old consumer as an ordinary elementwise operand.  Per GRU step: 64 KB less per
    workgroup, and the second launch — the longer one — loses a whole product.

    The earlier node keeps its original form in ``params["plain"]``: shapes are not known until
    run time, and when the moved product's shape differs from the node's own the handler computes
    it with a plain GEMM and runs the original body (dispatch/dotew.py)."""
    nodes = list(nodes)
    for j, D2 in enumerate(nodes):
        if D2.op != "DotEpilogue" or len(D2.params["dot_inputs"]) < 2:
            continue
        nb2 = len(D2.params["scalar"]["in_dtypes"])
        for i in range(j):
            D1 = nodes[i]
            if D1.op != "DotEpilogue" or len(D1.params["dot_inputs"]) != 1 or "plain" in D1.params:
                continue
            nb1 = len(D1.params["scalar"]["in_dtypes"])
            a1 = D1.inputs[D1.params["dot_inputs"][0]]
            hit = None
            for t, q in enumerate(D2.params["dot_inputs"]):
                if D2.inputs[q] == a1:
                    hit = (t, q)
                    break
            if hit is None or len(D1.inputs) + 3 > MAX_INPUTS:
                continue
            # the operands of the moved product must exist before D1: A does (D1 reads it), its
            # weights are loop constants (graph inputs)
            t, q = hit
            B, Bp = D2.inputs[nb2 + 2 * t], D2.inputs[nb2 + 2 * t + 1]
            dt = D2.params["scalar"]["in_dtypes"][q]
            if dt != D1.params["scalar"]["in_dtypes"][D1.params["dot_inputs"][0]]:
                continue
            raw = _fresh(inner, ivars, dt, (None, None), name="shared_left_product")
            b1 = D1.params["scalar"]
            body1 = {"in_dtypes": list(b1["in_dtypes"]) + [dt], "out_dtypes": list(b1["out_dtypes"]) + [dt],
                     "body": b1["body"], "outs": list(b1["outs"]) + [["i", nb1]]}
            ins1 = list(D1.inputs[:nb1]) + [a1] + list(D1.inputs[nb1:]) + [B, Bp]
            p1 = {"scalar": body1, "dot_inputs": list(D1.params["dot_inputs"]) + [nb1],
                  "plain": {"scalar": b1, "dot_inputs": list(D1.params["dot_inputs"]), "n_inputs": len(D1.inputs), "moved": [nb1, len(b1["out_dtypes"])]}}
            nodes[i] = Node("DotEpilogue", p1, ins1, list(D1.outputs) + [raw])
            # D2: the dot input becomes an elementwise operand; its (B, Bp) pair leaves
            ins2 = list(D2.inputs)
            ins2[q] = raw
            del ins2[nb2 + 2 * t : nb2 + 2 * t + 2]
            dots2 = [d for d in D2.params["dot_inputs"] if d != q]
            nodes[j] = Node("DotEpilogue", {"scalar": D2.params["scalar"], "dot_inputs": dots2}, ins2, list(D2.outputs))
            return _hoist_shared_left_operand(inner, nodes, ivars)  # further pairs
    return nodes
