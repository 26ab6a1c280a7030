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
