"""``Elemwise`` → ``CAReduce`` over SOME axes in one kernel (``ElemwiseAxisReduce``).

The reference fuses ``Elemwise → CAReduce`` only for a single-input ``Elemwise`` and only in the C backend
(``local_careduce_fusion``, pytensor/tensor/rewriting/elemwise.py:1098-1160, tagged ``cxx_only`` and so excluded
for this linker); ``fusion.fuse_elemwise_reduce`` restates it for FULL reductions.  Row / column reductions of
a fused expression — ``sum_j exp(x_ij - m_i)`` of a logsumexp, column sums of squares — otherwise write an
array the size of the input and read it straight back.  Here:

``duplicate_cheap_producers``   a cheap ``Elemwise`` with ONE full-size input (``X - m`` with ``m`` a broadcast
    row or column) read by several fusable clients is cloned per client: each client then re-reads ``X``
    (the same bytes as reading the materialised value) and the write disappears.  The stabilised logsumexp
    the reference's rewrites produce (tests/benchmarks/test_logsumexp.py:9-13 → Max, ``X - m`` feeding a Max
    and an Exp/Sum) goes from 7 passes over ``X`` to 3;
``fuse_elemwise_axis_reduce``   an ``Elemwise`` all of whose outputs are reduced over the same axis tuple by
    ``CAReduce`` nodes becomes one ``ElemwiseAxisReduce`` node (kernel: codegen_tile.tile_reduce_source).
"""

from __future__ import annotations

from pytensor_amd.inline import MAX_INPUTS, _copy, _index, _scalar_like, dead_code_elimination
from pytensor_amd.ir import Graph, Node, Var

_FUSABLE = {"Add", "Mul", "Maximum", "Minimum", "ScalarMaximum", "ScalarMinimum"}
_CANON = {"ScalarMaximum": "Maximum", "ScalarMinimum": "Minimum"}
_DTYPES = ("float64", "float32", "int64", "int32")
_CHEAP = {"Add", "Sub", "Mul", "Neg", "Sqr", "Abs", "Identity", "Cast", "Switch", "Maximum", "Minimum", "ScalarMaximum", "ScalarMinimum",
          "LT", "GT", "LE", "GE", "EQ", "NEQ", "IsNan", "IsInf", "Sign", "Clip", "TrueDiv"}
MAX_CHEAP_BODY = 6


def _axis_reducible(g, c, v):
    """``c`` is a ``CAReduce`` this linker can fold behind the Elemwise producing ``v``"""
    if c.op != "CAReduce" or c.params["scalar_op"] not in _FUSABLE:
        return False
    axes = sorted(set(int(a) for a in c.params["axis"]))
    nd = g.vars[v].ndim
    return bool(axes) and nd > 0 and g.vars[v].dtype in _DTYPES and all(0 <= a < nd for a in axes)


def _big_inputs(g, node, nd):
    """inputs that are not broadcast along any dimension (unknown extents count as full size)"""
    return [u for u in node.inputs if g.vars[u].ndim == nd and not any(s == 1 for s in g.vars[u].shape) and nd > 0]


def duplicate_cheap_producers(g: Graph) -> Graph:
    changed = True
    while changed:
        changed = False
        producer, consumers = _index(g)
        out_set = set(g.outputs)
        for kp, P in enumerate(g.nodes):
            if P.op != "Elemwise" or len(P.outputs) != 1:
                continue
            v = P.outputs[0]
            cons = consumers.get(v, [])
            if v in out_set or len(set(cons)) < 2 or _scalar_like(g.vars[v]):
                continue
            body = P.params["scalar"]
            if len(body["body"]) > MAX_CHEAP_BODY or any(b["op"] not in _CHEAP for b in body["body"]):
                continue
            nd = g.vars[v].ndim
            if any(g.vars[u].ndim != nd for u in P.inputs):
                continue
            nbig = len(_big_inputs(g, P, nd))
            # a SMALL value (statically broadcast along some dimension: the switch(isinf(max), 0, max) row of a
            # logsumexp) read by several full-size loops: recomputed in each of them instead of a launch of its own
            small = nbig == 0 and any(s_ == 1 for s_ in g.vars[v].shape)
            if nbig != 1 and not small:
                continue
            ok = True
            n_red = 1 if small else 0
            for kc in set(cons):
                c = g.nodes[kc]
                if c.op in ("Elemwise", "ElemwiseReduce"):
                    ok = ok and g.vars[c.outputs[0]].ndim in (nd, 0) and len(c.inputs) - 1 + len(P.inputs) <= MAX_INPUTS and not c.params.get("gather") \
                        and all(g.vars[i].ndim == nd for i in c.inputs)
                elif _axis_reducible(g, c, v) and not small:
                    n_red += 1
                else:
                    ok = False
            if not ok or n_red == 0:
                continue  # (without a reduction among the clients the plain vertical fusion rules already decide)
            # one clone per client beyond the first
            nodes = list(g.nodes)
            new_vars = {}
            clones = []
            for kc in sorted(set(cons))[1:]:
                vid = max(max(g.vars), max(new_vars, default=0)) + 1
                vv = g.vars[v]
                new_vars[vid] = Var(vid, vv.dtype, tuple(vv.shape), vv.kind, None, None)
                clones.append(Node("Elemwise", P.params, list(P.inputs), [vid]))
                c = nodes[kc]
                nodes[kc] = Node(c.op, c.params, [vid if i == v else i for i in c.inputs], list(c.outputs))
            nodes = nodes[: kp + 1] + clones + nodes[kp + 1 :]
            g = _copy(g, nodes)
            g.vars.update(new_vars)
            changed = True
            break
    return g


def fuse_elemwise_axis_reduce(g: Graph) -> Graph:
    producer, consumers = _index(g)
    out_set = set(g.outputs)
    absorbed, new_nodes = set(), {}
    for k, n in enumerate(g.nodes):
        if n.op != "Elemwise" or n.params.get("gather") or n.params.get("partial_inputs"):
            continue
        groups = {}  # axis tuple -> [(output position, CAReduce index)]
        for pos, o in enumerate(n.outputs):
            cons = consumers.get(o, [])
            if o in out_set or len(cons) != 1 or not _axis_reducible(g, g.nodes[cons[0]], o):
                groups = None
                break
            a = tuple(sorted(set(int(x) for x in g.nodes[cons[0]].params["axis"])))
            groups.setdefault(a, []).append((pos, cons[0]))
        nd = g.vars[n.outputs[0]].ndim
        if not groups or any(len(a) == nd for a in groups) or any(g.vars[i].ndim != nd for i in n.inputs):
            continue  # (a reduction over every axis is fusion.fuse_elemwise_reduce's)
        # one fused node per axis tuple: outputs reduced over different axes re-read the inputs (the bytes the
        # unfused reductions would read back) and the stores disappear
        body = n.params["scalar"]
        made = []
        for a, members in groups.items():
            sub = {"in_dtypes": list(body["in_dtypes"]), "out_dtypes": [body["out_dtypes"][pos] for pos, _ in members], "body": body["body"],
                   "outs": [body["outs"][pos] for pos, _ in members]}
            specs = []
            for pos, kc in members:
                c = g.nodes[kc]
                specs.append({"op": _CANON.get(c.params["scalar_op"], c.params["scalar_op"]), "acc_dtype": c.params["acc_dtype"], "dtype": c.params["dtype"]})
            made.append((Node("ElemwiseAxisReduce", {"scalar": sub, "axis": list(a), "reduce": specs}, list(n.inputs), [g.nodes[kc].outputs[0] for _, kc in members]),
                         max(kc for _, kc in members)))
            absorbed.update(kc for _, kc in members)
        new_nodes[k] = made
    if not new_nodes:
        return g
    # a fused node takes the place of its LAST absorbed reduction (every input is defined by then)
    place = {}
    for made in new_nodes.values():
        for node, at in made:
            place[at] = node
    nodes = []
    for k, n in enumerate(g.nodes):
        if k in new_nodes:
            continue
        if k in absorbed:
            if k in place:
                nodes.append(place[k])
            continue
        nodes.append(n)
    return dead_code_elimination(_copy(g, nodes))
