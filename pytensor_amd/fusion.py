"""IR-level fusion of ``Elemwise`` outputs into the full reductions that consume them.

The reference fuses ``Elemwise → Sum`` only for single-input Elemwise and only in
the C backend (``local_careduce_fusion``, pytensor/tensor/rewriting/elemwise.py:1098-1160,
tagged ``cxx_only`` — excluded for this linker).  On MI355X the pattern is the
difference between reading each N-vector once and writing + re-reading an
N-sized intermediate, so we generalise it: any number of inputs, any number of
outputs, each output independently either stored or fully reduced
(``ElemwiseReduce`` node).  An output is reduced in-kernel when its *only*
consumer is a ``CAReduce`` over all axes with Add/Mul/Maximum/Minimum and it is not
a graph output.
"""

from __future__ import annotations

from pytensor_amd.ir import Graph, Node

_FUSABLE = {"Add", "Mul", "Maximum", "Minimum", "ScalarMaximum", "ScalarMinimum"}
_CANON = {"ScalarMaximum": "Maximum", "ScalarMinimum": "Minimum"}


def fuse_elemwise_reduce(g: Graph) -> Graph:
    consumers = {}
    for k, n in enumerate(g.nodes):
        for i in n.inputs:
            consumers.setdefault(i, []).append(k)
    out_set = set(g.outputs)
    producer = {}
    for k, n in enumerate(g.nodes):
        for o in n.outputs:
            producer[o] = k

    absorbed = set()  # CAReduce node indices folded into their producer
    new_nodes = {}
    for k, n in enumerate(g.nodes):
        if n.op != "Elemwise":
            continue
        spec = []
        outs = list(n.outputs)
        any_fused = False
        for pos, o in enumerate(n.outputs):
            cons = consumers.get(o, [])
            fused = None
            if o not in out_set and len(cons) == 1:
                c = g.nodes[cons[0]]
                ov = g.vars[o]
                if (
                    c.op == "CAReduce"
                    and c.params["scalar_op"] in _FUSABLE
                    and sorted(c.params["axis"]) == list(range(ov.ndim))
                    and ov.ndim > 0
                    and ov.dtype in ("float64", "float32", "int64", "int32")
                ):
                    fused = {
                        "op": _CANON.get(c.params["scalar_op"], c.params["scalar_op"]),
                        "acc_dtype": c.params["acc_dtype"],
                        "dtype": c.params["dtype"],
                    }
                    absorbed.add(cons[0])
                    outs[pos] = c.outputs[0]
                    any_fused = True
            spec.append(fused)
        if any_fused:
            new_nodes[k] = Node("ElemwiseReduce", {"scalar": n.params["scalar"], "reduce": spec}, list(n.inputs), outs)

    if not new_nodes:
        return g
    out = Graph(name=g.name)
    out.vars = g.vars
    out.inputs = list(g.inputs)
    out.outputs = list(g.outputs)
    for k, n in enumerate(g.nodes):
        if k in absorbed:
            continue
        out.nodes.append(new_nodes.get(k, n))
    return out
