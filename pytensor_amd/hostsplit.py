"""Shape arithmetic split out of device ``Composite``s.

The reference's ``FusionOptimizer`` (pytensor/tensor/rewriting/elemwise.py:562-1007) fuses every
elementwise ``Apply`` it can reach, including 0-d integer/bool *shape arithmetic* (``Shape_i`` values
compared for the broadcast ``Assert``s of ``pytensor/tensor/extra_ops.py`` / ``raise_op.py:26``) into the
same multi-output ``Composite`` as floating-point work on computed data.  On the C backend that costs
nothing; here it turns a value the host already knows (a shape) into a device result, and the
``CheckAndRaise`` / ``ARange`` / ``Alloc`` that consume it then have to read the device — a stream
synchronisation per node, and a graph that cannot be frozen into a plan (the GP marginal likelihood of
tests/golden: 6 such asserts).

``split_host_shape_arithmetic`` finds the outputs of a multi-output ``Elemwise`` whose scalar cone touches
only host-known integer inputs (``Shape_i`` results, integer constants, and what the handlers keep on the
host: ``ScalarFromTensor`` / ``TensorFromScalar`` / ``CheckAndRaise`` of such, all-host ``Elemwise``) and
moves them into a node of their own, which ``dispatch/elemwise.py`` evaluates on the host (``_host_eval``).
Values are unchanged: the same scalar ops on the same integers.
"""

from __future__ import annotations

import numpy as np

from pytensor_amd.ir import Graph, Node

# ops the host evaluator implements (dispatch/elemwise.py _HOST_OPS)
_HOST_OPS = frozenset(
    "Add Mul Sub Neg IntDiv Mod Abs EQ NEQ LT GT LE GE AND OR Invert Maximum Minimum Switch Cast Identity Sign Sqr".split()
)
_PASS_THROUGH = ("ScalarFromTensor", "TensorFromScalar", "CheckAndRaise", "ViewOp", "DeepCopyOp")


def _is_small_int(var) -> bool:
    return np.dtype(var.dtype).kind in "iub" if var.kind in ("tensor", "scalar") else False


def host_known(g: Graph) -> set:
    """Variables whose value the host holds without reading the device: small integer constants,
    ``Shape_i`` results, and what is derived from those by host-evaluated nodes."""
    known = {vid for vid, v in g.vars.items() if v.const is not None and _is_small_int(v) and np.asarray(v.const).size <= 8}
    for n in g.nodes:
        if n.op == "Shape_i":
            known.update(n.outputs)
        elif n.op in _PASS_THROUGH:
            if n.inputs and n.inputs[0] in known:
                known.update(n.outputs)
        elif n.op == "Elemwise" and not n.params.get("gather"):
            body = n.params["scalar"]
            if all(i in known for i in n.inputs) and _cone_is_host(body, range(len(body["outs"]))):
                known.update(n.outputs)
    return known


def _cone(body, out_idx):
    """(body-op indices, input positions) reached from the outputs ``out_idx``"""
    ops, ins = set(), set()
    stack = [body["outs"][k] for k in out_idx]
    while stack:
        r = stack.pop()
        if r[0] == "i":
            ins.add(r[1])
        elif r[0] == "t" and r[1] not in ops:
            ops.add(r[1])
            stack.extend(body["body"][r[1]]["in"])
    return ops, ins


def _cone_is_host(body, out_idx) -> bool:
    ops, ins = _cone(body, out_idx)
    if any(np.dtype(body["in_dtypes"][i]).kind not in "iub" for i in ins):
        return False
    if any(np.dtype(body["out_dtypes"][k]).kind not in "iub" for k in out_idx):
        return False
    for t in ops:
        b = body["body"][t]
        if b["op"] not in _HOST_OPS or np.dtype(b["dtype"]).kind not in "iub" or "body" in b:
            return False
        if any(r[0] == "c" and np.dtype(r[2]).kind not in "iub" for r in b["in"]):
            return False
    return True


def _restrict(body, out_idx, node_inputs):
    """The scalar graph of the outputs ``out_idx`` alone: (body, the node inputs it keeps)."""
    ops, ins = _cone(body, out_idx)
    ins = sorted(ins)
    imap = {old: new for new, old in enumerate(ins)}
    order = sorted(ops)
    tmap = {old: new for new, old in enumerate(order)}

    def ref(r):
        if r[0] == "i":
            return ["i", imap[r[1]]]
        if r[0] == "t":
            return ["t", tmap[r[1]]]
        return list(r)

    new = {
        "in_dtypes": [body["in_dtypes"][i] for i in ins],
        "out_dtypes": [body["out_dtypes"][k] for k in out_idx],
        "body": [{**body["body"][t], "in": [ref(r) for r in body["body"][t]["in"]]} for t in order],
        "outs": [ref(body["outs"][k]) for k in out_idx],
    }
    return new, [node_inputs[i] for i in ins]


def split_host_shape_arithmetic(g: Graph) -> Graph:
    known = host_known(g)
    nodes, changed = [], False
    for n in g.nodes:
        if n.op != "Elemwise" or n.params.get("gather") or len(n.outputs) < 2:
            nodes.append(n)
            continue
        body = n.params["scalar"]
        # 0-d results only: shape arithmetic (a broadcast of a host scalar against a device operand stays put)
        host_out = [
            k for k in range(len(n.outputs))
            if g.vars[n.outputs[k]].ndim == 0
            and all(n.inputs[i] in known for i in _cone(body, [k])[1])
            and _cone(body, [k])[1]
            and _cone_is_host(body, [k])
        ]
        if not host_out or len(host_out) == len(n.outputs):
            nodes.append(n)
            continue
        dev_out = [k for k in range(len(n.outputs)) if k not in host_out]
        hb, hin = _restrict(body, host_out, n.inputs)
        db, din = _restrict(body, dev_out, n.inputs)
        if any(g.vars[i].ndim != 0 for i in hin):
            nodes.append(n)
            continue
        rest = {k: v for k, v in n.params.items() if k != "scalar"}
        nodes.append(Node("Elemwise", {**rest, "scalar": hb}, hin, [n.outputs[k] for k in host_out]))
        nodes.append(Node("Elemwise", {**rest, "scalar": db}, din, [n.outputs[k] for k in dev_out]))
        changed = True
    if not changed:
        return g
    out = Graph(name=g.name)
    out.vars = dict(g.vars)
    out.inputs, out.outputs = list(g.inputs), list(g.outputs)
    out.nodes = nodes
    return out


def device_reads_for_control(g: Graph) -> list:
    """``(node index, op, input position)`` of every assert condition / shape operand that is NOT
    host-known — each one is a device read (a stream synchronisation) in the eager executor and a
    reason a plan cannot be frozen.  Used by the tests and by ``tools/bench_gp.py``."""
    known = host_known(g)
    res = []
    for k, n in enumerate(g.nodes):
        if n.op == "CheckAndRaise":
            res += [(k, n.op, p) for p, i in enumerate(n.inputs) if p >= 1 and i not in known]
        elif n.op in ("ARange", "Alloc", "AllocEmpty", "Eye"):
            first = 1 if n.op == "Alloc" else 0
            res += [(k, n.op, p) for p, i in enumerate(n.inputs) if p >= first and i not in known]
    return res
