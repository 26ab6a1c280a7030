"""Tail fusion: the chain of tiny dependent launches at the end of an evaluation → two launches.

What is left after the streaming kernels of a logp+grad graph (SURVEY Appendix B census: the
``Sum`` second stages, two ``GemvFinish`` of per-workgroup slabs, the 11-input scalar
``Composite`` that combines the log-density terms, the G-vector ``Elemwise``) is microseconds of
arithmetic behind ≈4.5 µs of dependent-launch latency *each* — 6 launches and a hipGraph
boundary, 49 µs of a 254 µs evaluation (profiles/r1q_c4_timeline.md).  The reference has no
analogue (its CVM walks thunks at ≈1 µs per node, link/c/c_code/lazylinker_c.c:749); on a GPU
the fix is structural: every node at the end of the graph whose operands are small goes into ONE
``Tail`` node, executed as

* ``pthip_multi_finish`` (csrc/tail.hip): all large partial slabs shrunk to ≤16 rows, one launch;
* one generated single-workgroup kernel (``codegen.tail_chain_source``): sums those rows and
  the deferred per-workgroup partials of earlier reductions, applies the Gemv epilogues and runs
  the small ``Elemwise`` / ``ElemwiseReduce`` nodes in order with intermediates in LDS, then
  writes the results — inside a frozen plan straight into the pinned output block.

Sizes are not known until run time (``TensorType`` shapes are ``None``): the pass groups by
*structure*, the handler (dispatch/tail.py) checks the actual extents and falls back to the
member nodes' own handlers when something is not small.
"""

from __future__ import annotations

from pytensor_amd.ir import Graph, Node

# ops the tail kernel can execute (DimShuffle: only as a view of a scalar / vector)
_MEMBER_OPS = frozenset(["Elemwise", "ElemwiseReduce", "GemvFinish", "DimShuffle", "ScatterScalars"])
# producers whose reduced outputs can be handed over unfinished (DeferredReduce)
_DEFER_OPS = frozenset(["ElemwiseReduce", "GemvChain", "MultiElemwise"])
MIN_LAUNCHES = 3  # below this the two launches of the tail save nothing


def _member_ok(n: Node) -> bool:
    if n.op not in _MEMBER_OPS:
        return False
    if n.op in ("Elemwise", "ElemwiseReduce"):
        p = n.params
        if p.get("partial_inputs") or p.get("gather") or "scalar" not in p:
            return False
        if any(b["op"] in ("ScalarLoop", "LoopOut") for b in p["scalar"]["body"]):
            return False
    return True


def fuse_tail(g: Graph) -> Graph:
    consumers = {}
    for k, n in enumerate(g.nodes):
        for i in n.inputs:
            consumers.setdefault(i, set()).add(k)
    outs = set(g.outputs)
    member = set()
    # reverse pass: a node belongs to the tail when every consumer of every output does
    for k in range(len(g.nodes) - 1, -1, -1):
        n = g.nodes[k]
        if _member_ok(n) and all(consumers.get(o, set()) <= member for o in n.outputs):
            member.add(k)
    # views alone launch nothing; count what the group would have launched
    launches = sum(1 for k in member if g.nodes[k].op != "DimShuffle")
    if launches < MIN_LAUNCHES:
        return g
    order = sorted(member)
    produced = {o for k in order for o in g.nodes[k].outputs}
    ext_in = []
    for k in order:
        for i in g.nodes[k].inputs:
            if i not in produced and i not in ext_in:
                ext_in.append(i)
    ext_out = [o for k in order for o in g.nodes[k].outputs if o in outs or (consumers.get(o, set()) - member)]
    # reduced outputs of earlier fused kernels consumed only by the tail: leave them unfinished
    defer = {}
    producer = {o: (k, pos) for k, n in enumerate(g.nodes) for pos, o in enumerate(n.outputs)}
    for v in ext_in:
        if v in outs or v not in producer:
            continue
        k, pos = producer[v]
        n = g.nodes[k]
        if n.op not in _DEFER_OPS or not (consumers.get(v, set()) <= member):
            continue
        spec = n.params.get("reduce") or []
        # (GemvChain lists [r] first when it stores r; its reduce spec indexes the scalar outputs)
        rpos = pos - (1 if (n.op == "GemvChain" and n.params.get("store_r")) else 0)
        if 0 <= rpos < len(spec) and spec[rpos] is not None:
            defer.setdefault(k, []).append(rpos)
    new = Graph(name=g.name)
    new.vars, new.inputs, new.outputs = g.vars, list(g.inputs), list(g.outputs)
    first = order[0]
    tail = Node("Tail", {"nodes": [g.nodes[k] for k in order]}, ext_in, ext_out)
    for k, n in enumerate(g.nodes):
        if k in member:
            continue
        if k in defer:
            p = dict(n.params)
            p["defer_reduce"] = sorted(defer[k])
            n = Node(n.op, p, list(n.inputs), list(n.outputs))
        new.nodes.append(n)
    # every producer of an external input precedes the first member only if the member set is a
    # suffix in dependency order — append at the end (all members' consumers are members or outputs)
    new.nodes.append(tail)
    return new
