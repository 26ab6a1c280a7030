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
                         min(kc for _, kc in members)))
            absorbed.update(kc for _, kc in members)
        new_nodes[k] = made
    if not new_nodes:
        return g
    # The fused nodes take the place of the ELEMWISE itself: their only inputs are the Elemwise's inputs (all defined
    # before position k) and every reader of an absorbed reduction's output sits after that reduction, hence after k.
    # (Round 5 placed them at the LAST absorbed reduction: a reader of an EARLIER reduction's output — the DimShuffle
    # of the first of two keepdims sums — then ran before its producer.)
    nodes = []
    for k, n in enumerate(g.nodes):
        if k in new_nodes:
            nodes.extend(node for node, _ in new_nodes[k])
        elif k not in absorbed:
            nodes.append(n)
    return dead_code_elimination(_copy(g, nodes))


# ---------------------------------------------------------------------------------------------
# log(sum(exp(f))) over some axes -> ONE reduction (the running (max, scaled sum) pair)
# ---------------------------------------------------------------------------------------------


def _canon(body, ref, in_vars):
    """a scalar sub-expression as a hashable tree over GRAPH variables (two nodes' bodies can then be compared)"""
    if ref[0] == "i":
        return ("i", in_vars[ref[1]])
    if ref[0] == "t":
        n = body["body"][ref[1]]
        if n["op"] in ("ScalarLoop", "LoopOut"):
            return ("opaque", id(n))
        return (n["op"], n["dtype"], tuple(_canon(body, r, in_vars) for r in n["in"]))
    return ("c", repr(ref[1]), ref[2])


def _tree_inputs(tree, acc):
    if tree[0] == "i":
        acc.add(tree[1])
    elif tree[0] not in ("c", "opaque"):
        for ch in tree[2]:
            _tree_inputs(ch, acc)
    return acc


def _prune(node: Node) -> Node:
    """drop scalar ops no output needs, then inputs nothing reads"""
    b = node.params["scalar"]
    live, stack = set(), [r for r in b["outs"]]
    while stack:
        r = stack.pop()
        if r[0] == "t" and r[1] not in live:
            live.add(r[1])
            stack.extend(b["body"][r[1]]["in"])
    tmap, nb = {}, []
    for k, n in enumerate(b["body"]):
        if k in live:
            tmap[k] = len(nb)
            nb.append(n)
    used = sorted({r[1] for n in nb for r in n["in"] if r[0] == "i"} | {r[1] for r in b["outs"] if r[0] == "i"})
    imap = {p: q for q, p in enumerate(used)}

    def ref(r):
        return ["t", tmap[r[1]]] if r[0] == "t" else (["i", imap[r[1]]] if r[0] == "i" else list(r))

    body = {"in_dtypes": [b["in_dtypes"][p] for p in used], "out_dtypes": list(b["out_dtypes"]),
            "body": [{**n, "in": [ref(r) for r in n["in"]]} for n in nb], "outs": [ref(r) for r in b["outs"]]}
    params = dict(node.params)
    params["scalar"] = body
    return Node(node.op, params, [node.inputs[p] for p in used], list(node.outputs))


def fuse_logsumexp(g: Graph) -> Graph:
    """``log(sum_axes(exp(f(x))))`` — an ``ElemwiseAxisReduce`` whose one output is a Sum of an ``Exp`` and whose sum
    is read by nothing but a ``Log`` — becomes ONE reduction with the op ``LogSumExp`` (csrc/reduce_device.h OpLse:
    the running pair (max, scaled sum), one exp per element, cannot overflow); the consumer's ``Log`` becomes the
    identity.  Then the identity  LSE_axes(f - c) + c = LSE_axes(f)  for any ``c`` constant along the reduced axes
    removes the shifts that stabilised the two-pass form: the ``- max`` inside the reduction and the ``+ max``
    outside cancel, and the Max reductions that fed them die with their last reader.  The stabilised graph the
    reference's rewrites make of tests/benchmarks/test_logsumexp.py:9-13 (Max, Max of the shifted values, Sum of Exp)
    — three passes over X — becomes one."""
    changed = True
    while changed:
        changed = False
        producer, consumers = _index(g)
        out_set = set(g.outputs)
        for kr, R in enumerate(g.nodes):
            if R.op != "ElemwiseAxisReduce" or len(R.outputs) != 1 or R.params["reduce"][0]["op"] != "Add":
                continue
            rb = R.params["scalar"]
            o = rb["outs"][0]
            if o[0] != "t" or rb["body"][o[1]]["op"] != "Exp" or rb["out_dtypes"][0] not in ("float64", "float32"):
                continue
            if R.params["reduce"][0]["dtype"] != rb["out_dtypes"][0]:
                continue
            # the sum -> (views) -> ONE Elemwise, which reads it through exactly one Log
            v, chain = R.outputs[0], []
            while True:
                cons = consumers.get(v, [])
                if v in out_set or len(cons) != 1:
                    v = None
                    break
                c = g.nodes[cons[0]]
                if c.op == "DimShuffle":
                    chain.append(cons[0])
                    v = c.outputs[0]
                    continue
                break
            if v is None:
                continue
            ke = consumers[v][0]
            E = g.nodes[ke]
            if E.op != "Elemwise" or E.inputs.count(v) != 1:
                continue
            eb = E.params["scalar"]
            q = E.inputs.index(v)
            uses = [(k, n) for k, n in enumerate(eb["body"]) if ["i", q] in [list(r) for r in n["in"]]]
            if len(uses) != 1 or uses[0][1]["op"] != "Log" or any(list(r) == ["i", q] for r in eb["outs"]):
                continue
            kl = uses[0][0]
            axes = [int(a) for a in R.params["axis"]]
            # ---- LSE_axes(f - c1 - c2 ...) + c1 + c2 ... : peel the shifts off f
            pre = rb["body"][o[1]]["in"][0]
            small = {vid for vid in R.inputs if all(g.vars[vid].shape[a] == 1 for a in axes)}
            shifts, cur = [], pre
            while cur[0] == "t" and rb["body"][cur[1]]["op"] == "Sub":
                lhs, rhs = rb["body"][cur[1]]["in"]
                tree = _canon(rb, rhs, R.inputs)
                if not _tree_inputs(tree, set()) <= small or "opaque" in repr(tree):
                    break
                shifts.append(tree)
                cur = lhs
            # the Add (if any) that takes the Log in the consumer
            adds = [(k, n) for k, n in enumerate(eb["body"]) if n["op"] == "Add" and ["t", kl] in [list(r) for r in n["in"]]]
            new_pre, new_add = pre, None
            log_reads = sum(1 for n in eb["body"] for r in n["in"] if list(r) == ["t", kl])
            if shifts and len(adds) == 1 and log_reads == 1 and not any(list(r) == ["t", kl] for r in eb["outs"]):
                ka, A = adds[0]
                ops_left = [list(r) for r in A["in"]]
                trees = [None if r == ["t", kl] else _canon(eb, r, E.inputs) for r in ops_left]
                matched, cur2 = 0, pre
                # shifts were peeled outermost first: cancel them in that order while the consumer adds them back
                for tree in shifts:
                    hit = next((j for j, t in enumerate(trees) if t is not None and t == tree), None)
                    if hit is None:
                        break
                    trees.pop(hit)
                    ops_left.pop(hit)
                    cur2 = rb["body"][cur2[1]]["in"][0]
                    matched += 1
                if matched:
                    new_pre, new_add = cur2, (ka, ops_left)
            # ---- rewrite R and E
            nrb = {**rb, "outs": [list(new_pre)]}
            spec = [{**R.params["reduce"][0], "op": "LogSumExp"}]
            R2 = _prune(Node("ElemwiseAxisReduce", {**R.params, "scalar": nrb, "reduce": spec}, list(R.inputs), list(R.outputs)))
            nbody = [dict(n) for n in eb["body"]]
            nbody[kl] = {**nbody[kl], "op": "Identity"}
            if new_add is not None:
                ka, ops_left = new_add
                nbody[ka] = {**nbody[ka], "op": "Add", "in": ops_left} if len(ops_left) > 1 else {**nbody[ka], "op": "Identity", "in": ops_left}
            E2 = _prune(Node("Elemwise", {**E.params, "scalar": {**eb, "body": nbody}}, list(E.inputs), list(E.outputs)))
            nodes = list(g.nodes)
            nodes[kr], nodes[ke] = R2, E2
            g = dead_code_elimination(_copy(g, nodes))
            changed = True
            break
    return g


def drop_identity_elemwise(g: Graph) -> Graph:
    """An ``Elemwise`` whose scalar graph is nothing but ``Identity`` of its one input — what is left of
    ``log(sum(exp(x - m))) + m`` once ``fuse_logsumexp`` has folded the logarithm and both shifts into the reduction — is
    a copy launch (3 us behind a 30 us log-sum-exp).  Its readers read its operand instead.  Kept when the operand is a
    graph input or a constant (an output must not alias what the caller owns: link/vm.py no-recycling / aliasing.py
    semantics, tests/test_gpu_e2e.py::test_outputs_are_fresh_and_do_not_alias_inputs), when dtype or static shape
    differ (a cast / a broadcast), or when the operand is itself a graph output (two outputs would share a buffer)."""
    producer, _ = _index(g)
    inputs = set(g.inputs)
    outs = set(g.outputs)
    rename = {}
    keep = []
    for n in g.nodes:
        if n.op == "Elemwise" and len(n.inputs) == 1 and len(n.outputs) == 1 and not n.params.get("gather") and not n.params.get("partial_inputs"):
            b = n.params["scalar"]
            src = rename.get(n.inputs[0], n.inputs[0])
            vi, vo = g.vars[src], g.vars[n.outputs[0]]
            chain, ok = b["outs"][0], bool(b["body"])
            while ok and chain[0] == "t":
                node = b["body"][chain[1]]
                ok = node["op"] == "Identity" and node["dtype"] == vi.dtype
                chain = node["in"][0] if ok else chain
            ok = ok and list(chain) == ["i", 0] and len(b["body"]) <= 4
            if (ok and vi.dtype == vo.dtype and tuple(vi.shape) == tuple(vo.shape) and src not in inputs and vi.const is None and src in producer
                    and src not in outs and vi.kind == "tensor"):
                rename[n.outputs[0]] = src
                continue
        keep.append(Node(n.op, n.params, [rename.get(i, i) for i in n.inputs], list(n.outputs)) if any(i in rename for i in n.inputs) else n)
    if not rename:
        return g
    out = _copy(g, keep)
    out.outputs = [rename.get(o, o) for o in g.outputs]
    return out
