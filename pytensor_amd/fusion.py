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


LATENCY_OPS = {"Cholesky", "SolveTriangular", "CholeskySolve", "Blockwise", "CholeskyTrsv"}
_VIEW_OPS = {"DimShuffle", "Subtensor", "Shape_i", "ScalarFromTensor"}
# Experiment kept behind PTHIP_WIDE_STREAM=1: a graph with a many-term launch (MultiElemwise: north_star's 48 likelihood
# terms, ~230 us of fp64 VALU issue, little memory traffic) next to the one-pass matrix kernel (gchain: ~170 us of HBM
# streaming, little arithmetic) looks like the pair to overlap — the many-term launch alone on the second stream, the
# short latency chain in line with the streaming kernel.  Measured (profiles/r5s_wide_overlap.md): they do overlap and
# it does not pay — gchain holds 230 VGPRs (two workgroups per CU); with the other launch's workgroups in the slots it
# runs 335 us instead of 170 and the evaluation takes 0.52 ms instead of 0.49.  Default: the latency chain keeps the
# second stream.
_WIDE_STREAM = __import__("os").environ.get("PTHIP_WIDE_STREAM", "0") == "1"
# PTHIP_WIDE_STREAM=2: the many-term launch BEHIND the latency chain on the second stream (Cholesky -> solve -> many-term),
# so that only the streaming kernel's last quarter shares the chip with it.
_WIDE_BEHIND = __import__("os").environ.get("PTHIP_WIDE_STREAM", "0") == "2"


def stream_classes(g: Graph, staged_inputs=()):
    """0 = HBM-streaming work, 1 = the single-CU dense-linear-algebra chain and the
    scalar work hanging off it (see ``plan.StreamScheduler``)."""
    cls = []
    var_cls = {}
    wide = (_WIDE_STREAM or _WIDE_BEHIND) and any(n.op == "MultiElemwise" for n in g.nodes) and any(n.op == "GemvChain" for n in g.nodes)
    side_ops = (LATENCY_OPS | {"MultiElemwise"} if _WIDE_BEHIND else {"MultiElemwise"}) if wide else LATENCY_OPS
    for n in g.nodes:
        parents = [var_cls[v] for v in n.inputs if v in var_cls]
        if wide and n.op in _VIEW_OPS and not parents:
            # a view of a graph input launches nothing and orders nothing: it takes the side of whoever reads it
            cls.append(1)
            continue  # (its outputs stay out of var_cls: readers see no parent)
        c = 1 if (n.op in side_ops or (parents and all(p == 1 for p in parents))) else 0
        cls.append(c)
        for o in n.outputs:
            var_cls[o] = c
    return cls


def schedule_latency_chain_first(g: Graph) -> Graph:
    """Topological re-ordering that issues the latency-bound chain (class 1) as early as
    its inputs allow, so that in a multi-stream plan it is enqueued — and starts — before
    the long HBM-streaming kernels it overlaps with.  Pure re-ordering: same nodes."""
    cls = stream_classes(g)
    if not any(cls):
        return g
    produced_by = {}
    for k, n in enumerate(g.nodes):
        for o in n.outputs:
            produced_by[o] = k
    deps = [set(produced_by[v] for v in n.inputs if v in produced_by) for n in g.nodes]
    done, order = set(), []
    remaining = list(range(len(g.nodes)))
    while remaining:
        ready = [k for k in remaining if deps[k] <= done]
        pick = next((k for k in ready if cls[k] == 1), ready[0])
        order.append(pick)
        done.add(pick)
        remaining.remove(pick)
    out = Graph(name=g.name)
    out.vars, out.inputs, out.outputs = g.vars, list(g.inputs), list(g.outputs)
    out.nodes = [g.nodes[k] for k in order]
    return out


def segment_graph(g: Graph):
    """Split the node list into three consecutive segments for a multi-stream plan:

    * ``A`` — the latency chain: class-1 nodes with no class-0 ancestor
      (Cholesky → solves → the scalar work on their results);
    * ``B`` — class-0 nodes with no class-1 ancestor (the HBM-streaming work);
    * ``C`` — the rest (whatever combines the two).

    A and B are mutually independent, so a frozen plan captures them as two hipGraphs and
    launches them on two streams; C follows both.  Returns ``(reordered graph, seg)``
    with ``seg[k] ∈ {0: A, 1: B, 2: C}``; ``seg`` is ``None`` when there is nothing to
    overlap (ROCm 7.2 serialises the branches of a *single* captured graph, so the
    overlap has to come from separate graphs on separate streams).
    """
    cls = stream_classes(g)
    if not any(cls):
        return g, None
    wide = (_WIDE_STREAM or _WIDE_BEHIND) and any(n.op == "MultiElemwise" for n in g.nodes) and any(n.op == "GemvChain" for n in g.nodes)
    produced_by = {}
    for k, n in enumerate(g.nodes):
        for o in n.outputs:
            produced_by[o] = k
    anc0 = [False] * len(g.nodes)  # has a class-0 ancestor
    anc1 = [False] * len(g.nodes)  # has a class-1 ancestor
    for k, n in enumerate(g.nodes):
        for v in n.inputs:
            p = produced_by.get(v)
            if p is None:
                continue
            if wide and g.nodes[p].op in _VIEW_OPS and not any(u in produced_by for u in g.nodes[p].inputs):
                continue  # (a view of a graph input: no ancestry on either side)
            anc0[k] = anc0[k] or anc0[p] or cls[p] == 0
            anc1[k] = anc1[k] or anc1[p] or cls[p] == 1
    seg = []
    for k in range(len(g.nodes)):
        if cls[k] == 1 and not anc0[k]:
            seg.append(0)
        elif cls[k] == 0 and not anc1[k]:
            seg.append(1)
        else:
            seg.append(2)
    if 0 not in seg or 1 not in seg:
        return g, None
    order = [k for k in range(len(g.nodes)) if seg[k] == 0]
    order += [k for k in range(len(g.nodes)) if seg[k] == 1]
    order += [k for k in range(len(g.nodes)) if seg[k] == 2]
    out = Graph(name=g.name)
    out.vars, out.inputs, out.outputs = g.vars, list(g.inputs), list(g.outputs)
    out.nodes = [g.nodes[k] for k in order]
    return out, [seg[k] for k in order]


# ---------------------------------------------------------------------------
# Gemv(row) -> Elemwise -> Gemv(col) over the SAME matrix: read it once
# ---------------------------------------------------------------------------


def fuse_gemv_chain(g: Graph) -> Graph:
    """``r = b1*y1 + a1*A@x1 ; (…, w, …) = Elemwise(…, r, …) ; out = b2*y2 + a2*A.T@w``

    is the forward/backward pair of every regression-style logp+grad graph
    (``X@beta`` forward, ``X.T@residual`` backward).  The
    reference runs it as three nodes that stream the N×K matrix twice
    (pytensor/tensor/blas/gemv.py:64-108 twice + elemwise.py:755); on MI355X the matrix
    is the dominant HBM traffic, so the three are rewritten into

    * ``GemvChain``  (placed where the Elemwise was): one kernel that loads a block of
      rows once, forms the dot products, applies the scalar graph to them and
      accumulates ``A.T@w`` from the same registers → per-workgroup partials;
    * ``GemvFinish`` (placed where the second Gemv was): fixed-order sum of the
      partials with the ``alpha2/beta2/y2`` epilogue.

    Two neighbours are absorbed as well when present (the reference's
    numba-only ``FusedElemwise`` idea, tensor/rewriting/fused_elemwise.py):

    * a gather feeding the scalar graph, ``table[idx]`` (``AdvancedSubtensor`` on axis 0 of
      a vector, sole consumer) is read inside the kernel (input mode ``G``);
    * the scatter-add of one of its vector outputs, ``inc_subtensor(base[idx], o)``
      (``AdvancedIncSubtensor``, the gradient of such a gather), is accumulated in the
      same pass into per-workgroup bin partials and finished by a second ``GemvFinish``
      (``out = base + Σ partials``) where the scatter node was.
    """
    from pytensor_amd.ir import Var

    producer = {}
    for k, n in enumerate(g.nodes):
        for o in n.outputs:
            producer[o] = k
    consumers = {}
    for k, n in enumerate(g.nodes):
        for i in n.inputs:
            consumers.setdefault(i, []).append(k)
    out_set = set(g.outputs)
    replace = {}  # node index -> new Node or None (removed)
    used = set()
    new_vars = {}
    next_id = [max(g.vars) + 1]

    def fresh(dtype, shape, const=None, name=None):
        vid = next_id[0]
        next_id[0] += 1
        new_vars[vid] = Var(vid, dtype, shape, "tensor", const, name)
        return vid

    for k2, n2 in enumerate(g.nodes):
        if n2.op != "Gemv":
            continue
        y2, a2, At, w, b2 = n2.inputs
        kt = producer.get(At)
        if kt is None or g.nodes[kt].op != "DimShuffle" or g.nodes[kt].params["new_order"] != [1, 0]:
            continue
        A = g.nodes[kt].inputs[0]
        ke = producer.get(w)
        if ke is None or ke in used or g.nodes[ke].op not in ("Elemwise", "ElemwiseReduce"):
            continue
        ne = g.nodes[ke]
        # float64, or (round 5) float32 throughout: the kernel converts on load and keeps its double-precision core,
        # partial slabs stay float64, GemvFinish casts back (dispatch/fused.py)
        fdt = g.vars[w].dtype
        if g.vars[w].ndim != 1 or fdt not in ("float64", "float32") or g.vars[A].dtype != fdt:
            continue
        # find the forward Gemv feeding the elementwise node
        r_pos = None
        for pos, i in enumerate(ne.inputs):
            k1 = producer.get(i)
            if k1 is not None and g.nodes[k1].op == "Gemv" and g.nodes[k1].inputs[2] == A and k1 not in used:
                r_pos = pos
                break
        if r_pos is None:
            continue
        r = ne.inputs[r_pos]
        k1 = producer[r]
        n1 = g.nodes[k1]
        if ne.inputs.count(r) != 1 or g.vars[r].dtype != fdt or g.vars[n1.inputs[3]].dtype != fdt:
            continue
        # the chain / finish kernels take alpha and beta by value: they must be constants
        if any(g.vars[v].const is None for v in (n1.inputs[1], n1.inputs[4], a2, b2)):
            continue
        # every other operand of the scalar graph must be available before the chain starts: a
        # value that itself depends on r (e.g. the 2/N·r of a *mean* squared error, whose scale
        # comes after a full reduction of r) would make the fused node its own ancestor
        if any(_depends_on(g, producer, v, k1) for pos, v in enumerate(ne.inputs) if pos != r_pos):
            continue
        store_r = (r in out_set) or len(consumers.get(r, [])) != 1
        spec = ne.params.get("reduce") or [None] * len(ne.outputs)
        w_out = ne.outputs.index(w)
        if spec[w_out] is not None:
            continue
        # ---- gathers feeding the scalar graph --------------------------------------
        gather = []  # positions (in the elementwise input numbering) read as table[idx]
        e_ins = []
        removed = []
        for pos, v in enumerate(ne.inputs):
            if pos == r_pos:
                continue
            kp = producer.get(v)
            P = g.nodes[kp] if kp is not None else None
            if (
                P is not None
                and P.op == "AdvancedSubtensor"
                and P.params["idx_list"] == [0]
                and len(P.inputs) == 2
                and g.vars[P.inputs[0]].ndim == 1
                and g.vars[P.inputs[1]].ndim == 1
                and g.vars[P.inputs[1]].dtype == "int64"
                and consumers.get(v, []) == [ke]
                and v not in out_set
                and kp not in used
            ):
                gather.append(pos)
                e_ins += [P.inputs[0], P.inputs[1]]
                removed.append(kp)
            else:
                e_ins.append(v)
        # ---- scatter-add of a stored vector output -----------------------------------
        scatter = None
        for pos_o, o in enumerate(ne.outputs):
            if spec[pos_o] is not None or g.vars[o].ndim != 1 or g.vars[o].dtype != fdt:
                continue
            for ks in consumers.get(o, []):
                S = g.nodes[ks]
                if (
                    S.op == "AdvancedIncSubtensor"
                    and ks not in used
                    and S.params["idx_list"] == [0]
                    and not S.params["set_instead_of_inc"]
                    and not S.params.get("ignore_duplicates")
                    and len(S.inputs) == 3
                    and S.inputs[1] == o
                    and g.vars[S.inputs[0]].ndim == 1
                    and g.vars[S.inputs[0]].dtype == fdt
                    and g.vars[S.inputs[2]].dtype == "int64"
                    and producer.get(S.inputs[2], -1) < ke
                    and not any(_depends_on(g, producer, u, kk) for u in (S.inputs[0], S.inputs[2]) for kk in (k1, ke))
                ):
                    scatter = (pos_o, ks, S)
                    break
            if scatter:
                break
        # which vector outputs must still be written to HBM
        # (a consumer is "fused" only through the operand the chain kernel takes over: `w` as the
        #  vector of the second Gemv, the scattered values of the scatter-add.  The same node may
        #  read an output through another operand — the second Gemv's `y`, reference test
        #  tests/scan/test_basic.py::TestScan::test_inner_grad — which GemvFinish still loads)
        def _needs_store(o):
            if o in out_set:
                return True
            for c in consumers.get(o, []):
                if c == k2:
                    if any(v == o and pos != 3 for pos, v in enumerate(n2.inputs)):
                        return True
                elif scatter and c == scatter[1]:
                    if any(v == o and pos != 1 for pos, v in enumerate(scatter[2].inputs)):
                        return True
                else:
                    return True
            return False

        out_store = [spec[pos_o] is None and _needs_store(o) for pos_o, o in enumerate(ne.outputs)]
        part = fresh("float64", (None, None), name="gemv_chain_partials")
        chain_inputs = list(n1.inputs) + e_ins
        chain_outputs = ([r] if store_r else []) + list(ne.outputs) + [part]
        params = {
            "scalar": ne.params["scalar"], "reduce": spec, "r_pos": r_pos, "w_out": w_out,
            "store_r": store_r, "gather": gather, "out_store": out_store, "scatter_out": None,
        }
        if scatter:
            pos_o, ks, S = scatter
            partS = fresh("float64", (None, None), name="gemv_chain_scatter_partials")
            import numpy as _np

            chain_inputs.append(S.inputs[2])  # the scatter index vector
            chain_outputs.append(partS)
            params["scatter_out"] = pos_o
            one = fresh("float64", (), const=_np.asarray(1.0), name="one")
            base = S.inputs[0]
            kb = producer.get(base)
            B = g.nodes[kb] if kb is not None else None
            zero_base = (
                B is not None
                and B.op == "Alloc"
                and len(B.inputs) == 2
                and g.vars[B.inputs[0]].const is not None
                and not _np.any(_np.asarray(g.vars[B.inputs[0]].const))
            )
            if zero_base:
                # inc_subtensor(zeros(n)[idx], o): only n is needed — the zero fill is never
                # launched (the length scalar doubles as the unread `y` of the finish node)
                chain_inputs.append(B.inputs[1])
                params["scatter_len_input"] = True
                zero = fresh("float64", (), const=_np.asarray(0.0), name="zero")
                replace[ks] = Node("GemvFinish", {}, [partS, B.inputs[1], one, zero], list(S.outputs))
            else:
                chain_inputs.append(base)  # base: only its length is used by the chain
                replace[ks] = Node("GemvFinish", {}, [partS, base, one, one], list(S.outputs))
            used.add(ks)
        replace[k1] = None
        for kp in removed:
            replace[kp] = None
            used.add(kp)
        replace[ke] = Node("GemvChain", params, chain_inputs, chain_outputs)
        replace[k2] = Node("GemvFinish", {}, [part, y2, a2, b2], list(n2.outputs))
        used.update((k1, ke, k2))
    if not replace:
        return g
    out = Graph(name=g.name)
    out.vars = dict(g.vars)
    out.vars.update(new_vars)
    out.inputs = list(g.inputs)
    out.outputs = list(g.outputs)
    for k, n in enumerate(g.nodes):
        if k in replace:
            if replace[k] is not None:
                out.nodes.append(replace[k])
        else:
            out.nodes.append(n)
    # the fused node may now depend on values produced after the elementwise node's old
    # position (the base of the absorbed scatter): restore a valid order
    out.nodes = _stable_toposort(out.nodes)
    return out


def _depends_on(g, producer, v, k_anc, _memo=None):
    """True when variable ``v`` is computed (transitively) from an output of node ``k_anc``."""
    memo = {} if _memo is None else _memo
    stack = [v]
    seen = set()
    while stack:
        u = stack.pop()
        if u in seen:
            continue
        seen.add(u)
        k = producer.get(u)
        if k is None:
            continue
        if k == k_anc:
            return True
        stack.extend(g.nodes[k].inputs)
    return False


def _stable_toposort(nodes):
    """Kahn's algorithm that keeps the given order wherever dependencies allow."""
    produced_by = {}
    for k, n in enumerate(nodes):
        for o in n.outputs:
            produced_by[o] = k
    deps = [set(produced_by[v] for v in n.inputs if v in produced_by) - {k} for k, n in enumerate(nodes)]
    done, order = set(), []
    pending = list(range(len(nodes)))
    while pending:
        for k in pending:
            if deps[k] <= done:
                order.append(k)
                done.add(k)
                pending.remove(k)
                break
        else:  # pragma: no cover
            raise RuntimeError("cycle in lowered graph")
    return [nodes[k] for k in order]


# ---------------------------------------------------------------------------
# Scan: hoist sequence-only matrix products out of the loop
# ---------------------------------------------------------------------------


def hoist_scan_seq_dots(g: Graph) -> Graph:
    """``Dot22(x_t, W)`` inside a ``Scan`` step, with ``x_t`` a slice of a sequence and ``W``
    a non-sequence, does not depend on the recurrence: the T products
    ``(B×K)@(K×N)`` become ONE ``(T·B×K)@(K×N)`` MFMA GEMM before the loop
    (``SeqDot22``), whose result is fed back as an extra sequence.  The reference has the
    same idea as a graph rewrite (``scan_pushout_seq_operation``,
    pytensor/scan/rewriting/__init__.py) but it does not fire once ``BlasOpt`` has turned
    the inner ``dot + add`` into ``Dot22``/``Gemm``; on MI355X it is the difference
    between T skinny launches that cannot fill 256 CUs and one full-rate GEMM.
    """
    import copy

    from pytensor_amd.ir import Var

    if not any(n.op == "Scan" for n in g.nodes):
        return g
    out = Graph(name=g.name)
    out.vars = dict(g.vars)
    out.inputs, out.outputs = list(g.inputs), list(g.outputs)
    next_id = max(out.vars) + 1
    changed = False
    for n in g.nodes:
        if n.op != "Scan":
            out.nodes.append(n)
            continue
        info = dict(n.params["info"])
        inner: Graph = n.params["inner"]
        if info["mit_mot_in_slices"] or info["as_while"]:
            out.nodes.append(n)
            continue
        n_seqs = info["n_seqs"]
        n_inner_in = len(inner.inputs)
        n_non = info["n_non_seqs"]
        seq_in = inner.inputs[:n_seqs]
        non_in = inner.inputs[n_inner_in - n_non :]
        outer_seqs = n.inputs[1 : 1 + n_seqs]
        outer_non = n.inputs[len(n.inputs) - n_non :]
        hoist = []
        for k, m in enumerate(inner.nodes):
            if m.op == "Dot22" and m.inputs[0] in seq_in and m.inputs[1] in non_in and m.outputs[0] not in inner.outputs:
                hoist.append(k)
        if not hoist:
            out.nodes.append(n)
            continue
        new_inner = Graph(name=inner.name)
        new_inner.vars = dict(inner.vars)
        new_inner.outputs = list(inner.outputs)
        new_outer_seqs, new_inner_seqs = [], []
        pre_nodes = []
        for k in hoist:
            m = inner.nodes[k]
            s_pos = seq_in.index(m.inputs[0])
            w_pos = non_in.index(m.inputs[1])
            ov = inner.vars[m.outputs[0]]
            # outer result (T, B, N) and the SeqDot22 node producing it
            vid = next_id
            next_id += 1
            out.vars[vid] = Var(vid, ov.dtype, (None, *ov.shape), "tensor", None, "scan_hoisted_dot")
            # ("lazy": the Scan is the only consumer, as a sequence — its step driver may have the
            #  product computed chunk by chunk on a second stream while the loop runs, dispatch/scan.py)
            pre_nodes.append(Node("SeqDot22", {"lazy": True}, [outer_seqs[s_pos], outer_non[w_pos]], [vid]))
            new_outer_seqs.append(vid)
            new_inner_seqs.append(m.outputs[0])  # the inner var now arrives as a sequence slice
        new_inner.nodes = [m for k, m in enumerate(inner.nodes) if k not in hoist]
        new_inner.inputs = list(seq_in) + new_inner_seqs + list(inner.inputs[n_seqs:])
        info["n_seqs"] = n_seqs + len(hoist)
        new_inputs = [n.inputs[0]] + list(outer_seqs) + new_outer_seqs + list(n.inputs[1 + n_seqs :])
        out.nodes.extend(pre_nodes)
        out.nodes.append(Node("Scan", {"info": info, "inner": new_inner}, new_inputs, list(n.outputs)))
        changed = True
    return out if changed else g


# ---------------------------------------------------------------------------
# Cholesky followed by its first triangular solve: keep the factor in LDS
# ---------------------------------------------------------------------------


def fuse_cholesky_solve(g: Graph) -> Graph:
    """``L = cholesky(S); x = solve_triangular(L, b, lower=True)`` (the whitening step of a
    multivariate-normal logp) → one ``CholeskyTrsv`` node: the 128 KiB factor is still in
    the CU's LDS when the substitution runs, so the second launch and its re-staging of
    the matrix disappear (reference: two LAPACK calls, cholesky.py:52-83 + triangular.py:41-71).
    ``L`` is still written out for its other consumers."""
    producer = {}
    for k, n in enumerate(g.nodes):
        for o in n.outputs:
            producer[o] = k
    replace = {}
    for ks, ns in enumerate(g.nodes):
        if ns.op != "SolveTriangular":
            continue
        p = ns.params
        if not p["lower"] or p["unit_diagonal"] or p["b_ndim"] != 1:
            continue
        kc = producer.get(ns.inputs[0])
        if kc is None or kc in replace or g.nodes[kc].op != "Cholesky" or not g.nodes[kc].params["lower"]:
            continue
        nc = g.nodes[kc]
        if g.vars[nc.inputs[0]].ndim != 2 or g.vars[ns.inputs[1]].ndim != 1:
            continue
        replace[kc] = Node("CholeskyTrsv", {}, [nc.inputs[0], ns.inputs[1]], [nc.outputs[0], ns.outputs[0]])
        replace[ks] = None
    if not replace:
        return g
    out = Graph(name=g.name)
    out.vars, out.inputs, out.outputs = g.vars, list(g.inputs), list(g.outputs)
    nodes = []
    for k, n in enumerate(g.nodes):
        if k in replace:
            if replace[k] is not None:
                nodes.append(replace[k])
        else:
            nodes.append(n)
    out.nodes = _stable_toposort(nodes)
    return out
