"""CPU: the IR passes (fusion, hoisting, segmentation, donation) preserve semantics.

The oracle restates every fused node from its unfused parts (oracle/np_graph.py), so
``oracle(pass(graph)) == oracle(graph)`` checks the *rewrites* without a GPU.
"""
import numpy as np
import pytest

import np_graph
from pytensor_amd.passes import run_pipeline as _pipeline
from util import assert_parity, golden_cases, load_case


@pytest.mark.parametrize("name", golden_cases())
def test_passes_preserve_results(name):
    g, ins, cvm, py, meta = load_case(name)
    g2, seg = _pipeline(g)
    out = np_graph.run_graph(g2, ins)
    for k, (a, b) in enumerate(zip(out, cvm)):
        assert_parity(a, b, None, f"{name} out{k} after IR passes", case=name, k=k, py=py[k], slack=2.0)


def test_c4_gets_the_one_pass_gemv_chain_and_two_segments():
    g, *_ = load_case("c4_hier")
    g2, seg = _pipeline(g)
    tail = [n for n in g2.nodes if n.op == "Tail"]
    assert len(tail) == 1 and g2.nodes[-1] is tail[0], "the small nodes behind the streaming kernel form ONE Tail node"
    members = [m.op for m in tail[0].params["nodes"]]
    ops = [n.op for n in g2.nodes] + members
    assert ops.count("GemvChain") == 1 and ops.count("GemvFinish") == 2 and "Gemv" not in ops
    # both second stages of the chain's slabs, the G-vector node, the log-diag sum and the scalar
    # combine are members: 6 dependent launches + a graph boundary became 2 launches (tailfuse.py)
    assert members.count("GemvFinish") == 2 and members.count("ElemwiseReduce") == 2 and members.count("Elemwise") == 1
    chain = next(n for n in g2.nodes if n.op == "GemvChain")
    assert chain.params.get("defer_reduce") == [0, 1], "the chain's two sums are finished inside the tail kernel"
    # the gather a[gidx] and the scatter-add of its gradient are absorbed into the chain
    assert "AdvancedSubtensor" not in ops and "AdvancedIncSubtensor" not in ops
    assert ops.count("ElemwiseReduce") >= 3
    # exp(log_sigma), mu + sigma*z and the zero fill no longer launch anything before the chain;
    # the CAReduce of the scatter result rides on the node that reads it anyway
    # (the two bool Elemwise nodes are shape checks evaluated on the host)
    b_kernels = [
        n.op for n, s in zip(g2.nodes, seg)
        if s == 1 and n.op in ("Elemwise", "ElemwiseReduce", "Alloc", "CAReduce", "GemvChain", "GemvFinish")
        and g2.vars[n.outputs[0]].dtype != "bool"
    ]
    assert b_kernels == ["GemvChain"]
    # segment A = Cholesky/solve chain first, B = streaming, C = combine
    assert seg is not None and seg[0] == 0 and set(seg) == {0, 1, 2} and seg == sorted(seg)
    a_ops = {n.op for n, s in zip(g2.nodes, seg) if s == 0}
    b_ops = {n.op for n, s in zip(g2.nodes, seg) if s == 1}
    assert {"CholeskyTrsv", "SolveTriangular"} <= a_ops and "GemvChain" in b_ops and "Cholesky" not in ops
    # nothing in A or B consumes a value produced in the other
    prod = {}
    for n, s in zip(g2.nodes, seg):
        for o in n.outputs:
            prod[o] = s
    for n, s in zip(g2.nodes, seg):
        if s in (0, 1):
            assert all(prod.get(v, s) == s for v in n.inputs)


def test_c5_scan_dots_are_hoisted():
    g, *_ = load_case("c5_gru")
    g2, _ = _pipeline(g)
    ops = [n.op for n in g2.nodes]
    assert ops.count("SeqDot22") == 3
    scan = next(n for n in g2.nodes if n.op == "Scan")
    inner = scan.params["inner"]
    inner_ops = [m.op for m in inner.nodes]
    # the three recurrent products stay in the loop; each runs inside the generated kernel of the
    # gate that consumes it (gemmfuse.fuse_dot_epilogue): two launches per step, the weights
    # repacked once per evaluation outside the loop
    assert not {"Dot22", "Gemm", "GemmPartials", "Elemwise"} & set(inner_ops)
    de = [m for m in inner.nodes if m.op == "DotEpilogue"]
    assert sorted(len(m.params["dot_inputs"]) for m in de) == [1, 2]
    assert ops.count("PackB16") == 3 and max(k for k, o in enumerate(ops) if o == "PackB16") < ops.index("Scan")
    assert scan.params["info"]["n_non_seqs"] == 12 and len(inner.inputs) == 17
    for m in de:
        nb = len(m.params["scalar"]["in_dtypes"])
        # every packed operand is a non-sequence of the step function
        assert all(v in inner.inputs[-12:] for v in m.inputs[nb:])
    assert scan.params["info"]["n_seqs"] == 4
    # the original graph object is untouched (passes are functional)
    g_again, *_ = load_case("c5_gru")
    assert [n.op for n in g.nodes] == [n.op for n in g_again.nodes]


def test_c5_split_k_form_without_dot_epilogue(monkeypatch):
    """``PTHIP_DOT_EPILOGUE=0``: the round-1 form (split-K slabs folded into the gate kernels,
    sibling products merged) — still what runs when the right operand is not a loop constant."""
    monkeypatch.setenv("PTHIP_DOT_EPILOGUE", "0")
    g, *_ = load_case("c5_gru")
    g2, _ = _pipeline(g)
    ops = [n.op for n in g2.nodes]
    scan = next(n for n in g2.nodes if n.op == "Scan")
    inner_ops = [m.op for m in scan.params["inner"].nodes]
    assert "Dot22" not in inner_ops and "Gemm" not in inner_ops and inner_ops.count("GemmPartials") == 2
    assert ops.count("Join") == 1 and ops.index("Join") < ops.index("Scan")
    assert scan.params["info"]["n_non_seqs"] == 10
    ew = [m for m in scan.params["inner"].nodes if m.op == "Elemwise"]
    assert sorted(len(m.params["partial_inputs"]) for m in ew) == [1, 2]


def test_donations_only_fresh_single_consumer_values():
    from pytensor_amd.executor import HipExecutable

    g, *_ = load_case("c5_gru")
    exe = HipExecutable(g)
    consumers = {}
    for n in exe.graph.nodes:
        for v in n.inputs:
            consumers[v] = consumers.get(v, 0) + 1
    seen = 0
    for n, don in zip(exe.graph.nodes, exe._donations):
        for pos in don:
            v = n.inputs[pos]
            assert consumers[v] == 1 and v not in exe.graph.outputs and v not in exe.graph.inputs
            seen += 1
    assert seen > 0  # the trace buffer (AllocEmpty -> IncSubtensor -> Scan) is donated


def test_inline_passes_units():
    """push-gather, producer inlining, sibling reductions and DCE on a hand-built graph."""
    from pytensor_amd.inline import (
        dead_code_elimination,
        inline_elemwise_producers,
        merge_sibling_reductions,
        push_gather_through_elemwise,
    )
    from pytensor_amd.fusion import fuse_elemwise_reduce
    from pytensor_amd.ir import Graph

    def body(ops, nin, outs=None):
        return {"in_dtypes": ["float64"] * nin, "out_dtypes": ["float64"] * len(outs or [0]), "body": ops,
                "outs": outs or [["t", len(ops) - 1]]}

    g = Graph(name="unit")
    s = g.new_var("float64", (), name="s")
    z = g.new_var("float64", (None,), name="z")
    idx = g.new_var("int64", (None,), name="idx")
    y = g.new_var("float64", (None,), name="y")
    g.inputs = [s, z, idx, y]
    es = g.new_var("float64", ())
    g.add_node("Elemwise", {"scalar": body([{"op": "Exp", "in": [["i", 0]], "dtype": "float64"}], 1)}, [s], [es])
    es1 = g.new_var("float64", (1,))
    g.add_node("DimShuffle", {"new_order": ["x"]}, [es], [es1])
    t = g.new_var("float64", (None,))
    g.add_node("Elemwise", {"scalar": body([{"op": "Mul", "in": [["i", 0], ["i", 1]], "dtype": "float64"}], 2)}, [es1, z], [t])
    tg = g.new_var("float64", (None,))
    g.add_node("AdvancedSubtensor", {"idx_list": [0]}, [t, idx], [tg])
    d = g.new_var("float64", (None,))
    g.add_node("Elemwise", {"scalar": body([{"op": "Sub", "in": [["i", 0], ["i", 1]], "dtype": "float64"}], 2)}, [y, tg], [d])
    tot = g.new_var("float64", ())
    g.add_node("CAReduce", {"scalar_op": "Add", "axis": [0], "acc_dtype": "float64", "dtype": "float64"}, [y], [tot])
    unused = g.new_var("float64", (None,))
    g.add_node("Elemwise", {"scalar": body([{"op": "Neg", "in": [["i", 0]], "dtype": "float64"}], 1)}, [y], [unused])
    g.outputs = [d, tot, es]

    rng = np.random.default_rng(0)
    ins = [np.asarray(0.3), rng.normal(size=7), rng.integers(0, 7, size=20), rng.normal(size=20)]
    want = np_graph.run_graph(g, ins)

    g1 = dead_code_elimination(g)
    assert len(g1.nodes) == len(g.nodes) - 1
    g2 = push_gather_through_elemwise(g1)
    ops = [n.op for n in g2.nodes]
    assert ops.index("AdvancedSubtensor") < ops.index("Elemwise", ops.index("AdvancedSubtensor"))
    gather = next(n for n in g2.nodes if n.op == "AdvancedSubtensor")
    assert gather.inputs == [z, idx]
    g3 = inline_elemwise_producers(g2)
    ew = [n for n in g3.nodes if n.op == "Elemwise"]
    # exp(s) stays (it is a graph output) but is also recomputed inside the one vector kernel
    assert len(ew) == 2 and sorted(len(n.params["scalar"]["body"]) for n in ew) == [1, 3]
    g4 = merge_sibling_reductions(fuse_elemwise_reduce(g3))
    assert "CAReduce" not in [n.op for n in g4.nodes]
    for gg in (g1, g2, g3, g4):
        got = np_graph.run_graph(gg, ins)
        for a, b in zip(got, want):
            np.testing.assert_allclose(a, b, rtol=1e-14)


def test_dot_epilogue_source_compiles_for_both_dtypes():
    """The generated product+epilogue kernel builds for gfx950 without a GPU (hiprtc), with the
    operand loads of both register buffers issued ahead of the first MFMA."""
    from pytensor_amd import codegen, ffi

    for dt, K in (("float32", 1024), ("float64", 48), ("float32", 16)):
        body = {
            "in_dtypes": [dt, dt, dt], "out_dtypes": [dt],
            "body": [{"op": "Add", "in": [["i", 0], ["i", 1]], "dtype": dt}, {"op": "Tanh", "in": [["t", 0]], "dtype": dt},
                     {"op": "Mul", "in": [["t", 1], ["i", 2]], "dtype": dt}],
            "outs": [["t", 2]],
        }
        src = codegen.dot_epilogue_source("dotew_probe", body, [1], K, byvalue=(2,))
        assert src.count("= __builtin_amdgcn_mfma") == (K // 16 + 3) // 4 * 4  # per wave: K/64 groups x 4
        assert len(ffi.jit_compile(src, "dotew_probe.hip")) > 1000


def test_pack_b16_oracle_layout():
    import np_graph
    from pytensor_amd.ir import Graph

    g = Graph(name="pack")
    W = g.new_var("float64", (None, None))
    Wp = g.new_var("float64", (None,))
    g.add_node("PackB16", {}, [W], [Wp])
    g.inputs, g.outputs = [W], [Wp]
    K, N = 21, 35
    w = np.arange(K * N, dtype="float64").reshape(K, N) + 1
    (p,) = np_graph.run_graph(g, [w])
    Kp, Np = 32, 48
    assert p.shape == (Kp * Np,)
    p4 = p.reshape(Np // 16, Kp // 4, 16, 4)
    for ct, k4, j, q in ((0, 0, 0, 0), (2, 5, 2, 0), (1, 3, 15, 3), (2, 5, 3, 0), (0, 7, 0, 3)):
        k, c = 4 * k4 + q, 16 * ct + j
        assert p4[ct, k4, j, q] == (w[k, c] if k < K and c < N else 0.0)


def test_wide_graph_becomes_two_launches():
    """widefuse.py on the 12-term model (golden ``wide_terms``): ONE ``MultiElemwise`` for the
    per-term Elemwise+Sum kernels (all reductions handed on unfinished), everything after it —
    the scalar gradient algebra and both ``IncSubtensor`` chains (as ``ScatterScalars`` nodes, their
    zeros bases absorbed) — inside ONE ``Tail`` node; nothing else launches."""
    g, ins, cvm, py, meta = load_case("wide_terms")
    g2, _ = _pipeline(g)
    ops = [n.op for n in g2.nodes]
    assert ops.count("MultiElemwise") == 1 and ops.count("Tail") == 1
    assert not {"ElemwiseReduce", "IncSubtensor", "Alloc", "CAReduce"} & set(ops)
    launching = [o for o in ops if o not in ("DimShuffle", "Subtensor", "Shape_i", "MultiElemwise", "Tail")]
    assert [o for o in launching if o != "Elemwise"] == []  # (host shape arithmetic only)
    multi = next(n for n in g2.nodes if n.op == "MultiElemwise")
    assert len(multi.params["terms"]) == 12
    assert sorted(multi.params["defer_reduce"]) == list(range(len(multi.outputs)))
    tail = next(n for n in g2.nodes if n.op == "Tail")
    members = [m.op for m in tail.params["nodes"]]
    assert members.count("ScatterScalars") == 2
    for m in tail.params["nodes"]:
        if m.op == "ScatterScalars":
            assert m.params["base_fill"] == 0.0 and sorted(m.params["indices"]) == list(range(12)) and not any(m.params["set"])
    # with the passes switched off the graph keeps one kernel per term (and the same results:
    # test_passes_preserve_results runs both forms through the oracle)
    import os

    os.environ["PTHIP_WIDE"] = "0"
    try:
        g3, _ = _pipeline(g)
    finally:
        del os.environ["PTHIP_WIDE"]
    assert [n.op for n in g3.nodes].count("ElemwiseReduce") == 12


def test_multi_flat_source_shares_bodies_and_compiles():
    from pytensor_amd import codegen, ffi

    dt = "float64"
    b1 = {"in_dtypes": [dt, dt], "out_dtypes": [dt], "body": [{"op": "Sub", "in": [["i", 1], ["i", 0]], "dtype": dt}, {"op": "Sqr", "in": [["t", 0]], "dtype": dt}], "outs": [["t", 1]]}
    b2 = {"in_dtypes": [dt, dt], "out_dtypes": [dt], "body": [{"op": "Sub", "in": [["i", 1], ["i", 0]], "dtype": dt}, {"op": "Abs", "in": [["t", 0]], "dtype": dt}], "outs": [["t", 1]]}
    mk = lambda b, m: {"body": b, "modes": m, "vec": 2, "rs": [("Add", dt)], "unroll": 2}
    src = codegen.multi_flat_source("multi_probe", [mk(b1, "SV"), mk(b2, "CV"), mk(b1, "SV"), mk(b1, "CV")])
    assert src.count("static __device__ __forceinline__ void mt_") == 3  # (b1,SV) shared by two terms
    assert "switch ((blockIdx.x + blockIdx.y) % gridDim.x)" in src and src.count("case ") == 4
    assert len(ffi.jit_compile(src, "multi_probe.hip")) > 1000


def test_multi_flat_source_self_finishing_form_compiles():
    """`finish=True`: one block pointer per term (pair arrays + finished values at the offsets of multi_finish_layout), one
    shared ticket base and status word; the argument list shrinks (48 terms stay under the 4 KB kernel-argument block)."""
    from pytensor_amd import codegen, ffi

    dt = "float64"
    b1 = {"in_dtypes": [dt, dt], "out_dtypes": [dt, dt], "outs": [["t", 1], ["t", 0]],
          "body": [{"op": "Sub", "in": [["i", 1], ["i", 0]], "dtype": dt}, {"op": "Sqr", "in": [["t", 0]], "dtype": dt}]}
    mk = lambda g: {"body": b1, "modes": "SV", "vec": 2, "rs": [("Add", dt), ("Add", dt)], "unroll": 2, "groups": g}
    plain = codegen.multi_flat_source("multi_probe_p", [mk(7), mk(64)])
    src = codegen.multi_flat_source("multi_probe_f", [mk(7), mk(64)], finish=True)
    assert "t0_blk" in src and "t1_blk" in src and "pt_tickets + 1" in src and "t0_part0" not in src
    assert src.count("__restrict__ t") < plain.count("__restrict__ t")  # fewer kernel arguments than with partial arrays
    assert codegen.multi_finish_layout(2, 7) == (28, 30) and codegen.multi_finish_layout(3, 64) == (384, 388)
    assert f"(t0_blk + {28})" in src and f"(t1_blk + {2 * 64})" in src  # term 0's first finished value; term 1's second pair array
    assert len(ffi.jit_compile(src, "multi_probe_f.hip")) > 1000


def test_branch_guards_of_lazy_ifelse():
    """executor.branch_guards: nodes that reach the outputs only through one branch of an IfElse are
    guarded by it (innermost conditional when they nest); anything shared with an unconditional
    consumer, the condition's own producers and graph outputs stay eager.  The oracle's interpreter
    carries the same analysis (np_graph._branch_guards) and must agree."""
    import np_graph
    from pytensor_amd.executor import branch_guards
    from pytensor_amd.ir import Graph

    g = Graph(name="guards")
    x = g.new_var("float64", (None,), name="x")
    c1, c2 = g.new_var("bool", (), name="c1"), g.new_var("bool", (), name="c2")
    v = {k: g.new_var("float64", (None,)) for k in "abcdefgho"}
    ew = lambda op: {"scalar": {"in_dtypes": ["float64"], "out_dtypes": ["float64"], "body": [{"op": op, "in": [["i", 0]], "dtype": "float64"}], "outs": [["t", 0]]}}
    g.add_node("Elemwise", ew("Exp"), [x], [v["a"]])        # 0: only the inner "then" branch
    g.add_node("Elemwise", ew("Sqr"), [x], [v["b"]])        # 1: inner "else" branch AND an output -> eager
    g.add_node("IfElse", {"n_outs": 1}, [c2, v["a"], v["b"]], [v["c"]])  # 2: itself inside the outer "then" branch
    g.add_node("Elemwise", ew("Neg"), [v["c"]], [v["d"]])   # 3: outer "then"
    g.add_node("Elemwise", ew("Abs"), [x], [v["e"]])        # 4: outer "else" ...
    g.add_node("Elemwise", ew("Log1p"), [v["e"]], [v["f"]])  # 5: ... chain of two
    g.add_node("IfElse", {"n_outs": 1}, [c1, v["d"], v["f"]], [v["o"]])  # 6
    g.inputs, g.outputs = [x, c1, c2], [v["o"], v["b"]]
    guard, members = branch_guards(g)
    assert guard == [(2, 0), None, (6, 0), (6, 0), (6, 1), (6, 1), None]
    assert members == {(2, 0): [0], (2, 1): [], (6, 0): [2, 3], (6, 1): [4, 5]}
    g2, m2 = np_graph._branch_guards(g)
    assert (g2, m2) == (guard, members)
    xv = np.array([0.5, -2.0, 3.0])
    for a in (True, False):
        for b in (True, False):
            out, sq = np_graph.run_graph(g, [xv, np.asarray(a), np.asarray(b)])
            want = (-(np.exp(xv) if b else xv**2)) if a else np.log1p(np.abs(xv))
            np.testing.assert_allclose(out, want, rtol=1e-15)
            np.testing.assert_allclose(sq, xv**2, rtol=1e-15)


def test_shape_asserts_fused_into_device_composites_go_back_to_the_host():
    """The GP marginal likelihood's broadcast ``Assert``s compare ``Shape_i`` values inside the same
    ``Composite`` as the log-likelihood: split out (hostsplit.py), no assert condition / ``ARange`` /
    ``Alloc`` shape operand is a device value any more — nothing forces a stream synchronisation, so
    the graph can be frozen into a plan.  Values are those of the original graph."""
    from pytensor_amd import hostsplit
    from pytensor_amd.passes import run_pipeline

    for name in ("gp_marginal_likelihood", "hmm_garch_scans"):
        g, ins, *_ = load_case(name)
        assert hostsplit.device_reads_for_control(g)
        g2 = hostsplit.split_host_shape_arithmetic(g)
        assert g2 is not g and not hostsplit.device_reads_for_control(g2)
        for a, b in zip(np_graph.run_graph(g, ins), np_graph.run_graph(g2, ins)):
            np.testing.assert_array_equal(a, b)
        g3, _ = run_pipeline(g)
        assert not hostsplit.device_reads_for_control(g3)
    # a graph without such nodes is returned as is
    g, *_ = load_case("c4_hier")
    assert hostsplit.split_host_shape_arithmetic(g) is g


def test_stabilised_logsumexp_becomes_one_reduction():
    """tests/benchmarks/test_logsumexp.py:9-13 after the reference's rewrites is Max, a second Max of the shifted values
    and a Sum of Exp (three passes over X).  axisfuse: the Sum of Exp read by a Log is ONE LogSumExp reduction, the
    shifts cancel against the additions outside, the Max reductions die."""
    for name in ("logsumexp_axis0", "logsumexp_axis1"):
        g, ins, cvm, py, meta = load_case(name)
        assert [n.op for n in g.nodes].count("CAReduce") == 3
        g2, _ = _pipeline(g)
        ops = [n.op for n in g2.nodes]
        assert "CAReduce" not in ops and ops.count("ElemwiseAxisReduce") == 1
        r = next(n for n in g2.nodes if n.op == "ElemwiseAxisReduce")
        assert [s["op"] for s in r.params["reduce"]] == ["LogSumExp"] and len(r.inputs) == 1  # reads X, nothing else
        assert not r.params["scalar"]["body"]  # the reduced expression is X itself
        out = np_graph.run_graph(g2, ins)
        np.testing.assert_allclose(out[0], cvm[0], rtol=1e-13)


def test_elemwise_feeding_axis_reductions_is_fused_per_axis_tuple():
    g, ins, cvm, py, meta = load_case("elemwise_axis_reduce")
    g2, _ = _pipeline(g)
    ops = [n.op for n in g2.nodes]
    # six row / column reductions of fused expressions (two of them outputs of ONE multi-output Elemwise over different
    # axes: one fused node per axis tuple), no Elemwise left that stores a full-size intermediate
    assert ops.count("ElemwiseAxisReduce") >= 5 and ops.count("Elemwise") <= 1
    for n in g2.nodes:
        if n.op == "ElemwiseAxisReduce":
            assert 0 < len(n.params["axis"]) < g2.vars[n.inputs[0]].ndim


def test_cheap_producer_is_recomputed_instead_of_stored():
    """X - m (m a broadcast row / column) read by a Max and by an Exp/Sum: cloned per reader, never materialised"""
    from pytensor_amd.axisfuse import duplicate_cheap_producers

    g, *_ = load_case("logsumexp_axis1")
    subs = [n for n in g.nodes if n.op == "Elemwise" and [b["op"] for b in n.params["scalar"]["body"]] == ["Sub"]]
    assert subs, "the graph has a plain X - m"
    g2 = duplicate_cheap_producers(g)
    cons = {}
    for n in g2.nodes:
        for i in n.inputs:
            cons[i] = cons.get(i, 0) + 1
    for n in g2.nodes:
        if n.op == "Elemwise" and [b["op"] for b in n.params["scalar"]["body"]] == ["Sub"]:
            assert cons.get(n.outputs[0], 0) == 1


def _assert_def_before_use(g, what=""):
    defined = set(g.inputs) | {v for v, var in g.vars.items() if var.const is not None or var.kind == "none"}
    for k, n in enumerate(g.nodes):
        for i in n.inputs:
            assert i in defined, f"{what}: node {k} ({n.op}) reads variable {i} before its producer runs"
        defined.update(n.outputs)
        for m in n.params.get("nodes", []) if isinstance(n.params.get("nodes"), list) else []:
            defined.update(getattr(m, "outputs", []))
    for o in g.outputs:
        assert o in defined, f"{what}: output {o} is never produced"


def test_axis_fusion_keeps_definition_before_use():
    """[Elemwise(2 outs), Sum1, DimShuffle1, Sum2, DimShuffle2] — the reference's natural toposort of
    two keepdims sums of a two-output Composite.  The fused node was placed at the LAST absorbed reduction, so
    DimShuffle1 ran before the node that produces its input (np_graph: KeyError; DCE could drop the producer)."""
    from pytensor_amd.axisfuse import fuse_elemwise_axis_reduce
    from pytensor_amd.ir import Graph

    g = Graph(name="two_keepdims_sums")
    x = g.new_var("float64", (None, None), name="x")
    y = g.new_var("float64", (None, None), name="y")
    g.inputs = [x, y]
    e0, e1 = g.new_var("float64", (None, None)), g.new_var("float64", (None, None))
    body = {"in_dtypes": ["float64", "float64"], "out_dtypes": ["float64", "float64"],
            "body": [{"op": "Mul", "in": [["i", 0], ["i", 1]], "dtype": "float64"}, {"op": "Add", "in": [["i", 0], ["i", 1]], "dtype": "float64"}],
            "outs": [["t", 0], ["t", 1]]}
    g.add_node("Elemwise", {"scalar": body}, [x, y], [e0, e1])
    outs = []
    for e in (e0, e1):
        s = g.new_var("float64", (None,))
        g.add_node("CAReduce", {"scalar_op": "Add", "axis": [1], "acc_dtype": "float64", "dtype": "float64"}, [e], [s])
        d = g.new_var("float64", (None, 1))
        g.add_node("DimShuffle", {"new_order": [0, "x"], "input_ndim": 1}, [s], [d])
        outs.append(d)
    g.outputs = outs
    rng = np.random.default_rng(0)
    ins = [rng.normal(size=(5, 7)), rng.normal(size=(5, 7))]
    want = np_graph.run_graph(g, ins)
    for g2 in (fuse_elemwise_axis_reduce(g), _pipeline(g)[0]):
        ops = [n.op for n in g2.nodes]
        assert ops.count("ElemwiseAxisReduce") == 1 and "CAReduce" not in ops
        _assert_def_before_use(g2, "two keepdims sums")
        got = np_graph.run_graph(g2, ins)
        for a, b in zip(got, want):
            np.testing.assert_allclose(a, b, rtol=1e-14)
        np.testing.assert_allclose(got[0], (ins[0] * ins[1]).sum(1, keepdims=True), rtol=1e-14)


@pytest.mark.parametrize("name", golden_cases())
def test_passes_keep_definition_before_use(name):
    g, *_ = load_case(name)
    _assert_def_before_use(_pipeline(g)[0], name)


def test_identity_elemwise_behind_a_fused_logsumexp_is_dropped():
    """round 6: what is left of ``log(sum(exp(x - m))) + m`` after fuse_logsumexp is an Elemwise of Identity nodes — a copy
    launch; its readers (here: the graph output) take its operand instead.  Never for a graph input (outputs must not
    alias what the caller owns)."""
    from pytensor_amd.axisfuse import drop_identity_elemwise
    from pytensor_amd.ir import Graph

    for name in ("logsumexp_axis0", "logsumexp_axis1"):
        g, ins, cvm, py, meta = load_case(name)
        g2, _ = _pipeline(g)
        assert [n.op for n in g2.nodes if n.op not in ("DimShuffle",)] == ["ElemwiseAxisReduce"], [n.op for n in g2.nodes]
        _assert_def_before_use(g2, name)
        np.testing.assert_allclose(np_graph.run_graph(g2, ins)[0], cvm[0], rtol=1e-13)
    # an identity of a graph INPUT stays (the output would alias the caller's array)
    g = Graph(name="id_of_input")
    x = g.new_var("float64", (None,), name="x")
    g.inputs = [x]
    y = g.new_var("float64", (None,))
    body = {"in_dtypes": ["float64"], "out_dtypes": ["float64"], "body": [{"op": "Identity", "in": [["i", 0]], "dtype": "float64"}], "outs": [["t", 0]]}
    g.add_node("Elemwise", {"scalar": body}, [x], [y])
    g.outputs = [y]
    assert [n.op for n in drop_identity_elemwise(g).nodes] == ["Elemwise"]


def test_generated_scalar_code_shares_exp_and_reciprocals():
    """round 6 (codegen.emit_body): sigmoid and softplus of ONE float64 operand -> one pt_sig_sp; divisions by one float64
    denominator -> one reciprocal, but only where every output is summed (flat_kernel_source decides): an element-wise
    output keeps its exact quotient."""
    from pytensor_amd import codegen

    f64 = "float64"
    body = {"in_dtypes": [f64, f64], "out_dtypes": [f64, f64],
            "body": [{"op": "Sigmoid", "in": [["i", 0]], "dtype": f64}, {"op": "Softplus", "in": [["i", 0]], "dtype": f64},
                     {"op": "TrueDiv", "in": [["t", 0], ["i", 1]], "dtype": f64}, {"op": "TrueDiv", "in": [["t", 1], ["i", 1]], "dtype": f64}],
            "outs": [["t", 2], ["t", 3]]}
    summed = codegen.flat_kernel_source("k_sum", body, "VV", 2, [("Add", f64), ("Add", f64)], 2)
    stored = codegen.flat_kernel_source("k_store", body, "VV", 2, [None, None], 2)
    loop = lambda src: src[src.index('extern "C" __global__'):]
    assert "pt_sig_sp(" in loop(summed) and "pt_sigmoid(" not in loop(summed) and "pt_softplus(" not in loop(summed)
    assert "_rcp = 1.0 / " in loop(summed) and " / (double)a1" not in loop(summed).replace("1.0 / (double)a1", "")
    assert "pt_sig_sp(" in loop(stored)  # sharing the exp changes no quotient: allowed everywhere
    assert "_rcp" not in loop(stored), "an element-wise output must keep x / d"
    # different operands: nothing to share
    body2 = {"in_dtypes": [f64, f64], "out_dtypes": [f64, f64],
             "body": [{"op": "Sigmoid", "in": [["i", 0]], "dtype": f64}, {"op": "Softplus", "in": [["i", 1]], "dtype": f64}], "outs": [["t", 0], ["t", 1]]}
    assert "pt_sig_sp(" not in loop(codegen.flat_kernel_source("k2", body2, "VV", 2, [("Add", f64), ("Add", f64)], 2))
