"""CPU: the IR passes (fusion, hoisting, segmentation, donation) preserve semantics.

The oracle restates every fused node from its unfused parts (oracle/np_graph.py), so
``oracle(pass(graph)) == oracle(graph)`` checks the *rewrites* without a GPU.
"""
import numpy as np
import pytest

import np_graph
from pytensor_amd.fusion import (
    fuse_cholesky_solve,
    fuse_elemwise_reduce,
    fuse_gemv_chain,
    hoist_scan_seq_dots,
    segment_graph,
)
from util import assert_parity, golden_cases, load_case


def _pipeline(g):
    g = fuse_elemwise_reduce(g)
    g = hoist_scan_seq_dots(g)
    g = fuse_cholesky_solve(g)
    g = fuse_gemv_chain(g)
    g, seg = segment_graph(g)
    return g, seg


@pytest.mark.parametrize("name", golden_cases())
def test_passes_preserve_results(name):
    g, ins, cvm, py, meta = load_case(name)
    g2, seg = _pipeline(g)
    out = np_graph.run_graph(g2, ins)
    for k, (a, b) in enumerate(zip(out, cvm)):
        assert_parity(a, b, max(meta["rtol"], 1e-11), f"{name} out{k} after IR passes")


def test_c4_gets_the_one_pass_gemv_chain_and_two_segments():
    g, *_ = load_case("c4_hier")
    g2, seg = _pipeline(g)
    ops = [n.op for n in g2.nodes]
    assert ops.count("GemvChain") == 1 and ops.count("GemvFinish") == 2 and "Gemv" not in ops
    # the gather a[gidx] and the scatter-add of its gradient are absorbed into the chain
    assert "AdvancedSubtensor" not in ops and "AdvancedIncSubtensor" not in ops
    assert ops.count("ElemwiseReduce") >= 3
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
    inner_ops = [m.op for m in scan.params["inner"].nodes]
    assert "Dot22" not in inner_ops and inner_ops.count("Gemm") == 3
    assert scan.params["info"]["n_seqs"] == 4
    # the original graph object is untouched (passes are functional)
    g_again, *_ = load_case("c5_gru")
    assert [n.op for n in g.nodes] == [n.op for n in g_again.nodes]


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
