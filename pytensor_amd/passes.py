"""The IR pass pipeline of the ``hip`` linker (one place, used by the executor and the tests).

Order matters:

0. ``split_host_shape_arithmetic`` (hostsplit.py) — 0-d integer shape arithmetic the reference fused into a
   device ``Composite`` goes back to the host (no device read for an ``Assert`` on shapes: the graph can freeze);
1. ``push_gather_through_elemwise`` / ``inline_elemwise_producers`` (inline.py) — finish the
   elementwise fusion the reference's ``FusionOptimizer`` stops short of;
2. ``fuse_elemwise_reduce`` (fusion.py) — full reductions folded into the producing kernel;
   ``duplicate_cheap_producers`` / ``fuse_elemwise_axis_reduce`` (axisfuse.py) — reductions over SOME axes too;
3. ``merge_sibling_reductions`` (inline.py);
4. ``hoist_scan_seq_dots`` — sequence-only products out of ``Scan``;
5. ``fuse_cholesky_solve`` — Cholesky + its first triangular solve, factor kept in LDS;
   ``defer_gemm_finish`` (gemmfuse.py) — split-K finish + Gemm epilogue folded into the
   consuming elementwise kernel (also inside Scan inner graphs); ``fuse_dot_epilogue`` —
   inside a Scan step, product against a loop-constant matrix + consuming ``Composite`` in one
   generated kernel (``DotEpilogue``; weights repacked once outside the loop);
6. ``fuse_gemv_chain`` — ``X@b → Composite → X.T@w`` (+ gathers, + scatter-add) in one pass;
7. ``dead_code_elimination``; ``collect_scalar_updates`` / ``fuse_independent_reductions``
   (widefuse.py) — wide graphs: scalar ``IncSubtensor`` chains as one node, independent
   ``ElemwiseReduce`` terms in one launch; ``fuse_tail`` (tailfuse.py) — the small nodes at the end
   of the graph in two launches;
8. ``segment_graph`` — latency chain / streaming / combine segments for multi-stream plans.

``fuse=False`` leaves the lowered graph untouched (one launch per reference ``Apply``: the
parity tests compare the two); ``fuse="elemwise"`` stops after step 2.
"""

from __future__ import annotations

import os

from pytensor_amd.axisfuse import drop_identity_elemwise, duplicate_cheap_producers, fuse_elemwise_axis_reduce, fuse_logsumexp
from pytensor_amd.fusion import (
    fuse_cholesky_solve,
    fuse_elemwise_reduce,
    fuse_gemv_chain,
    hoist_scan_seq_dots,
    segment_graph,
)
from pytensor_amd.gatherfuse import absorb_gathers
from pytensor_amd.gemmfuse import defer_gemm_finish, fuse_dot_epilogue, merge_sibling_gemms
from pytensor_amd.inline import (
    dead_code_elimination,
    inline_elemwise_producers,
    merge_sibling_reductions,
    push_gather_through_elemwise,
)
from pytensor_amd.hostsplit import split_host_shape_arithmetic
from pytensor_amd.ir import Graph
from pytensor_amd.tailfuse import fuse_tail
from pytensor_amd.widefuse import collect_scalar_updates, fuse_independent_reductions


def run_pipeline(graph: Graph, fuse=True, tail=True):
    """Returns ``(graph, segments)``; ``segments`` is ``None`` when there is nothing to overlap.
    ``tail=False`` (Scan inner graphs: run once per step, nothing small enough to matter) skips
    the tail fusion."""
    if not fuse:
        return graph, None
    if fuse == "elemwise":
        return fuse_elemwise_reduce(graph), None
    g = split_host_shape_arithmetic(graph)  # shape asserts fused into device Composites: back to the host
    if os.environ.get("PTHIP_AXIS_FUSE", "1") != "0":
        g = duplicate_cheap_producers(g)  # `X - m` read by a Max and an Exp/Sum: recomputed per client, never stored
    g = inline_elemwise_producers(push_gather_through_elemwise(g))
    g = fuse_elemwise_reduce(g)
    if os.environ.get("PTHIP_AXIS_FUSE", "1") != "0":
        g = fuse_elemwise_axis_reduce(g)  # row / column reductions of a fused expression in one kernel
        if os.environ.get("PTHIP_LSE_FUSE", "1") != "0":
            g = drop_identity_elemwise(fuse_logsumexp(g))  # log(sum(exp(x - max))) + max: one pass, the Max reductions die
    g = merge_sibling_reductions(g)
    g = hoist_scan_seq_dots(g)
    g = fuse_cholesky_solve(g)
    g = defer_gemm_finish(g)
    if os.environ.get("PTHIP_DOT_EPILOGUE", "1") != "0":
        g = fuse_dot_epilogue(g)
    g = merge_sibling_gemms(g)  # what is left as split-K slabs
    g = absorb_gathers(fuse_gemv_chain(g))  # gchain takes the gathers it can use first
    g = dead_code_elimination(g)
    if os.environ.get("PTHIP_WIDE", "1") != "0":
        g = fuse_independent_reductions(collect_scalar_updates(g))
    if tail and os.environ.get("PTHIP_TAIL", "1") != "0":
        g = fuse_tail(g)
    return segment_graph(g)
