"""Gather reads absorbed into the fused elementwise loop.

The reference has this rewrite for its Numba backend only (``FusedElemwise``,
pytensor/tensor/rewriting/fused_elemwise.py:107-875): an ``AdvancedSubtensor`` that feeds an
``Elemwise`` is read inside the fused loop instead of materialising ``table[idx]`` (an N-sized
intermediate written to and re-read from HBM, plus a launch).  ``fuse_gemv_chain`` already
does it for the one-pass regression kernel; this pass covers every other ``Elemwise`` /
``ElemwiseReduce`` over 1-d operands, e.g. a hierarchical model without a design matrix
(``eta = a[gidx]``).

Representation: the node keeps its scalar graph; body input ``pos`` becomes the *table*
variable, the index vector is appended after the body inputs, and
``params["gather"] = [[pos, extra], …]`` records the pairing (``extra`` = position among the
appended inputs).  The generated kernel reads ``table[idx[i]]`` with NumPy index semantics
(negative wrap, IndexError through the device error flag).
"""

from __future__ import annotations

from pytensor_amd.inline import MAX_INPUTS, _copy, _index
from pytensor_amd.ir import Graph, Node

_EW = ("Elemwise", "ElemwiseReduce")


def absorb_gathers(g: Graph) -> Graph:
    while True:
        producer, consumers = _index(g)
        out_set = set(g.outputs)
        hit = None
        for ke, E in enumerate(g.nodes):
            if E.op not in _EW or not E.outputs or E.params.get("partial_inputs"):
                continue
            nbody = len(E.params["scalar"]["in_dtypes"])
            # iteration rank = rank of the body inputs (reduced outputs of an ElemwiseReduce are 0-d)
            if nbody == 0 or max(g.vars[i].ndim for i in E.inputs[:nbody]) != 1:
                continue
            if len(E.inputs) + 1 > MAX_INPUTS:
                continue
            taken = {p for p, _ in (E.params.get("gather") or [])}
            for pos, v in enumerate(E.inputs[:nbody]):
                if pos in taken or v in out_set or consumers.get(v, []) != [ke]:
                    continue
                kp = producer.get(v)
                if kp is None:
                    continue
                P = g.nodes[kp]
                if P.op != "AdvancedSubtensor" or P.params.get("idx_list") != [0] or len(P.inputs) != 2:
                    continue
                table, idx = P.inputs
                if g.vars[table].ndim != 1 or g.vars[idx].ndim != 1 or g.vars[idx].dtype != "int64":
                    continue
                if g.vars[table].dtype != E.params["scalar"]["in_dtypes"][pos]:
                    continue
                hit = (ke, pos, kp)
                break
            if hit:
                break
        if hit is None:
            return g
        ke, pos, kp = hit
        E, P = g.nodes[ke], g.nodes[kp]
        nbody = len(E.params["scalar"]["in_dtypes"])
        table, idx = P.inputs
        extras = list(E.inputs[nbody:])
        if idx in extras:
            extra = extras.index(idx)
        else:
            extra = len(extras)
            extras.append(idx)
        params = dict(E.params)
        params["gather"] = [list(p) for p in (E.params.get("gather") or [])] + [[pos, extra]]
        body_inputs = list(E.inputs[:nbody])
        body_inputs[pos] = table
        merged = Node(E.op, params, body_inputs + extras, list(E.outputs))
        nodes = [merged if k == ke else n for k, n in enumerate(g.nodes) if k != kp]
        g = _copy(g, nodes)
