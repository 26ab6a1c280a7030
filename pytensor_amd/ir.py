"""Portable lowered-graph IR for the ``hip`` linker.

``HipLinker.fgraph_convert`` (``pytensor_amd/linker.py``) lowers a rewritten
PyTensor ``FunctionGraph`` into this IR; the device executor
(``pytensor_amd/executor.py``) runs it through the C-ABI, and the CPU oracle
(``oracle/np_graph.py``) interprets the *same* IR with NumPy/SciPy.  The IR is
JSON-serialisable so that graphs lowered where PyTensor is importable travel as
fixtures (``tests/golden/*.json``) to machines where it is not.

Vocabulary follows the reference: ``Apply`` nodes become :class:`Node` (op name
= the reference ``Op`` class name), ``Variable`` becomes :class:`Var`
(reference: pytensor/graph/basic.py:192,359,744).
"""

from __future__ import annotations

import base64
import json
from dataclasses import dataclass, field

import numpy as np

_INLINE_MAX = 64  # elements; larger constants are stored base64


@dataclass
class Var:
    id: int
    dtype: str
    shape: tuple  # static shape, ``None`` for unknown dims (TensorType.shape)
    kind: str = "tensor"  # "tensor" | "scalar" (ScalarType) | "slice" | "none"
    const: np.ndarray | None = None
    name: str | None = None

    @property
    def ndim(self) -> int:
        return len(self.shape)


@dataclass
class Node:
    op: str
    params: dict
    inputs: list
    outputs: list


@dataclass
class Graph:
    vars: dict = field(default_factory=dict)
    nodes: list = field(default_factory=list)
    inputs: list = field(default_factory=list)
    outputs: list = field(default_factory=list)
    name: str = "graph"

    # -- construction ------------------------------------------------------
    def new_var(self, dtype, shape, kind="tensor", const=None, name=None) -> int:
        vid = len(self.vars)
        self.vars[vid] = Var(vid, str(dtype), tuple(shape), kind, const, name)
        return vid

    def add_node(self, op, params, inputs, outputs) -> Node:
        n = Node(op, params, list(inputs), list(outputs))
        self.nodes.append(n)
        return n

    # -- serialisation -----------------------------------------------------
    def to_dict(self) -> dict:
        return {
            "name": self.name,
            "inputs": self.inputs,
            "outputs": self.outputs,
            "vars": [_var_to_dict(v) for v in self.vars.values()],
            "nodes": [
                {
                    "op": n.op,
                    "params": _params_to_json(n.params),
                    "inputs": n.inputs,
                    "outputs": n.outputs,
                }
                for n in self.nodes
            ],
        }

    def to_json(self, **kw) -> str:
        return json.dumps(self.to_dict(), **kw)

    @classmethod
    def from_dict(cls, d: dict) -> "Graph":
        g = cls(name=d.get("name", "graph"))
        for vd in d["vars"]:
            v = _var_from_dict(vd)
            g.vars[v.id] = v
        for nd in d["nodes"]:
            g.nodes.append(
                Node(nd["op"], _params_from_json(nd["params"]), nd["inputs"], nd["outputs"])
            )
        g.inputs = list(d["inputs"])
        g.outputs = list(d["outputs"])
        return g

    @classmethod
    def from_json(cls, s: str) -> "Graph":
        return cls.from_dict(json.loads(s))

    def summary(self) -> str:
        from collections import Counter

        c = Counter(n.op for n in self.nodes)
        return ", ".join(f"{k}×{v}" for k, v in sorted(c.items()))


def encode_array(a: np.ndarray) -> dict:
    a = np.asarray(a)
    if a.size <= _INLINE_MAX and a.dtype.kind in "biuf":
        return {"dtype": str(a.dtype), "shape": list(a.shape), "data": _tolist_exact(a)}
    return {
        "dtype": str(a.dtype),
        "shape": list(a.shape),
        "b64": base64.b64encode(np.ascontiguousarray(a).tobytes()).decode("ascii"),
    }


def _tolist_exact(a: np.ndarray):
    # floats are stored as hex so that the round trip is bit exact (nan/inf included)
    if a.dtype.kind == "f":
        return [float(x).hex() for x in a.astype(np.float64).ravel()]
    return [int(x) for x in a.ravel()]


def decode_array(d: dict) -> np.ndarray:
    dt = np.dtype(d["dtype"])
    if "b64" in d:
        return np.frombuffer(base64.b64decode(d["b64"]), dtype=dt).reshape(d["shape"]).copy()
    data = d["data"]
    if dt.kind == "f":
        arr = np.array([float.fromhex(x) for x in data], dtype=np.float64).astype(dt)
    else:
        arr = np.array(data, dtype=dt)
    return arr.reshape(d["shape"])


def _var_to_dict(v: Var) -> dict:
    d = {"id": v.id, "dtype": v.dtype, "shape": list(v.shape), "kind": v.kind}
    if v.name:
        d["name"] = v.name
    if v.const is not None:
        d["const"] = encode_array(v.const)
    return d


def _var_from_dict(d: dict) -> Var:
    const = decode_array(d["const"]) if "const" in d else None
    return Var(d["id"], d["dtype"], tuple(d["shape"]), d.get("kind", "tensor"), const, d.get("name"))


def _params_to_json(p):
    if isinstance(p, Graph):
        return {"__graph__": p.to_dict()}
    if isinstance(p, np.ndarray):
        return {"__array__": encode_array(p)}
    if isinstance(p, (np.integer,)):
        return int(p)
    if isinstance(p, (np.floating,)):
        return {"__array__": encode_array(np.asarray(p))}
    if isinstance(p, (np.bool_,)):
        return bool(p)
    if isinstance(p, slice):
        return {"__slice__": [_params_to_json(p.start), _params_to_json(p.stop), _params_to_json(p.step)]}
    if isinstance(p, dict):
        return {str(k): _params_to_json(v) for k, v in p.items()}
    if isinstance(p, (list, tuple)):
        return [_params_to_json(x) for x in p]
    return p


def _params_from_json(p):
    if isinstance(p, dict):
        if "__graph__" in p:
            return Graph.from_dict(p["__graph__"])
        if "__array__" in p:
            return decode_array(p["__array__"])
        if "__slice__" in p:
            a, b, c = (_params_from_json(x) for x in p["__slice__"])
            return slice(a, b, c)
        return {k: _params_from_json(v) for k, v in p.items()}
    if isinstance(p, list):
        return [_params_from_json(x) for x in p]
    return p
