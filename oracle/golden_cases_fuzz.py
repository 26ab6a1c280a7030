"""Seeded random graphs (imported by ``make_golden.py``): differential parity beyond the
hand-written cases.  TEST INFRASTRUCTURE.

Each case draws a few input tensors (mixed ranks, broadcastable dims, float64 / float32 /
int64), grows random expression trees over the reference's public elementwise API, and ends
every tree in one of: the tensor itself, a reduction over a random axis subset, a gather, or a
matrix product.  The reference's rewrites (fusion, BLAS, canonicalisation) then shape the graph
the linker sees — which is the point: the lowered IR contains whatever ``Composite``s, views and
BLAS nodes the reference decides to build from them.
"""

from __future__ import annotations

import numpy as np
import pytensor.tensor as pt

from make_golden import case

_UNARY = [
    (pt.exp, (-3, 3)), (pt.tanh, (-4, 4)), (pt.sigmoid, (-8, 8)), (pt.sin, (-6, 6)), (pt.cos, (-6, 6)),
    (pt.sqr, (-3, 3)), (abs, (-3, 3)), (pt.neg, (-3, 3)), (pt.softplus, (-20, 20)), (pt.arctan, (-5, 5)),
    (pt.erf, (-3, 3)), (pt.expm1, (-2, 2)), (pt.sign, (-3, 3)), (pt.floor, (-5, 5)), (pt.ceil, (-5, 5)),
]
_UNARY_POS = [(pt.log, (0.1, 5)), (pt.sqrt, (0.0, 5)), (pt.log1p, (-0.9, 5)), (pt.gammaln, (0.2, 6)), (pt.reciprocal, (0.3, 4))]
_BINARY = [pt.add, pt.sub, pt.mul, pt.maximum, pt.minimum, pt.true_div, pt.arctan2]


def _grow(rng, leaves, depth):
    """A random expression over ``leaves`` (all broadcast-compatible)."""
    if depth == 0 or rng.random() < 0.15:
        return leaves[rng.integers(len(leaves))]
    r = rng.random()
    if r < 0.35:
        f, _ = _UNARY[rng.integers(len(_UNARY))]
        return f(_grow(rng, leaves, depth - 1))
    if r < 0.45:
        f, _ = _UNARY_POS[rng.integers(len(_UNARY_POS))]
        return f(abs(_grow(rng, leaves, depth - 1)) + 0.5)
    if r < 0.9:
        f = _BINARY[rng.integers(len(_BINARY))]
        a, b = _grow(rng, leaves, depth - 1), _grow(rng, leaves, depth - 1)
        if f is pt.true_div:
            b = abs(b) + 0.7
        return f(a, b)
    c = _grow(rng, leaves, depth - 1)
    return pt.switch(c > 0.1, _grow(rng, leaves, depth - 1), _grow(rng, leaves, depth - 1) * 0.5)


def _make(seed, dtype):
    def build():
        rng = np.random.default_rng(1000 + seed)
        n0, n1, n2 = (int(rng.integers(2, 9)) for _ in range(3))
        shapes = {"m": (n0, n1), "r": (1, n1), "c": (n0, 1), "t": (n2, n0, n1), "s": ()}
        ins, vals, leaves = [], {}, []
        for name, shp in shapes.items():
            bc = tuple(1 if s == 1 else None for s in shp)
            v = pt.tensor(name, dtype=dtype, shape=bc)
            ins.append(v)
            vals[name] = (rng.normal(size=shp) * 1.5).astype(dtype)
            leaves.append(v)
        iv = pt.tensor("iv", dtype="int64", shape=(None,))
        ins.append(iv)
        vals["iv"] = rng.integers(-n0, n0, size=n0 + 3)
        outs = []
        for _ in range(3):
            e = _grow(rng, leaves, 4)
            k = rng.integers(5)
            if e.ndim == 0 or k == 0:
                outs.append(e)
            elif k == 1:
                axes = tuple(sorted(rng.choice(e.ndim, size=int(rng.integers(1, e.ndim + 1)), replace=False).tolist()))
                red = [pt.sum, pt.max, pt.min, pt.mean, pt.prod][rng.integers(5)]
                outs.append(red(pt.tanh(e) if red is pt.prod else e, axis=axes))
            elif k == 2:
                outs.append(e.sum())
            elif k == 3 and e.ndim >= 2:
                x2 = e if e.ndim == 2 else e[0]
                outs.append(pt.dot(x2, pt.tanh(x2).T))
            else:
                x2 = e if e.ndim <= 2 else e[-1]
                x2 = pt.broadcast_to(x2, (n0, n1)) if x2.ndim == 2 else x2
                outs.append(x2[iv] if x2.ndim >= 1 else x2)
        return ins, outs, vals

    return build


for _s in range(8):
    case(f"fuzz_f64_{_s}", rtol=1e-10)(_make(_s, "float64"))
for _s in range(4):
    case(f"fuzz_f32_{_s}", rtol=5e-5)(_make(100 + _s, "float32"))
