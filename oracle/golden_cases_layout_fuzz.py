"""Seeded random LAYOUTS (imported by ``make_golden.py``): the same few expressions over operands that reach the
linker as transposed / strided / reversed / broadcast views of what the caller passed.  TEST INFRASTRUCTURE.

What the hand-written layout cases of ``golden_cases_r5.py`` pin for chosen shapes, these pin for shapes nobody chose:
every case draws a rank (2-4) and extents from a pool that straddles the tile sizes of the N-d kernels (1, 2, 3, 5, 8,
17, 33, 64, 65, 130), builds each operand as ``x[slices].dimshuffle(perm)`` of a differently laid out input (steps of
+-1 / +-2, any permutation, broadcast dimensions of extent 1, a missing leading dimension), and ends in an elementwise
graph, reductions (sum / max / min / prod / all / any) over random axis subsets, ``logsumexp`` and ``softmax`` along
random axes.  Reference entry points: Elemwise.perform / CAReduce.perform (pytensor/tensor/elemwise.py:375, 1233),
special.py:102 (LogSumExp), special.py Softmax.
"""

from __future__ import annotations

import numpy as np
import pytensor.tensor as pt

from make_golden import case

_POOL = [1, 2, 3, 5, 8, 17, 33, 64, 65, 130]


def _shape(rng, rank, budget=14000):
    while True:
        s = [int(_POOL[rng.integers(len(_POOL))]) for _ in range(rank)]
        if 8 <= int(np.prod(s)) <= budget:
            return s


def _operand(rng, name, shape, dtype, allow_bcast=True):
    """(symbolic view of shape `shape` (or broadcastable to it), input variable, input value)"""
    rank = len(shape)
    shp = list(shape)
    if allow_bcast:
        for d in range(rank):
            if shp[d] > 1 and rng.random() < 0.2:
                shp[d] = 1
        drop = 0
        while drop < rank - 1 and rng.random() < 0.15:
            drop += 1
        shp = shp[drop:]
    r = len(shp)
    perm = [int(p) for p in rng.permutation(r)]  # storage axis k holds view axis perm[k]
    steps = [int(rng.choice([1, 1, -1, 2, -2])) for _ in range(r)]
    store = [shp[perm[k]] * abs(steps[k]) + (int(rng.integers(0, 2)) if abs(steps[k]) == 2 else 0) for k in range(r)]
    if dtype == "int64":
        val = rng.integers(-4, 5, size=store).astype("int64")
    else:
        val = (rng.normal(size=store) * 1.2).astype(dtype)
    x = pt.tensor(name, dtype=dtype, shape=(None,) * r)
    sl = []
    for k in range(r):
        n, st = shp[perm[k]], steps[k]
        if st > 0:
            sl.append(slice(0, n * st, st))
        else:
            sl.append(slice(n * (-st) - 1, None, st))
    v = x[tuple(sl)]
    inv = [perm.index(d) for d in range(r)]
    v = v.dimshuffle(*inv)
    # (extent-1 dimensions broadcast at run time only where the type says so: make them static)
    bc = [d for d in range(r) if shp[d] == 1 and shape[len(shape) - r + d] != 1]
    if bc:
        v = pt.specify_broadcastable(v, *bc)
    return v, x, val


def _axes(rng, rank):
    k = int(rng.integers(1, rank + 1))
    return tuple(sorted(int(a) for a in rng.choice(rank, size=k, replace=False)))


def _make(seed, dtype):
    def build():
        rng = np.random.default_rng(7000 + seed)
        rank = int(rng.integers(2, 5))
        shape = _shape(rng, rank)
        a, xa, va = _operand(rng, "a", shape, dtype, allow_bcast=False)
        b, xb, vb = _operand(rng, "b", shape, dtype)
        c, xc, vc = _operand(rng, "c", shape, dtype)
        ins, vals = [xa, xb, xc], {"a": va, "b": vb, "c": vc}
        outs = []
        if dtype == "int64":
            e = a * b + c
            outs.append(e)
            outs.append(pt.sum(e, axis=_axes(rng, rank)))
            outs.append(pt.max(a - c, axis=_axes(rng, rank)))
            outs.append(pt.min(a + b, axis=_axes(rng, rank)))
            outs.append(pt.any(pt.gt(e, 6), axis=_axes(rng, rank)))
            outs.append(pt.all(pt.lt(a, 4), axis=_axes(rng, rank)))
            return ins, outs, vals
        # (no sum or difference BEHIND an inexact operation: parity is element-wise relative, and an ulp of a different
        #  tanh or a contracted multiply-add in front of a cancelling add is a large relative error in what is left)
        e = pt.tanh(a) * b * c
        outs.append(e)
        outs.append(pt.sum(pt.sqr(e) + 0.125, axis=_axes(rng, rank)))
        outs.append(pt.max(a * c, axis=_axes(rng, rank)))
        outs.append(pt.min(a - b, axis=_axes(rng, rank)))
        outs.append(pt.prod(pt.tanh(a * b) * 0.5 + 1.0, axis=_axes(rng, rank)))
        outs.append(pt.sum(pt.sqr(a), axis=None))
        ax = int(rng.integers(rank))
        outs.append(pt.logsumexp(a + 4.0, axis=ax))
        ax2 = int(rng.integers(rank))
        outs.append(pt.special.softmax(a, axis=ax2))
        if dtype == "float64":
            outs.append(pt.special.log_softmax(a - c, axis=int(rng.integers(rank))))
        return ins, outs, vals

    return build


for _s in range(10):
    case(f"layout_fuzz_f64_{_s}", rtol=1e-10)(_make(_s, "float64"))
for _s in range(3):
    case(f"layout_fuzz_f32_{_s}", rtol=1e-4)(_make(50 + _s, "float32"))
for _s in range(3):
    case(f"layout_fuzz_i64_{_s}", rtol=0)(_make(80 + _s, "int64"))


# ---- second family: the same random views into the indexing / ordering / scan ops ----
def _make2(seed):
    def build():
        rng = np.random.default_rng(9000 + seed)
        rank = int(rng.integers(2, 5))
        shape = _shape(rng, rank, budget=9000)
        a, xa, va = _operand(rng, "a", shape, "float64", allow_bcast=False)
        b, xb, vb = _operand(rng, "b", shape, "float64")
        c, xc, vc = _operand(rng, "c", shape, "float64", allow_bcast=False)
        ins, vals = [xa, xb, xc], {"a": va, "b": vb, "c": vc}
        ax = lambda: int(rng.integers(rank))  # noqa: E731
        outs = []
        outs.append(pt.argmax(a, axis=ax()))
        outs.append(pt.argmin(c, axis=ax()))
        outs.append(pt.cumsum(pt.sqr(a) + 0.125, axis=ax()))
        outs.append(pt.cumprod(pt.tanh(c) * 0.25 + 1.0, axis=ax()))
        outs.append(pt.sort(a, axis=ax()))
        outs.append(pt.argsort(c, axis=ax()))
        outs.append(pt.special.softmax(a * b, axis=ax()))
        k = ax()
        n = shape[k]
        iv = pt.tensor("iv", dtype="int64", shape=(None,))
        ins.append(iv)
        vals["iv"] = rng.integers(-n, n, size=n + 2)
        outs.append(pt.take(a, iv, axis=k))
        outs.append(pt.where(pt.gt(a, b), a, c))
        outs.append(pt.concatenate([a, c], axis=ax()))
        outs.append(pt.sum(pt.sqr(a.reshape((-1,))[::2])))
        if rank >= 2:
            i, j = sorted(int(t) for t in rng.choice(rank, size=2, replace=False))
            outs.append(pt.diagonal(a, axis1=i, axis2=j))
        outs.append(pt.max(a, axis=_axes(rng, rank)) * pt.min(c, axis=None))
        return ins, outs, vals

    return build


for _s in range(12):
    case(f"layout_fuzz2_{_s}", rtol=1e-10)(_make2(_s))
