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
    elif np.dtype(dtype).kind in "iu":
        info = np.iinfo(dtype)
        val = rng.integers(max(info.min, -100), min(info.max, 100) + 1, size=store).astype(dtype)
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
def _make2(seed, dtype="float64"):
    def build():
        rng = np.random.default_rng(9000 + seed)
        rank = int(rng.integers(2, 5))
        shape = _shape(rng, rank, budget=9000)
        a, xa, va = _operand(rng, "a", shape, dtype, allow_bcast=False)
        b, xb, vb = _operand(rng, "b", shape, dtype)
        c, xc, vc = _operand(rng, "c", shape, dtype, allow_bcast=False)
        if np.dtype(dtype).kind in "iu":
            # distinct values (argmax / argsort of ties is a convention, not arithmetic): a permutation plus noise-free offsets
            va = (rng.permutation(va.size).reshape(va.shape) - va.size // 2).astype(dtype)
            vc = (rng.permutation(vc.size).reshape(vc.shape) * 3 - vc.size).astype(dtype)
            ins, vals = [xa, xb, xc], {"a": va, "b": vb, "c": vc}
            ax = lambda: int(rng.integers(rank))  # noqa: E731
            k = ax()
            n = shape[k]
            iv = pt.tensor("iv", dtype="int64", shape=(None,))
            ins.append(iv)
            vals["iv"] = rng.integers(-n, n, size=n + 2)
            outs = [pt.argmax(a, axis=ax()), pt.argmin(c, axis=ax()), pt.cumsum(a, axis=ax()), pt.sort(a, axis=ax()), pt.argsort(c, axis=ax()),
                    pt.take(a, iv, axis=k), pt.where(pt.gt(a, b), a, c), pt.concatenate([a, c], axis=ax()), pt.sum(a.reshape((-1,))[::2]),
                    pt.max(a, axis=_axes(rng, rank)) - pt.min(c, axis=None)]
            return ins, outs, vals
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
for _s in range(3):
    case(f"layout_fuzz2_f32_{_s}", rtol=1e-4)(_make2(40 + _s, "float32"))
for _s in range(3):
    case(f"layout_fuzz2_i64_{_s}", rtol=0)(_make2(60 + _s, "int64"))


# ---- third family: scalar semantics across dtypes (integer division / modulo signs, rounding, overflow wrap-around,
#      bitwise ops, comparisons, casts) on the same random views ----
_INT_DTYPES = ["int8", "int16", "int32", "int64", "uint8", "uint16"]


def _make3(seed):
    def build():
        rng = np.random.default_rng(11000 + seed)
        rank = int(rng.integers(1, 4))
        shape = _shape(rng, rank, budget=3000) if rank > 1 else [int(rng.choice([7, 64, 130, 1000]))]
        di, dj = (str(rng.choice(_INT_DTYPES)) for _ in range(2))
        fdt = str(rng.choice(["float64", "float32"]))

        def ints(name, dt, nonzero=False):
            v, x, val = _operand(rng, name, shape, dt, allow_bcast=name != "i")
            if dt == "int64":
                val = rng.integers(-100, 101, size=val.shape).astype(dt)
            if nonzero:
                val = np.where(val == 0, 3, val).astype(dt)
            return v, x, val

        i, xi, vi = ints("i", di)
        j, xj, vj = ints("j", dj, nonzero=True)
        f, xf, vf = _operand(rng, "f", shape, fdt, allow_bcast=False)
        g, xg, vg = _operand(rng, "g", shape, fdt)
        vg = np.where(np.abs(vg) < 0.05, 0.7, vg).astype(fdt)
        ins, vals = [xi, xj, xf, xg], {"i": vi, "j": vj, "f": vf * 3, "g": vg}
        outs = [
            i // j, i % j, i * j + i, i - j, pt.bitwise_and(i, j), pt.bitwise_or(i, j), pt.bitwise_xor(i, j), pt.invert(i), abs(i), -i,
            pt.eq(i, j), pt.lt(i, j), pt.ge(i, j), pt.maximum(i, j), pt.minimum(i, j), pt.sgn(i), pt.cast(i, "float32") / pt.cast(j, "float32"),
            pt.floor(f), pt.ceil(f), pt.round(f), pt.trunc(f), f // g, f % g, pt.sgn(f), pt.clip(f, -1.0, g * g),
            pt.cast(f, "int32"), pt.cast(pt.clip(f * 20, -120.0, 120.0), "int8"), pt.cast(i, fdt) * g, pt.switch(pt.gt(f, g), i, j), pt.isinf(1.0 / pt.floor(abs(f))),
            pt.sum(i, axis=None), pt.sum(pt.cast(i, "int8"), axis=None, dtype="int8"), pt.prod(pt.clip(j, -2, 2), axis=None), pt.max(i, axis=0), pt.min(j, axis=rank - 1),
            pt.any(pt.gt(i, 90), axis=None), pt.all(pt.neq(j, 0), axis=0), pt.mean(pt.sqr(f), axis=None), pt.var(f, axis=rank - 1) + 1.0, pt.power(abs(g) + 0.5, pt.cast(i % 4, fdt)),
        ]
        return ins, outs, vals

    return build


for _s in range(12):
    case(f"dtype_fuzz_{_s}", rtol=2e-5)(_make3(_s))


# ---- fourth family: whole-tensor and keepdims forms, and the shape-changing ops, on the same random views ----
def _make4(seed):
    def build():
        rng = np.random.default_rng(17000 + seed)
        rank = int(rng.integers(2, 4))
        shape = _shape(rng, rank, budget=5000)
        a, xa, va = _operand(rng, "a", shape, "float64", allow_bcast=False)
        c, xc, vc = _operand(rng, "c", shape, "float64", allow_bcast=False)
        ins, vals = [xa, xc], {"a": va, "c": vc}
        ax = lambda: int(rng.integers(rank))  # noqa: E731
        outs = [
            pt.special.softmax(a, axis=None),
            pt.special.log_softmax(a * 0.5, axis=None),
            pt.mean(pt.sqr(a), axis=ax(), keepdims=True),
            pt.var(a, axis=ax(), keepdims=True) + 1.0,
            pt.std(c, axis=_axes(rng, rank)) + 1.0,
            pt.cumsum(pt.sqr(a) + 0.5, axis=None),
            pt.argmax(a, axis=None),
            pt.argmax(c, axis=ax(), keepdims=True),
            a - pt.max(a, axis=ax(), keepdims=True),
            pt.roll(a, int(rng.integers(-5, 6)), axis=ax()),
            pt.roll(c, int(rng.integers(1, 7))),
            pt.repeat(a, int(rng.integers(2, 4)), axis=ax()),
            pt.tile(c, tuple(int(rng.integers(1, 3)) for _ in range(rank))),
            pt.tril(a) if rank == 2 else pt.tril(a[0]),
            pt.triu(c, k=1) if rank == 2 else pt.triu(c[-1], k=-1),
            pt.outer(a.ravel()[:37], c.ravel()[1:20:2]),
            pt.swapaxes(a, 0, rank - 1) * 2.0,
            pt.flatten(c, 1) if rank > 1 else c,
            pt.stack([a, c], axis=ax()),
            pt.sum(pt.sqr(a), axis=ax(), keepdims=True) / (pt.sum(pt.sqr(a)) + 1.0),
        ]
        return ins, outs, vals

    return build


for _s in range(8):
    case(f"layout_fuzz4_{_s}", rtol=1e-10)(_make4(_s))
