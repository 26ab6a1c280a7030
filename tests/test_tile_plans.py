"""CPU: the host-side planning of the tiled N-d kernels (dispatch/elemwise.py ``tile_plan`` / ``reduce_plan``) — the part
of the broadcasting ``Elemwise`` loop and of ``CAReduce`` over an axis tuple that is not HIP: which dimension the tile
rows walk, operand classes, pack width, how kept / reduced dimensions are sorted and merged, the output strides.

The kernels' addressing is restated here in NumPy from the plan alone (every tile visit of every workgroup, the same
formulas as codegen_tile.py emits) and run on small arrays: each output element must be written exactly once
(Elemwise) / must receive exactly the terms NumPy's reduction gives it (CAReduce), for the layouts of the reference's
own benchmarks (tests/benchmarks/test_careduce.py:7-61: C-contiguous, transposed (2,0,1), strided [::2] + transposed;
seven axis sets) and for the operand classes of tests/golden/ew_*.
"""
import itertools

import numpy as np
import pytest

from pytensor_amd.dispatch.elemwise import BLOCK, _collapse, _cstrides, reduce_plan, tile_plan


def _estrides(a):
    return tuple(s // a.itemsize for s in a.strides)


def _run_reduce_plan(x, axes):
    """sum over `axes` of the strided view x, computed the way tile_reduce_source addresses memory"""
    base = x.base if x.base is not None else x
    flat = np.ascontiguousarray(base).ravel() if base.flags.c_contiguous else None
    assert flat is not None
    off0 = (x.__array_interface__["data"][0] - base.__array_interface__["data"][0]) // x.itemsize
    out_shape = tuple(s for d, s in enumerate(x.shape) if d not in axes)
    plan = reduce_plan(tuple(x.shape), axes, [_estrides(x)], [str(x.dtype)], [0], out_shape)
    assert plan is not None
    TX, RPT, V = plan["TX"], plan["RPT"], plan["V"]
    TY = BLOCK // TX
    TC, TR = TX * V, TY * RPT
    R, D, row, inner, kb, rd = plan["R"], plan["D"], plan["row"], plan["inner"], plan["kb"], plan["rd"]
    out = np.zeros(plan["n_out"])
    hits = np.zeros(x.size, dtype=int) if x.size == np.unique(np.lib.stride_tricks.as_strided(np.arange(base.size), x.shape, tuple(s * 8 for s in _estrides(x)))).size else None
    rows = np.arange(plan["nrb"] * TR)
    cols = np.arange(plan["ncb"] * TC)
    rows, cols = rows[rows < R], cols[cols < D]
    for kc in itertools.product(*[range(d["n"]) for d in kb]):
        for rc in itertools.product(*[range(d["n"]) for d in rd]):
            boff = sum(c * d["st"][0] for c, d in zip(kc, kb)) + sum(c * d["st"][0] for c, d in zip(rc, rd))
            oo = sum(c * d["ost"] for c, d in zip(kc, kb))
            src = off0 + boff + rows[:, None] * (row["st"][0] if row is not None else 0) + cols[None, :] * inner["st"][0]
            dst = oo + (rows[:, None] * row["ost"] if plan["row_kept"] else 0) + (cols[None, :] * inner["ost"] if plan["inner_kept"] else 0)
            np.add.at(out, np.broadcast_to(dst, src.shape).ravel(), flat[src.ravel()])
    return out.reshape(out_shape), plan


@pytest.mark.parametrize("layout", ["c_contiguous", "transposed", "strided"])
@pytest.mark.parametrize("axes", [(0,), (1,), (2,), (0, 1), (0, 2), (1, 2), (0, 1, 2)])
@pytest.mark.parametrize("shape", [(5, 6, 7), (33, 70, 129), (256, 4, 300)])
def test_reduce_plan_reaches_every_term_once(layout, axes, shape):
    rng = np.random.default_rng(abs(hash((layout, axes, shape))) % 2**31)
    if layout == "strided":
        x = rng.uniform(size=(2 * shape[0], *shape[1:]))[::2].transpose(2, 0, 1)
    else:
        x = rng.uniform(size=shape)
        if layout == "transposed":
            x = x.transpose(2, 0, 1)
    got, plan = _run_reduce_plan(x, list(axes))
    np.testing.assert_allclose(got, x.sum(axis=axes), rtol=1e-13)
    # the tile never walks a dimension at a stride when the operand has a unit-stride one to offer
    assert abs(plan["inner"]["st"][0]) == 1


def test_reduce_plan_merges_a_transposed_view_back_to_memory_order():
    x = np.zeros((8, 9, 10)).transpose(2, 0, 1)  # strides (1, 90, 10)
    plan = reduce_plan(x.shape, [0, 1, 2], [_estrides(x)], ["float64"], [0], ())
    # all reduced: sorted by stride and merged into ONE contiguous dimension of 720 elements
    assert plan["D"] == 720 and plan["row"] is None and not plan["kb"] and not plan["rd"] and plan["cls"] == "V"
    plan = reduce_plan(x.shape, [1], [_estrides(x)], ["float64"], [0], (10, 9))
    # axis 1 of the view = the OUTER dimension in memory: the inner dimension stays the contiguous (kept) one
    assert plan["inner_kept"] and plan["inner"]["n"] == 10 and plan["inner"]["st"] == [1] and plan["inner"]["ost"] == 9
    assert not plan["row_kept"] and plan["row"]["n"] == 8 and plan["row"]["st"] == [90]


def test_reduce_plan_vector_width_follows_alignment():
    n = 64
    st = (n, 1)
    assert reduce_plan((n, n), [1], [st], ["float64"], [0], (n,))["V"] == 2
    assert reduce_plan((n, n), [1], [st], ["float64"], [8], (n,))["V"] == 1  # base pointer not 16-byte aligned
    assert reduce_plan((n, n + 1), [1], [(n + 1, 1)], ["float64"], [0], (n,))["V"] == 1  # odd rows: packs would straddle
    assert reduce_plan((n, n), [1], [st], ["float32"], [0], (n,))["V"] == 4


def _run_tile_plan(out_shape, operands):
    """elementwise sum of the (strided, broadcast) operands, addressed the way tile_kernel_source does"""
    strides = []
    for a in operands:
        st = _estrides(a)
        strides.append(tuple(0 if a.shape[d] == 1 and out_shape[d] != 1 else st[d] for d in range(len(out_shape))))
    cshape, cstr = _collapse(tuple(out_shape), strides + [_cstrides(out_shape)])
    cstr = cstr[:-1]
    plan = tile_plan(cshape, cstr, [str(a.dtype) for a in operands], ["float64"], [0] * len(operands), [0])
    jr, batch = plan["jr"], plan["batch"]
    ocs = _cstrides(cshape)
    n = int(np.prod(out_shape))
    out = np.zeros(n)
    written = np.zeros(n, dtype=int)
    flats, offs = [], []
    for a in operands:
        base = a.base if a.base is not None else a
        flats.append(np.ascontiguousarray(base).ravel())
        offs.append((a.__array_interface__["data"][0] - base.__array_interface__["data"][0]) // a.itemsize)
    R, D = plan["R"], plan["D"]
    rows, cols = np.arange(R), np.arange(D)
    for bc in itertools.product(*[range(cshape[j]) for j in batch]):
        dst = sum(c * ocs[j] for c, j in zip(bc, batch)) + rows[:, None] * (ocs[jr] if jr is not None else 0) + cols[None, :]
        acc = np.zeros(dst.shape)
        for f, o, st in zip(flats, offs, cstr):
            src = o + sum(c * st[j] for c, j in zip(bc, batch)) + rows[:, None] * (st[jr] if jr is not None else 0) + cols[None, :] * st[-1]
            acc = acc + f[src]
        out[dst.ravel()] = acc.ravel()
        np.add.at(written, dst.ravel(), 1)
    assert (written == 1).all(), "every output element exactly once"
    return out.reshape(out_shape), plan


def test_tile_plan_operand_classes():
    rng = np.random.default_rng(0)
    A, r, c = rng.normal(size=(37, 53)), rng.normal(size=(1, 53)), rng.normal(size=(37, 1))
    got, plan = _run_tile_plan((37, 53), [A, r, c])
    np.testing.assert_allclose(got, A + r + c, rtol=1e-15)
    assert plan["cls"] == "VRB"  # streamed / row vector held in registers / one scalar per row
    B = rng.normal(size=(53, 37))
    got, plan = _run_tile_plan((37, 53), [A, B.T])
    np.testing.assert_allclose(got, A + B.T, rtol=1e-15)
    assert plan["cls"] == "VT" and plan["lds_rows"] == 64 and plan["TX"] * plan["V"] == 64  # the transposed operand through LDS
    x = rng.normal(size=(40, 12))
    got, plan = _run_tile_plan((40, 12), [x, x[:, ::-1]])
    np.testing.assert_allclose(got, x + x[:, ::-1], rtol=1e-15)
    assert plan["cls"] == "VG"  # reversed: scalar loads at the operand's own stride
    T = rng.normal(size=(6, 10, 21))
    got, plan = _run_tile_plan((10, 6, 21), [T.transpose(1, 0, 2)])
    np.testing.assert_allclose(got, T.transpose(1, 0, 2), rtol=0)
    assert plan["cls"] == "V" and len(plan["batch"]) == 1  # outer dimensions swapped: still packs along the inner one
    a, b = rng.normal(size=(5, 1, 7, 1)), rng.normal(size=(1, 4, 1, 9))
    got, plan = _run_tile_plan((5, 4, 7, 9), [a, b])
    np.testing.assert_allclose(got, a + b, rtol=1e-15)


def test_tile_plan_small_inner_dimension_keeps_lanes_on_adjacent_rows():
    S, s = np.zeros((1000, 10)), np.zeros((1, 10))
    _, plan = _run_tile_plan((1000, 10), [S, s])
    # 10 columns: 5 two-element packs -> 8 lanes per row, 32 thread rows: a wave covers 8 consecutive rows of memory
    assert plan["V"] == 2 and plan["TX"] == 8 and plan["cls"] == "VR"
