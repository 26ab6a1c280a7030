"""GPU: the generated skinny-product + epilogue kernel (``DotEpilogue``/``PackB16``) against the
oracle, node level — shapes with row/column tails, broadcast operands, two products in one
launch, fp32 and fp64, and the layouts that must take the GEMM + Elemwise fallback.

Reference semantics: ``Dot22`` (pytensor/tensor/blas/gemm.py:248-275) followed by ``Elemwise``
(pytensor/tensor/elemwise.py:755-823); tolerances are north_star's (fp64 rtol 1e-12, fp32 1e-5)
plus the summation-order bound of a K-term dot product (tests/test_gpu_fullsize.py docstring).
"""
import numpy as np
import pytest

import bounds
import np_graph
from pytensor_amd.ir import Graph

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from pytensor_amd import ffi

    if ffi.device_count() <= 0:
        pytest.fail("no HIP device visible: GPU tests must run on the MI355X box")
    ffi.init(0)
    return ffi


def _gate_body(dt, two):
    """one product: sigmoid(x + d0 + b) * h ; two: (1-s)*h + s*tanh(x + d0 + b) with s = sigmoid(d1)"""
    if not two:
        body = [
            {"op": "Add", "in": [["i", 1], ["i", 2]], "dtype": dt},
            {"op": "Add", "in": [["t", 0], ["i", 3]], "dtype": dt},
            {"op": "Sigmoid", "in": [["t", 1]], "dtype": dt},
            {"op": "Mul", "in": [["t", 2], ["i", 0]], "dtype": dt},
        ]
        return {"in_dtypes": [dt] * 4, "out_dtypes": [dt, dt], "body": body, "outs": [["t", 3], ["t", 1]]}
    body = [
        {"op": "Add", "in": [["i", 1], ["i", 2]], "dtype": dt},
        {"op": "Add", "in": [["t", 0], ["i", 3]], "dtype": dt},
        {"op": "Tanh", "in": [["t", 1]], "dtype": dt},
        {"op": "Sigmoid", "in": [["i", 4]], "dtype": dt},
        {"op": "Mul", "in": [["t", 3], ["t", 2]], "dtype": dt},
        {"op": "Sub", "in": [["c", "0x1.0000000000000p+0", dt], ["t", 3]], "dtype": dt},
        {"op": "Mul", "in": [["t", 5], ["i", 0]], "dtype": dt},
        {"op": "Add", "in": [["t", 6], ["t", 4]], "dtype": dt},
    ]
    return {"in_dtypes": [dt] * 5, "out_dtypes": [dt], "body": body, "outs": [["t", 7]]}


def _graph(dt, two, bias_shape, same_left=False):
    g = Graph(name="dotew_unit")
    h = g.new_var(dt, (None, None), name="h")
    x = g.new_var(dt, (None, None), name="x")
    A = g.new_var(dt, (None, None), name="A")
    b = g.new_var(dt, bias_shape, name="b")
    W = g.new_var(dt, (None, None), name="W")
    Wp = g.new_var(dt, (None,))
    g.add_node("PackB16", {}, [W], [Wp])
    ins, extra, dpos = [h, x, A, b], [W, Wp], [2]
    g.inputs = [h, x, A, b, W]
    if two:
        A2 = A if same_left else g.new_var(dt, (None, None), name="A2")
        W2 = g.new_var(dt, (None, None), name="W2")
        W2p = g.new_var(dt, (None,))
        g.add_node("PackB16", {}, [W2], [W2p])
        ins, extra, dpos = ins + [A2], extra + [W2, W2p], [2, 4]
        g.inputs += [W2] if same_left else [A2, W2]
    body = _gate_body(dt, two)
    outs = [g.new_var(dt, (None, None)) for _ in body["out_dtypes"]]
    g.add_node("DotEpilogue", {"scalar": body, "dot_inputs": dpos}, ins + extra, outs)
    g.outputs = outs
    return g


CASES = [
    # dt, M, K, N, two products, bias shape, A as a strided view
    ("float32", 64, 1024, 1024, False, (1, None), False),
    ("float32", 64, 1024, 1024, True, (1, None), False),
    ("float64", 5, 48, 40, False, (1, None), False),     # row and column tails
    ("float64", 70, 32, 24, True, (None, 1), False),     # column-broadcast operand, two products
    ("float32", 16, 16, 16, False, (1, 1), False),
    ("float64", 33, 80, 17, False, (1, None), True),     # A rows 16-byte aligned inside a wider buffer
    ("float32", 7, 24, 9, False, (1, None), False),      # K % 16 != 0: GEMM + Elemwise path
    ("float64", 3, 37, 5, True, (1, None), False),       # same, two products
    ("float32", 64, 1024, 1024, True, (1, None), "same_left"),  # h@U_r and h@U_z: the left operand streamed once
    ("float64", 21, 48, 33, True, (1, None), "same_left"),
]


@pytest.mark.parametrize("dt,M,K,N,two,bshape,strided", CASES)
def test_dot_epilogue_matches_oracle(hip, dt, M, K, N, two, bshape, strided):
    from pytensor_amd.executor import HipExecutable

    rng = np.random.default_rng(M * 131 + K * 7 + N)
    same_left = strided == "same_left"
    strided = strided is True
    g = _graph(dt, two, bshape, same_left)
    h = rng.normal(size=(M, N)).astype(dt)
    x = rng.normal(size=(M, N)).astype(dt)
    if strided:
        A = rng.normal(size=(M, K + 16)).astype(dt)[:, 8 : 8 + K]
    else:
        A = rng.normal(size=(M, K)).astype(dt)
    b = rng.normal(size=tuple(1 if s == 1 else (N if k == 1 else M) for k, s in enumerate(bshape))).astype(dt)
    W = (rng.normal(size=(K, N)) / np.sqrt(K)).astype(dt)
    ins = [h, x, A, b, W]
    if two and same_left:
        ins += [(rng.normal(size=(K, N)) / np.sqrt(K)).astype(dt)]
    elif two:
        ins += [rng.normal(size=(M, K)).astype(dt), (rng.normal(size=(K, N)) / np.sqrt(K)).astype(dt)]
    want = np_graph.run_graph(g, ins)
    exe = HipExecutable(g, fuse=False)
    got = exe(*ins)
    eps = bounds.EPS32 if dt == "float32" else bounds.EPS64
    rtol = 1e-5 if dt == "float32" else 1e-12
    # |d(dot)| <= C * eps * sum_k |a_k w_k|; every epilogue here is 1-Lipschitz in the product
    # up to the factor |h| <= ~5 of the final multiply
    absdot = np.abs(A) @ np.abs(W)
    if two:
        absdot = absdot + (np.abs(A) @ np.abs(ins[5]) if same_left else np.abs(ins[5]) @ np.abs(ins[6]))
    atol = bounds.C_SUM * eps * absdot * (1.0 + np.abs(h))
    for a, w in zip(got, want):
        assert a.shape == w.shape and a.dtype == w.dtype
        err = np.abs(a.astype(np.float64) - w.astype(np.float64))
        assert np.all(err <= atol + rtol * np.abs(w)), float(np.max(err / (atol + rtol * np.abs(w))))
    # deterministic: the same bits on a second run
    for a, a2 in zip(got, exe(*ins)):
        np.testing.assert_array_equal(a, a2)


def test_pack_b16_layout_bit_exact(hip):
    """The device repack against the oracle's restatement of the layout (bit-exact tier)."""
    from pytensor_amd.executor import HipExecutable

    for dt, K, N in (("float32", 35, 21), ("float64", 64, 48), ("float32", 16, 16)):
        g = Graph(name="pack")
        W = g.new_var(dt, (None, None))
        Wp = g.new_var(dt, (None,))
        g.add_node("PackB16", {}, [W], [Wp])
        g.inputs, g.outputs = [W], [Wp]
        w = np.random.default_rng(K + N).normal(size=(N, K)).astype(dt).T  # column-major source
        (got,) = HipExecutable(g, fuse=False)(w)
        (want,) = np_graph.run_graph(g, [w])
        np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("B,H,T", [(64, 128, 7), (24, 64, 5), (5, 48, 4)])
def test_scan_steps_with_packed_left_operands_are_bit_identical(hip, monkeypatch, B, H, T):
    """Inside a Scan the step kernels hand each other their left operands in the MFMA operand order
    (``r*h`` within the step, ``h`` from the previous step: dispatch/scan.py::_pack_plan).  The packed
    image holds the same values, so the MFMA chains are the same: the GRU trajectory must be
    BIT-identical with the hand-over switched off (``PTHIP_DOTEW_PACKA=0``), eager and replayed,
    for batches that fill the 16-row tiles and batches that do not."""
    import json
    import os

    from pytensor_amd import configs
    from pytensor_amd.executor import HipExecutable

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = json.load(open(os.path.join(root, "tests", "golden", "c5_gru.json")))
    g = Graph.from_dict(d)
    v = configs.c5_inputs(T=T, B=B, H=H, seed=50 + B)
    v["h0"] = (0.1 * np.random.default_rng(B).normal(size=(B, H))).astype("float32")
    ins = [v[n] for n in d["input_names"]]
    monkeypatch.setenv("PTHIP_DOTEW_PACKA", "1")
    exe = HipExecutable(g)
    got = exe(*ins)
    got_replay = exe.freeze(*ins)(*ins)
    monkeypatch.setenv("PTHIP_DOTEW_PACKA", "0")
    plain = HipExecutable(g)(*ins)
    for a, b, c in zip(got, got_replay, plain):
        np.testing.assert_array_equal(a, c)
        np.testing.assert_array_equal(b, c)
    want = np_graph.run_graph(g, ins)
    for a, b in zip(got, want):
        np.testing.assert_allclose(a, b, rtol=1e-5, atol=2e-5)


def test_scan_overlapped_hoisted_products_are_bit_identical(hip, monkeypatch):
    """The hoisted sequence products (``x_t @ W`` for all t) of a long Scan are computed chunk by
    chunk on a second stream while the steps run (dispatch/blas.py::LazySeq, dispatch/scan.py).
    Opt-in (measured: no gain on this chip, see dispatch/blas.py::LazySeq) but kept correct: row
    chunks of a GEMM are the same per-element fma chains, so results are BIT-identical with the
    overlap switched off (the default: one product before the loop), eager and replayed; the
    replayed plan is exercised twice (the fork/join must survive re-launching)."""
    import json
    import os

    from pytensor_amd import configs
    from pytensor_amd.dispatch.blas import LazySeq  # noqa: F401  (the class under test exists)
    from pytensor_amd.executor import HipExecutable

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = json.load(open(os.path.join(root, "tests", "golden", "c5_gru.json")))
    g = Graph.from_dict(d)
    v = configs.c5_inputs(T=230, B=64, H=1024, seed=77)  # 3.6 chunks of 64 steps: a ragged last chunk
    ins = [v[n] for n in d["input_names"]]
    monkeypatch.setenv("PTHIP_SCAN_OVERLAP", "1")
    exe = HipExecutable(g, resident=range(len(ins)))
    got = exe(*ins)
    plan = exe.freeze(*ins)
    r1, r2 = plan(*ins), plan(*ins)
    monkeypatch.setenv("PTHIP_SCAN_OVERLAP", "0")
    plain = HipExecutable(g, resident=range(len(ins)))(*ins)
    for a, b, c, e in zip(got, r1, r2, plain):
        np.testing.assert_array_equal(a, e)
        np.testing.assert_array_equal(b, e)
        np.testing.assert_array_equal(c, e)
