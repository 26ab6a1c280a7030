"""GPU: the error convention of the boundary: invalid inputs raise the Python
exception class the reference raises (ValueError for shape / runtime-broadcast violations,
IndexError for out-of-range indices, the CheckAndRaise exception type), numerical failure of
Cholesky is NaN-fill and not an error, and a frozen plan refuses a changed signature."""
import numpy as np
import pytest

import np_graph
from util import load_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from pytensor_amd import ffi

    if ffi.device_count() <= 0:
        pytest.fail("no HIP device visible: GPU tests must run on the MI355X box")
    ffi.init(0)
    return ffi


def _both_raise(g, ins, exc):
    """the device path and the oracle (= the reference's perform semantics) agree on the class"""
    from pytensor_amd.executor import HipExecutable

    with pytest.raises(exc):
        np_graph.run_graph(g, ins)
    with pytest.raises(exc):
        HipExecutable(g)(*ins)
    # and the executable is still usable afterwards
    return HipExecutable(g)


def test_elemwise_shape_mismatch_is_value_error(hip):
    g, ins, cvm, *_ = load_case("elemwise_bcast")
    bad = list(ins)
    bad[0] = np.ascontiguousarray(ins[0][:, :4])  # a: (7, 4) against b: (1, 5)
    exe = _both_raise(g, bad, ValueError)
    for a, b in zip(exe(*ins), cvm):
        np.testing.assert_allclose(a, b, rtol=1e-12)


def test_runtime_broadcast_is_value_error(hip):
    # a (None, None) operand of length 1 along a non-broadcastable dim (elemwise.py:825-840)
    g, ins, *_ = load_case("elemwise_bcast")
    bad = list(ins)
    bad[0] = ins[0][:1]  # a: (1, 5) while c: (7, 1) makes the output 7 rows
    _both_raise(g, bad, ValueError)


def test_gemv_shape_mismatch_is_value_error(hip):
    g, ins, *_ = load_case("c3_gemv")
    bad = list(ins)
    k = [i for i, a in enumerate(ins) if a.ndim == 1][0]
    bad[k] = ins[k][:-1]
    _both_raise(g, bad, ValueError)


def test_out_of_range_index_is_index_error(hip):
    g, ins, *_ = load_case("indexing")
    bad = list(ins)
    k = [i for i, a in enumerate(ins) if a.dtype.kind == "i" and a.ndim == 1][0]
    idx = ins[k].copy()
    idx[1] = 10_000
    bad[k] = idx
    _both_raise(g, bad, IndexError)


def test_out_of_range_index_inside_fused_gather_is_index_error(hip):
    # the gather is read inside the generated elementwise kernel (gatherfuse.py): the bounds
    # check is the device error flag, surfaced as IndexError by the executor
    from pytensor_amd.executor import HipExecutable

    g, ins, cvm, *_ = load_case("gather_elemwise")
    assert any(n.params.get("gather") for n in HipExecutable(g).graph.nodes)
    bad = list(ins)
    k = [i for i, a in enumerate(ins) if a.dtype.kind == "i"][0]
    idx = ins[k].copy()
    idx[7] = 19  # table length is 19: one past the end
    bad[k] = idx
    exe = _both_raise(g, bad, IndexError)
    for a, b in zip(exe(*ins), cvm):  # the flag was cleared: the next valid call is clean
        np.testing.assert_allclose(a, b, rtol=1e-12)
    # a frozen plan learns about it from the error word packed next to its results
    plan = exe.freeze(*ins)
    with pytest.raises(IndexError):
        plan(*bad)
    for a, b in zip(plan(*ins), cvm):
        np.testing.assert_allclose(a, b, rtol=1e-12)
    plan.close()


def test_singular_matrix_conventions(hip):
    # np.linalg.inv raises LinAlgError; Solve NaN-fills (general.py:74-75); det is exactly 0
    from pytensor_amd.executor import HipExecutable
    from pytensor_amd.ir import Graph

    def unary(op, nout=1):
        g = Graph(name=op)
        a = g.new_var("float64", (None, None), name="A")
        outs = [g.new_var("float64", (None, None) if op == "MatrixInverse" else ()) for _ in range(nout)]
        g.add_node(op, {}, [a], outs)
        g.inputs, g.outputs = [a], outs
        return g

    S = np.array([[1.0, 2.0, 3.0], [2.0, 4.0, 6.0], [0.5, -1.0, 2.0]])
    with pytest.raises(np.linalg.LinAlgError):
        np_graph.run_graph(unary("MatrixInverse"), [S])
    exe = HipExecutable(unary("MatrixInverse"))
    with pytest.raises(np.linalg.LinAlgError):
        exe(S)
    good = S + np.eye(3)
    np.testing.assert_allclose(exe(good)[0], np.linalg.inv(good), rtol=1e-12)  # flag cleared
    assert HipExecutable(unary("Det"))(S)[0] == 0.0
    sign, logabs = HipExecutable(unary("SLogDet", 2))(S)
    assert sign == 0.0 and logabs == -np.inf
    g = Graph(name="solve")
    a = g.new_var("float64", (None, None), name="A")
    b = g.new_var("float64", (None,), name="b")
    x = g.new_var("float64", (None,))
    g.add_node("Solve", {"assume_a": "gen", "lower": False, "b_ndim": 1}, [a, b], [x])
    g.inputs, g.outputs = [a, b], [x]
    assert np.isnan(HipExecutable(g)(S, np.ones(3))[0]).all()
    assert np.isnan(np_graph.run_graph(g, [S, np.ones(3)])[0]).all()


def test_check_and_raise_type(hip):
    # C4 asserts y.shape[0] == X.shape[0] (CheckAndRaise nodes of the Gemv shape checks)
    from pytensor_amd.executor import HipExecutable

    g, ins, *_ = load_case("c4_hier_small")
    names = [g.vars[v].name for v in g.inputs]
    bad = list(ins)
    k = names.index("y")
    bad[k] = ins[k][:-3]
    with pytest.raises(Exception) as e_or:
        np_graph.run_graph(g, bad)
    with pytest.raises(Exception) as e_hip:
        HipExecutable(g)(*bad)
    assert type(e_hip.value) is type(e_or.value)


def test_indefinite_cholesky_is_nan_not_an_error(hip):
    from pytensor_amd.executor import HipExecutable

    g, ins, cvm, *_ = load_case("cholesky_indefinite")
    out = HipExecutable(g)(*ins)
    assert all(np.isnan(o).all() for o in out) and all(np.isnan(c).all() for c in cvm)


def test_frozen_plan_refuses_changed_signature(hip):
    from pytensor_amd.executor import HipExecutable

    g, ins, *_ = load_case("elemwise_bcast")
    exe = HipExecutable(g)
    exe(*ins)
    plan = exe.freeze(*ins)
    bad = list(ins)
    bad[0] = np.ascontiguousarray(ins[0][:, :4])
    with pytest.raises(TypeError):
        plan(*bad)
    plan.close()
