"""CPU, build container only: the HipLinker boundary against the live reference."""
import json
import os

import numpy as np
import pytest

import make_ref

pytestmark = pytest.mark.skipif(not make_ref.available(), reason="/root/reference not present (GPU box)")


@pytest.fixture(scope="module")
def pt():
    make_ref.activate()
    import pytensor
    import pytensor.tensor as ptt

    import pytensor_amd

    pytensor_amd.register()
    return pytensor, ptt


def test_mode_hip_is_registered_and_lowers(pt):
    pytensor, ptt = pt
    from pytensor.compile.mode import get_mode, predefined_linkers

    from pytensor_amd.linker import HipLinker

    assert isinstance(get_mode("hip").linker, HipLinker)
    assert "hip" in predefined_linkers
    x = ptt.dvector("x")
    mu = ptt.dscalar("mu")
    y = ptt.exp(-0.5 * (x - mu) ** 2).sum()
    f = pytensor.function([x, mu], [y, pytensor.grad(y, x)], mode="hip")
    g = f.maker.linker.last_ir
    assert [n.op for n in g.nodes].count("Elemwise") == 1 and "CAReduce" in [n.op for n in g.nodes]
    assert not any(n.op == "HostPerform" for n in g.nodes)
    # the committed fixture is exactly what the linker produces today
    d = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "c1_gauss.json")))
    assert g.to_dict()["nodes"] == d["nodes"]


def test_thunk_fails_loudly_without_gpu(pt):
    pytensor, ptt = pt
    from pytensor_amd import ffi

    if ffi.device_count() > 0:
        pytest.skip("a GPU is visible")
    x = ptt.dvector("x")
    f = pytensor.function([x], [ptt.tanh(x).sum()], mode="hip")
    with pytest.raises(Exception) as ei:
        f(np.ones(4))
    assert "no HIP device" in str(ei.value) or "HipError" in repr(ei.type)


def test_linker_is_copyable_and_reusable(pt):
    # FunctionMaker does copy(mode.linker) and accept() on a linker bound to another fgraph
    # (pytensor/compile/maker.py:600-609, link/basic.py:308-317)
    import copy

    pytensor, ptt = pt
    from pytensor_amd.linker import HipLinker

    l = HipLinker()
    l2 = copy.copy(l)
    assert repr(l2) == "HipLinker()"
    x = ptt.fmatrix("x")
    f1 = pytensor.function([x], [x.sum(axis=0)], mode="hip")
    f2 = pytensor.function([x], [ptt.dot(x, x.T)], mode="hip")
    assert f1.maker.linker.last_ir.nodes[0].op == "CAReduce"
    assert any(n.op in ("Dot22", "Dot") for n in f2.maker.linker.last_ir.nodes)


def test_scan_and_ofg_inner_graphs_compile(pt):
    pytensor, ptt = pt
    xs = ptt.dmatrix("xs")
    s0 = ptt.dvector("s0")
    out = pytensor.scan(lambda x, s: ptt.tanh(s + x), sequences=[xs], outputs_info=[s0], return_updates=False)
    f = pytensor.function([xs, s0], [out[-1]], mode="hip")
    ops = [n.op for n in f.maker.linker.last_ir.nodes]
    assert "Scan" in ops
    from pytensor.tensor.special import softmax

    f = pytensor.function([xs], [softmax(xs, axis=-1)], mode="hip")  # SymbolicOp / OpFromGraph path
    assert not any(n.op == "HostPerform" for n in f.maker.linker.last_ir.nodes)


def test_widening_rows_lower_to_their_own_nodes(pt):
    """SURVEY §8f rows 3/4 and the index tier: each op keeps a node of its own in the IR (no
    HostPerform), OpFromGraph/SymbolicOps are inlined, unsupported scalar ops fail at compile
    time with the op's name."""
    pytensor, ptt = pt
    x, y = ptt.dvector("x"), ptt.dvector("y")
    M = ptt.dmatrix("M")
    rng = pytensor.shared(np.random.default_rng(3), name="rng")
    nr, u = ptt.random.uniform(0.0, 1.0, size=(8,), rng=rng).owner.outputs
    w, v = ptt.linalg.eigh(M)
    outs = [ptt.sort(x), ptt.argsort(x), x[x > 0], ptt.nonzero(x)[0], *pytensor.grad((ptt.concatenate([x, y]) ** 2).sum(), [x, y]),
            ptt.gammainc(ptt.abs(x) + 1, ptt.abs(y[: x.shape[0]])), u + x[:8], pytensor.grad((w * w).sum(), M), ptt.linalg.det(M),
            pytensor.grad(ptt.gammainc(ptt.abs(x) + 1.0, 2.0).sum(), x)]
    f = pytensor.function([x, y, M], outs, updates={rng: nr}, mode="hip", on_unused_input="ignore")
    ops = [n.op for n in f.maker.linker.last_ir.nodes]
    for name in ("SortOp", "ArgSortOp", "Nonzero", "Split", "RandomVariable", "Eigh", "Det"):
        assert name in ops, (name, ops)
    assert "HostPerform" not in ops and "OpFromGraph" not in ops and "AllocDiag" not in ops
    loops = [b for n in f.maker.linker.last_ir.nodes if "scalar" in n.params for b in n.params["scalar"]["body"] if b["op"] == "ScalarLoop"]
    assert loops, "the gradient of gammainc wrt its shape parameter is a ScalarLoop inside the fused kernel"
    with pytest.raises(NotImplementedError, match="Hyp2F1"):
        pytensor.function([x], ptt.hyp2f1(0.5, 1.0, 1.5, ptt.sigmoid(x)), mode="hip")


def test_every_random_variable_of_the_reference_lowers(pt):
    """tensor/random/basic.py: each RandomVariable class has a device sampler (no HostPerform); batched
    permutation / choice-without-replacement, which the reference loops over on the host, are refused."""
    pytensor, ptt = pt
    import re

    import pytensor.tensor.random.basic as rb
    from pytensor_amd.dispatch.random import DISTRIBUTIONS, STRUCTURED

    names = set(re.findall(r'^    name = "(\w+)"', open(rb.__file__).read(), re.M))
    assert names and names <= set(DISTRIBUTIONS) | set(STRUCTURED), names - set(DISTRIBUTIONS) - set(STRUCTURED)
    rng = pytensor.shared(np.random.default_rng(3), name="rng")
    x = ptt.dmatrix("x")
    R = ptt.random
    outs = [R.wald(1.5, 2.5, size=(5,), rng=rng), R.truncexpon(2.5, -0.5, 1.5, size=(5,), rng=rng), R.gengamma(3.0, 1.5, 0.8, size=(5,), rng=rng),
            R.betabinom(25, 2.0, 3.5, size=(5,), rng=rng), R.vonmises(0.7, 2.5, size=(5,), rng=rng), R.hypergeometric(30, 45, 20, size=(5,), rng=rng),
            R.multinomial(40, np.array([0.1, 0.2, 0.7]), size=(5,), rng=rng), R.permutation(10, rng=rng), R.permutation(x, rng=rng),
            R.choice(x, size=(2,), replace=False, rng=rng), R.choice(10, size=(2, 2), replace=False, p=np.full(10, 0.1), rng=rng)]
    f = pytensor.function([x], outs, mode="hip")
    rvs = [n.params["name"] for n in f.maker.linker.last_ir.nodes if n.op == "RandomVariable"]
    assert sorted(rvs) == sorted(["wald", "truncexpon", "gengamma", "beta_binomial", "vonmises", "hypergeometric", "multinomial",
                                  "permutation", "permutation", "choice_without_replacement", "choice_without_replacement"])
    assert not any(n.op == "HostPerform" for n in f.maker.linker.last_ir.nodes)
    x3 = ptt.dtensor3("x3")
    with pytest.raises(NotImplementedError, match="permutation"):
        pytensor.function([x3], rb.PermutationRV(signature="(x)->(x)", dtype="float64")(x3[0], rng=rng, size=(2,), return_next_rng=True)[1], mode="hip")


def test_library_functions_lower_whole_and_match_the_c_linker(pt):
    """pt.einsum / pad / interpolate / the linalg helpers are graphs of Ops (or OpFromGraphs): they must
    come through the hip linker without a host node, and the lowered graph — run by the oracle — must
    reproduce the reference's C linker."""
    pytensor, ptt = pt
    import np_graph
    from pytensor.tensor.interpolate import interp

    x, y, v, t3 = ptt.dmatrix("x"), ptt.dmatrix("y"), ptt.dvector("v"), ptt.dtensor3("t3")
    L = ptt.linalg
    outs = [
        ptt.einsum("ij,jk->ik", x, y), ptt.einsum("bij,jk->bik", t3, y), ptt.einsum("ij,ij->", x, x),
        ptt.pad(x, ((1, 2), (0, 3)), mode="constant", constant_values=2.0), ptt.pad(x, 2, mode="edge"), ptt.pad(x, 1, mode="reflect"),
        ptt.pad(x, 1, mode="wrap"), ptt.pad(x, 1, mode="mean"), interp(v, ptt.as_tensor(np.linspace(0, 1, 7)), ptt.as_tensor(np.linspace(0, 1, 7) ** 2)),
        ptt.tril(x), ptt.diff(v), ptt.roll(x, 2, axis=1), ptt.median(v), ptt.logsumexp(x, axis=1), ptt.bincount(ptt.cast(ptt.abs(v) * 3, "int64")),
        ptt.tile(v, (2, 3)), ptt.tensordot(x, y, axes=[[1], [0]]), ptt.ptp(x, axis=0), ptt.searchsorted(ptt.sort(v), v), ptt.unique(ptt.round(v)),
        ptt.repeat(v, 2), L.kron(x[:2, :2], y[:2, :2]), L.norm(x), L.matrix_power(x, 3), L.pinv(y), L.svd(y)[1], L.qr(y)[1], L.lstsq(y, v[:5], -1.0)[0],
        L.slogdet(x)[1], L.block_diag(x, y), L.lu_solve(L.lu_factor(x), v[:5]), L.expm(x * 0.1), L.solve_continuous_lyapunov(x - 4 * ptt.eye(5), x @ x.T),
    ]
    rng = np.random.default_rng(0)
    vals = {"x": rng.normal(size=(5, 5)), "y": rng.normal(size=(5, 4)), "v": rng.normal(size=6), "t3": rng.normal(size=(2, 3, 5))}
    ins = [x, y, v, t3]
    f = pytensor.function(ins, outs, mode="hip", on_unused_input="ignore")
    g = f.maker.linker.last_ir
    assert not [n.params.get("name") for n in g.nodes if n.op == "HostPerform"]
    want = pytensor.function(ins, outs, mode="CVM", on_unused_input="ignore")(*[vals[i.name] for i in ins])
    got = np_graph.run_graph(g, [vals[i.name] for i in f.maker.fgraph.inputs])
    for k, (a, b) in enumerate(zip(got, want)):
        np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-11, err_msg=f"output {k}")


def test_op_coverage_of_the_registry(pt):
    """Every ``Op`` class of pytensor.tensor (+ ifelse, scan, compile.ops, raise_op) either has a
    ``hip_funcify`` registration or is on this list with the reason it has none."""
    import importlib
    import pkgutil

    import pytensor
    from pytensor.graph.op import Op

    from pytensor_amd.lower import hip_funcify

    not_lowered = {
        # abstract bases
        "BaseBLAS", "GemmRelated", "BaseBlockDiagonal", "SolveBase", "ScipyWrapperOp", "ScipyScalarWrapperOp",
        "ScipyVectorWrapperOp", "RNGConsumerOp", "AbstractRNGConstructor",
        # complex-valued results (complex dtypes are a compile-time NotImplementedError)
        "Eig", "Schur", "QZ", "TRSYL", "Fourier",
        # call back into Python / construct host objects
        "MinimizeScalarOp", "MinimizeOp", "RootScalarOp", "RootOp", "FromFunctionOp", "DefaultGeneratorMakerOp", "MakeSlice",
    }
    mods = []
    for m in pkgutil.walk_packages(pytensor.tensor.__path__, "pytensor.tensor."):
        if ".rewriting" in m.name or "xtensor" in m.name:
            continue
        try:
            mods.append(importlib.import_module(m.name))
        except Exception:
            pass
    mods += [importlib.import_module(n) for n in ("pytensor.ifelse", "pytensor.raise_op", "pytensor.compile.ops", "pytensor.scan.op", "pytensor.compile.builders")]
    default = hip_funcify.dispatch(object)
    seen, missing = set(), set()
    for m in mods:
        for name, c in list(vars(m).items()):
            try:
                is_op = isinstance(c, type) and issubclass(c, Op)
            except TypeError:  # (typing generics)
                is_op = False
            if is_op and c.__module__ == m.__name__ and c not in seen:
                seen.add(c)
                if hip_funcify.dispatch(c) is default:
                    missing.add(name)
    assert len(seen) >= 160
    assert missing == not_lowered, (sorted(missing - not_lowered), sorted(not_lowered - missing))


def test_all_reduce_op_lowers_and_differentiates(pt):
    """The explicit collective (north_star: "the rare explicit all-reduce Op"): an ordinary Op for
    every linker (perform = the host reduction, identity on one rank), an ``AllReduce`` IR node
    under ``mode="hip"``, with a gradient."""
    pytensor, ptt = pt
    from pytensor_amd.collective import AllReduce, all_reduce

    x = ptt.dvector("x")
    logp_shard = -0.5 * (x**2).sum()
    total = all_reduce(logp_shard)  # sum of the per-rank shards
    cost = ptt.exp(0.1 * total)
    gx = pytensor.grad(cost, x)
    assert any(isinstance(n.op, AllReduce) for n in pytensor.graph.traversal.applys_between([x], [gx]))
    xv = np.linspace(-1, 1, 7)
    f_py = pytensor.function([x], [total, cost, gx], mode=pytensor.compile.mode.Mode("py", "fast_run"))
    t, c, g = f_py(xv)
    np.testing.assert_allclose(t, -0.5 * (xv**2).sum(), rtol=1e-14)  # one rank: identity
    np.testing.assert_allclose(g, np.exp(0.1 * t) * 0.1 * (-xv), rtol=1e-13)
    f = pytensor.function([x], [total, cost, gx], mode="hip")
    ir = f.maker.linker.last_ir
    ops = [n.op for n in ir.nodes]
    assert ops.count("AllReduce") == 2 and "HostPerform" not in ops  # forward + the gradient's
    assert all(n.params == {"op": "sum"} for n in ir.nodes if n.op == "AllReduce")
    # the oracle interprets the lowered graph (single process: identity) like the py linker
    import np_graph

    for a, b in zip(np_graph.run_graph(ir, [xv]), (t, c, g)):
        np.testing.assert_allclose(a, b, rtol=1e-13)
    with pytest.raises(ValueError):
        all_reduce(x, "mean")
