"""CPU: the NumPy oracle against the reference's own outputs (golden vectors)."""
import numpy as np
import pytest

import np_graph
from util import assert_parity, golden_cases, load_case


@pytest.mark.parametrize("name", golden_cases())
def test_oracle_matches_reference_c_linker(name):
    g, ins, cvm, py, meta = load_case(name)
    out = np_graph.run_graph(g, ins)
    for k, (a, b) in enumerate(zip(out, cvm)):
        assert_parity(a, b, None, f"{name} out{k} (oracle vs reference C linker)", case=name, k=k, py=py[k])


@pytest.mark.parametrize("name", golden_cases())
def test_reference_py_and_c_linkers_agree(name):
    # sanity of the fixtures themselves: NumPy linker vs C linker of the reference
    g, ins, cvm, py, meta = load_case(name)
    for k, (a, b) in enumerate(zip(py, cvm)):
        # (py_rtol: cases where the reference's SciPy-backed impl and its C support code are
        #  different algorithms — the C values are the parity target, this is only a sanity check)
        assert_parity(a, b, meta.get("py_rtol", max(meta["rtol"], 1e-10)), f"{name} out{k} (reference py vs C)")


def test_ir_roundtrip():
    from pytensor_amd.ir import Graph

    for name in golden_cases():
        g, ins, cvm, py, meta = load_case(name)
        g2 = Graph.from_json(g.to_json())
        out = np_graph.run_graph(g2, ins)
        for k, (a, b) in enumerate(zip(out, cvm)):
            assert_parity(a, b, None, name, case=name, k=k, py=py[k])


def test_runtime_broadcast_is_an_error():
    # Elemwise._check_runtime_broadcast (pytensor/tensor/elemwise.py:825-840)
    g, ins, cvm, py, meta = load_case("c2_cheap")
    bad = [ins[0], ins[1][:1]]
    with pytest.raises(ValueError):
        np_graph.run_graph(g, bad)
