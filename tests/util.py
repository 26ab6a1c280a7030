import json
import os

import numpy as np

from pytensor_amd.ir import Graph

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_cases():
    return sorted(f[:-5] for f in os.listdir(GOLDEN) if f.endswith(".json"))


def load_case(name):
    d = json.load(open(os.path.join(GOLDEN, f"{name}.json")))
    g = Graph.from_dict(d)
    z = np.load(os.path.join(GOLDEN, f"{name}.npz"))
    ins = [z[f"in{k}"] for k in range(len(g.inputs))]
    for k, vid in enumerate(g.inputs):
        if g.vars[vid].kind == "rng":  # stored as Philox key (2 words) + counter (4 words)
            w = ins[k]
            ins[k] = np.random.Generator(np.random.Philox(key=w[:2], counter=w[2:]))
    cvm = [z[f"cvm{k}"] for k in range(len(g.outputs))]
    py = [z[f"py{k}"] for k in range(len(g.outputs))]
    return g, ins, cvm, py, d


def assert_parity(got, want, rtol, what):
    """bit-exact for integer/bool, rtol (fp64 1e-12 / fp32 1e-5 per north_star) for floats"""
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape, f"{what}: shape {got.shape} != {want.shape}"
    assert got.dtype == want.dtype, f"{what}: dtype {got.dtype} != {want.dtype}"
    if want.dtype.kind in "biu":
        np.testing.assert_array_equal(got, want, err_msg=what)
    else:
        scale = float(np.max(np.abs(want[np.isfinite(want)]))) if np.isfinite(want).any() else 1.0
        np.testing.assert_allclose(got, want, rtol=rtol, atol=rtol * max(scale, 1e-300), equal_nan=True, err_msg=what)
