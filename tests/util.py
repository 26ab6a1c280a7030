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
    gen = _generated_inputs(name, d)
    ins = [gen[k] if k in gen else z[f"in{k}"] for k in range(len(g.inputs))]
    for k, vid in enumerate(g.inputs):
        if g.vars[vid].kind == "rng":  # stored as Philox key (2 words) + counter (4 words)
            w = ins[k]
            ins[k] = np.random.Generator(np.random.Philox(key=w[:2], counter=w[2:]))
    cvm = [z[f"cvm{k}"] for k in range(len(g.outputs))]
    py = [z[f"py{k}"] for k in range(len(g.outputs))]
    return g, ins, cvm, py, d


def _generated_inputs(name, d):
    """Large inputs of a fixture are stored as their RECIPE (``generated_inputs`` in the case's JSON: a generator of
    ``pytensor_amd.configs``, its keyword arguments, and the SHA-256 of every array it must reproduce) instead of
    hundreds of MB of incompressible normal draws; the expected outputs in the ``.npz`` are the reference's, computed
    from exactly those arrays (oracle/make_golden.py).  ``{input position: array}``."""
    spec = d.get("generated_inputs")
    if not spec:
        return {}
    import hashlib

    from pytensor_amd import configs

    vals = getattr(configs, spec["fn"])(**spec["kwargs"])
    out = {}
    for k, nm in enumerate(d["input_names"]):
        if nm in spec["sha256"]:
            a = np.ascontiguousarray(vals[nm])
            h = hashlib.sha256(a.tobytes()).hexdigest()
            assert h == spec["sha256"][nm], f"{name}: regenerated input {nm!r} differs from the one the reference outputs were computed from (NumPy version?)"
            out[k] = a
    return out


_C4_SUMS = {}


def _c4_sums(case):
    """sum|term| of the expanded sums behind config #4's six outputs, from the fixture's own inputs (oracle/bounds.py)"""
    if case not in _C4_SUMS:
        import bounds

        g, ins, cvm, py, d = load_case(case)
        _C4_SUMS.clear()  # (one case at a time: the operands are 160 MB)
        _C4_SUMS[case] = bounds.c4_term_sums(dict(zip(d["input_names"], ins)))
    return _C4_SUMS[case]


NORTH_STAR_RTOL = {"float64": 1e-12, "float32": 1e-5, "float16": 1e-3}  # BASELINE.json north_star
_TOLERANCES = None


def tolerance_for(case, k, want, py=None):
    """``(rtol, atol)`` for output ``k`` of golden case ``case``: north_star's element-wise rtol
    and NO atol, except for the outputs listed — each with its reason and measured error — in
    ``tests/tolerances.json``."""
    global _TOLERANCES
    if _TOLERANCES is None:
        _TOLERANCES = json.load(open(os.path.join(os.path.dirname(GOLDEN), "tolerances.json")))
    want = np.asarray(want)
    rtol, atol = NORTH_STAR_RTOL.get(str(want.dtype), 1e-12), 0.0
    e = _TOLERANCES.get(case, {}).get(str(k))
    if e and want.size:
        fin = want[np.isfinite(want)]
        scale = float(np.max(np.abs(fin))) if fin.size else 0.0
        if "atol_eps_scale" in e:
            atol += e["atol_eps_scale"] * float(np.finfo(want.dtype).eps) * scale
        if "c4_expanded_sum_bound" in e:
            atol = atol + e["c4_expanded_sum_bound"] * float(np.finfo(want.dtype).eps) * np.asarray(_c4_sums(case)[int(k)])
        if "ref_backends_differ" in e:
            assert py is not None, f"{case} out{k}: the reference NumPy-linker output is needed for this tolerance"
            d = np.abs(np.asarray(py, dtype=np.float64) - want.astype(np.float64))
            d = d[np.isfinite(d)]
            atol += e["ref_backends_differ"] * (float(d.max()) if d.size else 0.0)
    return rtol, atol


def assert_parity(got, want, rtol, what, case=None, k=None, py=None, slack=1.0):
    """bit-exact for integer/bool.  Floats: with ``case``/``k`` the element-wise north_star rule of
    :func:`tolerance_for` (``rtol`` is ignored; ``slack`` = 2 when two device results, each within
    tolerance of the reference, are compared with each other); without them (comparisons that are
    not against a golden vector) ``|got-want| <= rtol*|want| + rtol*max|want|``."""
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape, f"{what}: shape {got.shape} != {want.shape}"
    assert got.dtype == want.dtype, f"{what}: dtype {got.dtype} != {want.dtype}"
    if want.dtype.kind in "biu":
        np.testing.assert_array_equal(got, want, err_msg=what)
    elif case is not None:
        r, a = tolerance_for(case, k, want, py)
        if np.ndim(a):  # a per-element absolute term (assert_allclose cannot format an array tolerance)
            g64, w64 = got.astype(np.float64), want.astype(np.float64)
            bad = ~((np.abs(g64 - w64) <= slack * np.asarray(a) + slack * r * np.abs(w64)) | (np.isnan(g64) & np.isnan(w64)) | (g64 == w64))
            assert not bad.any(), (f"{what}: {int(bad.sum())} of {bad.size} entries over rtol {slack * r:g} + the per-element bound; "
                                   f"worst |err| / bound = {float(np.max(np.abs(g64 - w64) / np.maximum(slack * np.asarray(a) + slack * r * np.abs(w64), 1e-300))):.3g}")
            return
        np.testing.assert_allclose(got, want, rtol=slack * r, atol=slack * a, equal_nan=True, err_msg=what)
    else:
        scale = float(np.max(np.abs(want[np.isfinite(want)]))) if np.isfinite(want).any() else 1.0
        np.testing.assert_allclose(got, want, rtol=rtol, atol=rtol * max(scale, 1e-300), equal_nan=True, err_msg=what)
