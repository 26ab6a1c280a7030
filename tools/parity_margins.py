"""Element-wise parity margins of every golden output at north_star's tolerances (fp64 rtol 1e-12,
fp32 1e-5, no atol): for the device executor (on a GPU) and — always — for the reference's own
NumPy linker against its C linker (both stored in the fixtures).  An output whose two *reference*
backends already differ by more than the tolerance cannot be held to it; the table this prints is
where tests/tests/tolerances.json comes from.

usage: python tools/parity_margins.py [--device] [--oracle] [case ...] > margins.json
"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from util import golden_cases, load_case

NS = {"float64": 1e-12, "float32": 1e-5, "float16": 1e-3}


def margin(got, want):
    got, want = np.asarray(got), np.asarray(want)
    if want.dtype.kind not in "f" or got.shape != want.shape:
        return None
    r = NS.get(str(want.dtype), 1e-12)
    g, w = got.astype(np.float64).ravel(), want.astype(np.float64).ravel()
    fin = np.isfinite(w)
    same_nonfinite = bool(np.array_equal(g[~fin], w[~fin], equal_nan=True))
    g, w = g[fin], w[fin]
    if w.size == 0:
        return {"over": 0.0, "same_nonfinite": same_nonfinite}
    err = np.abs(g - w)
    with np.errstate(divide="ignore", invalid="ignore"):
        rel = np.where(w != 0, err / np.abs(w), np.where(err == 0, 0.0, np.inf))
    k = int(np.argmax(rel))
    return {"over": float(rel[k] / r), "max_rel": float(rel[k]), "at_want": float(w[k]), "max_abs": float(err.max()),
            "scale": float(np.abs(w).max()), "n_over": int((rel > r).sum()), "n": int(w.size), "same_nonfinite": same_nonfinite}


def main():
    device = "--device" in sys.argv
    if device:
        from pytensor_amd import ffi
        from pytensor_amd.executor import HipExecutable
        ffi.init(0)
    out = {}
    only = [a for a in sys.argv[1:] if not a.startswith("--")]
    for name in golden_cases():
        if only and name not in only:
            continue
        g, ins, cvm, py, d = load_case(name)
        rec = {"py_vs_cvm": [margin(a, b) for a, b in zip(py, cvm)]}
        if "--oracle" in sys.argv:
            import np_graph
            rec["oracle_vs_cvm"] = [margin(a, b) for a, b in zip(np_graph.run_graph(g, ins), cvm)]
        if device:
            try:
                got = HipExecutable(g)(*ins)
                rec["hip_vs_cvm"] = [margin(a, b) for a, b in zip(got, cvm)]
            except Exception as e:  # noqa: BLE001
                rec["error"] = repr(e)[:200]
        out[name] = rec
    json.dump(out, sys.stdout, indent=0)


if __name__ == "__main__":
    main()
