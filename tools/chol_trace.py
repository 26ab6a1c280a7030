"""Where the task-graph Cholesky (csrc/linalg.hip ``chol_dag_kernel``) spends its critical path.

Runs ``pthip_potrf`` once with ``PTHIP_CHOL_TRACE`` set (the kernel then stamps the 100 MHz clock at
up to sixteen points of every task) and prints, per phase, the time along the chain
diag(j) -> solve(j+1, j) -> diag(j+1): the wait for the last dependency, the accumulation that was still
left, the factorisation / substitution itself, the write-back + publication — plus what the other
workgroups did meanwhile (mean task length, share of time spent waiting).

usage: python tools/chol_trace.py [n] [dtype]      (on the MI355X box)
"""
import os
import sys
import tempfile

import numpy as np

path = os.path.join(tempfile.gettempdir(), "pthip_chol_trace.bin")
os.environ["PTHIP_CHOL_TRACE"] = path
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytensor_amd import ffi  # noqa: E402
from pytensor_amd.device import DeviceArray  # noqa: E402


def main(n, dtype):
    ffi.init(0)
    lib = ffi.lib()
    rng = np.random.default_rng(n)
    A = rng.normal(size=(n, n + 8))
    S = (A @ A.T / n + np.eye(n)).astype(dtype)
    dS = DeviceArray.from_host(S)
    L = DeviceArray.empty((n, n), dtype)
    for _ in range(3):  # the last run's trace is the one read
        ffi.check(lib.pthip_potrf(ffi.np_dtype_code(dtype), 1, 1, n, dS.ptr, L.ptr))
    raw = np.fromfile(path, dtype=np.int64)
    nT, grid = int(raw[0]), int(raw[1])
    tr = raw[2:].reshape(-1, 16).astype(np.float64) / 100.0  # us
    t0 = tr[:, 0].min()
    tr = tr - t0
    # task numbering: column j holds its head task H(j) (tiles (j, j-1) and (j, j)) + tiles (i, j), i >= j + 2
    col_start = np.concatenate([[0], np.cumsum([1 + max(0, nT - j - 2) for j in range(nT)])]).astype(int)
    print(f"n={n} {dtype}: nT={nT} grid={grid} total {tr[:, 15].max():.1f} us")
    names = ["accumulate (+waits)", "diag(j-1) seen + its factor loaded", "substitution", "L(j,j-1) stores issued (waves 1-3)",
             "last update X X^T", "-", "barrier", "residual -> LDS, barrier",
             "potrf: panel 0 rows -> registers, 16 columns", "potrf: panel 0 -> LDS, trailing update (6 tiles)", "potrf: panel 1 (3 tiles)", "potrf: panel 2 (1 tile)",
             "potrf: panel 3", "other waves' tail + release fence", "barrier + flag (j,j)"]
    hsum = np.zeros(15)
    hand = 0.0
    for j in range(1, nT):
        h = tr[col_start[j]]
        hsum += np.diff(h)
        hand += h[2] - tr[col_start[j - 1], 15]  # H(j-1) published its diagonal tile -> H(j) holds the factor in LDS
    k = nT - 1
    print("head task H(j), mean us per phase:")
    for nm, v in zip(names, hsum):
        print(f"  {nm:40s} {v / k:8.2f}")
    print(f"hand-over H(j-1) published -> H(j) has the factor in LDS: {hand / k:.2f} us")
    chain = [tr[col_start[j + 1], 15] - tr[col_start[j], 15] for j in range(nT - 1)]
    print(f"chain period: mean {np.mean(chain):.2f} us (min {np.min(chain):.2f}, max {np.max(chain):.2f})")
    ordinary = np.ones(len(tr), bool)
    ordinary[col_start[:-1]] = False
    o = tr[ordinary]
    if len(o):
        print(f"ordinary tasks: accumulate (+waits) {np.mean(o[:, 1] - o[:, 0]):.1f} us, wait diag + fetch {np.mean(o[:, 2] - o[:, 1]):.2f} us, "
              f"substitution {np.mean(o[:, 3] - o[:, 2]):.2f} us, stores + fence {np.mean(o[:, 6] - o[:, 3]):.2f} us, barrier + flag {np.mean(o[:, 15] - o[:, 6]):.2f} us")
    dur = tr[:, 15] - tr[:, 0]
    print(f"all tasks: mean length {dur.mean():.1f} us, max {dur.max():.1f} us; busy sum {dur.sum() / grid:.1f} us per workgroup")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 4096, sys.argv[2] if len(sys.argv) > 2 else "float64")
