"""Synthetic inputs for the BASELINE.json configs (NumPy only — no PyTensor).

The *graphs* of the configs are built with the reference in
``oracle/make_golden.py`` and lowered to the IR fixtures under
``tests/golden/``; this module produces the matching synthetic inputs by
variable name at any size, with fixed ``default_rng`` seeds.
"""

from __future__ import annotations

import numpy as np

C4_K = 128
C4_G = 128


def c1_inputs(N=100_000, seed=0):
    rng = np.random.default_rng(seed)
    return {"x": rng.normal(size=N), "mu": np.asarray(0.3)}


def c2_inputs(N=10_000_000, seed=1):
    rng = np.random.default_rng(seed)
    return {"x": rng.normal(size=N), "y": rng.normal(size=N)}


def c3_inputs(M=4096, B=512, Bn=256, seed=2):
    rng = np.random.default_rng(seed)
    return {
        "A": rng.normal(size=(M, M)),
        "B": rng.normal(size=(M, M)),
        "v": rng.normal(size=M),
        "X3": rng.normal(size=(B, Bn, Bn)).astype("float32"),
        "Y3": rng.normal(size=(B, Bn, Bn)).astype("float32"),
    }


def c4_inputs(N=1_000_000, K=C4_K, G=C4_G, seed=0, chain=0):
    """Hierarchical-normal data (shared, device resident) + one parameter draw.

    ``chain`` selects the parameter draw (8 independent chains = 8 draws over the
    same data).
    """
    rng = np.random.default_rng(seed)
    y = rng.normal(size=N)
    X = rng.normal(size=(N, K))
    gidx = rng.integers(0, G, size=N).astype("int64")
    A = rng.normal(size=(K, K))
    Sigma = A @ A.T + K * np.eye(K)
    prng = np.random.default_rng(1000 + chain)
    return {
        "y": y,
        "X": X,
        "gidx": gidx,
        "Sigma": Sigma,
        "mu_g": np.asarray(0.1 + 0.01 * chain),
        "log_tau": np.asarray(-0.2),
        "z": prng.normal(size=G),
        "beta": 0.1 * prng.normal(size=K),
        "log_sigma": np.asarray(0.3),
    }


C4_DATA = ("y", "X", "gidx", "Sigma")  # shared (resident) inputs
C4_PARAMS = ("mu_g", "log_tau", "z", "beta", "log_sigma")

WIDE_T = 48  # likelihood terms of the wide model


def wide200_inputs(N=1_000_000, T=WIDE_T, K=C4_K, G=C4_G, seed=0, chain=0):
    """north_star's literal target graph ("~200 fused Elemwise + 1 Gemm + 1 Cholesky, N=1e6"): the
    hierarchical-normal model of config #4 (design matrix, Cholesky(K) prior) plus ``T`` independent likelihood
    terms of four families over their own N-vectors ``w0 .. w{T-1}`` with location ``wmu[k]`` and log-scale
    ``wls[k]``.  Data shared (device resident), parameters explicit."""
    d = c4_inputs(N=N, K=K, G=G, seed=seed, chain=chain)
    rng = np.random.default_rng(seed + 77)
    for k in range(T):
        d[f"w{k}"] = rng.normal(size=N) + 0.1 * (k % 7)
    prng = np.random.default_rng(2000 + chain)
    d["wmu"] = prng.normal(size=T) * 0.1
    d["wls"] = prng.normal(size=T) * 0.1
    return d


def wide200_params():
    return (*C4_PARAMS, "wmu", "wls")


WIDE_R = 8  # response columns of the multi-response variant


def wide200_gemm_inputs(N=1_000_000, T=WIDE_T, K=C4_K, G=C4_G, R=WIDE_R, seed=0, chain=0):
    """The multi-response form of :func:`wide200_inputs` (oracle/ref_graphs.build_wide200_gemm): ``Y (N, R)`` responses
    and a fixed offset matrix ``O (N, R)`` over the same design matrix, coefficients ``B (K, R)`` — the graph then
    holds a real ``Gemm`` (``O + X @ B``, tensor/blas/gemm.py:76)."""
    d = wide200_inputs(N=N, T=T, K=K, G=G, seed=seed, chain=chain)
    rng = np.random.default_rng(seed + 177)
    d["Y"] = rng.normal(size=(N, R))
    d["O"] = 0.1 * rng.normal(size=(N, R))
    prng = np.random.default_rng(3000 + chain)
    d["B"] = 0.1 * prng.normal(size=(K, R))
    del d["y"], d["beta"]
    if T == 0:
        del d["wmu"], d["wls"]
    return d


def wide200_gemm_params():
    return ("mu_g", "log_tau", "z", "B", "log_sigma", "wmu", "wls")


def c5_inputs(T=1000, B=64, H=1024, seed=5):
    rng = np.random.default_rng(seed)
    d = {
        "xs": rng.normal(size=(T, B, H)).astype("float32"),
        "h0": np.zeros((B, H), dtype="float32"),
    }
    for n in ("Wz", "Wr", "Wh", "Uz", "Ur", "Uh"):
        d[n] = (0.03 * rng.normal(size=(H, H))).astype("float32")
    for n in ("bz", "br", "bh"):
        d[n] = (0.03 * rng.normal(size=H)).astype("float32")
    return d
