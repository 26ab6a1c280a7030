"""Summation error yardsticks for the full-size parity checks — TEST INFRASTRUCTURE (NumPy only).

An output that is a sum of mixed-sign terms is compared with ``|err| <= rtol*|want| + c*eps*S``,
``S = sum_i |term_i|`` — the forward error bound of *any* summation order (the reference sums
sequentially / in OpenBLAS blocks, the device in wave butterflies and split-K slabs; both are
entitled to ~log2(n)..n times eps*S, ``c = 8`` is far inside that).  Used by
``tests/test_gpu_fullsize.py`` and the parity gate of ``bench.py``.
"""

from __future__ import annotations

import numpy as np

EPS64 = 2.0**-52
EPS32 = 2.0**-23
C_SUM = 8.0


def c4_term_sums(v):
    """``S`` for the six outputs of BASELINE configs[3]:
    [logp, d/dmu_g, d/dlog_tau, d/dz, d/dbeta, d/dlog_sigma].

    logp is a sum of same-sign terms (S = 0: plain rtol); every gradient is a sum over the N
    observations:  d/dmu_g = sum r/sigma,  d/dlog_tau = sum (r/sigma) tau z[g],
    d/dz[g] = (tau/sigma) sum_{i in g} r_i,  d/dbeta = X^T r / sigma,  d/dlog_sigma = sum (r^2 - 1).

    Every addend is a function of ``r_i = (y_i - a[g_i] - sum_k x_ik beta_k) / sigma`` — itself a sum of K + 2 terms.
    The rule "c * eps * sum |term|" applies to the EXPANDED sum: the K-term dot product inside each addend
    contributes ``D_i = sum_k |x_ik beta_k| / sigma`` per observation, weighted by |d addend / d r_i|.  For K = 128
    and N = 1e6 that part is invisible next to the N-term sums; at K >= 1000 with about one observation per group
    (round 5's sweep rows K in {1000, 2048, 4096} x G = 1e5) it IS the error of d/dz[g]: |r_i| ~ sqrt(K) |beta| while
    D_i ~ K |beta|, so the outer bound alone is sqrt(K)/8 too tight for ANY pair of summation orders (OpenBLAS
    blocks vs the reference's own C loop differ by as much).
    """
    X, y, gidx = v["X"], v["y"], v["gidx"]
    sigma = float(np.exp(v["log_sigma"]))
    tau = float(np.exp(v["log_tau"]))
    a = v["mu_g"] + tau * v["z"]
    r = (y - a[gidx] - X @ v["beta"]) / sigma
    absr = np.abs(r)
    absX = np.abs(X)
    D = (absX @ np.abs(v["beta"])) / sigma  # forward-error scale of r_i: the dot product's own sum |term|
    G = v["z"].shape[0]
    per_group = np.bincount(gidx, weights=absr + D, minlength=G)
    tz = np.abs(tau * v["z"][gidx])
    return [0.0, (absr + D).sum() / sigma, ((absr + D) * tz).sum() / sigma, tau / sigma * per_group,
            (absX.T @ (absr + D)) / sigma, float((r * r + 1.0 + 2.0 * absr * D).sum())]


def check_c4(got, want, v, what="c4"):
    """Raises AssertionError naming the output and the measured excess; returns the worst
    ``err / bound`` per output."""
    S = c4_term_sums(v)
    used = []
    for k, (a, b) in enumerate(zip(got, want)):
        a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
        bound = 1e-12 * np.abs(b) + C_SUM * EPS64 * np.asarray(S[k])
        u = float(np.max(np.abs(a - b) / np.maximum(bound, 1e-300)))
        used.append(u)
        assert u <= 1.0, f"{what} output {k}: |err| is {u:.2f}x over 1e-12*|want| + {C_SUM}*eps*sum|term|"
    return used
