"""Reference-side builders of the BASELINE graphs — TEST INFRASTRUCTURE.

Used by ``oracle/make_golden.py`` (fixtures), by ``bench.py``'s ``cpu_baseline`` leg (the
reference's C linker timed on the bench host: ``mode="CVM"``) and by the end-to-end
GPU tests.  Needs an importable reference (``oracle/make_ref.py``); nothing here is product code.
"""

from __future__ import annotations

import numpy as np


def build_c4(vals, dtype="float64"):
    """BASELINE configs[3]: hierarchical-normal logp + grad; data as shared
    variables, parameters as explicit inputs.  Returns (params, outputs).  ``dtype="float32"``: the same model as a
    ``floatX=float32`` user would build it (values cast by the caller)."""
    import pytensor
    import pytensor.tensor as pt
    from pytensor.tensor.linalg import cholesky, solve_triangular

    if dtype != "float64":
        return _build_c4_typed(vals, dtype)
    K = vals["X"].shape[1]
    y = pytensor.shared(vals["y"], name="y")
    X = pytensor.shared(vals["X"], name="X")
    gidx = pytensor.shared(vals["gidx"], name="gidx")
    Sigma = pytensor.shared(vals["Sigma"], name="Sigma")
    mu_g, log_tau, log_sigma = pt.dscalar("mu_g"), pt.dscalar("log_tau"), pt.dscalar("log_sigma")
    z, beta = pt.dvector("z"), pt.dvector("beta")
    tau = pt.exp(log_tau)
    sigma = pt.exp(log_sigma)
    a = mu_g + tau * z
    L = cholesky(Sigma)
    alpha = solve_triangular(L, beta, lower=True)
    logp_beta = -0.5 * pt.sum(alpha**2) - pt.sum(pt.log(pt.diag(L))) - 0.5 * K * np.log(2 * np.pi)
    eta = a[gidx] + X @ beta
    r = (y - eta) / sigma
    logp_y = pt.sum(-0.5 * r**2 - log_sigma - 0.5 * np.log(2 * np.pi))
    logp_z = pt.sum(-0.5 * z**2 - 0.5 * np.log(2 * np.pi))
    logp_hyp = -0.5 * (mu_g**2 + log_tau**2 + log_sigma**2)
    logp = logp_y + logp_z + logp_beta + logp_hyp
    params = [mu_g, log_tau, z, beta, log_sigma]
    return params, [logp, *pytensor.grad(logp, params)]


def build_wide200(vals, T=None):
    """north_star's literal target ("PyMC-style logp+grad graph, ~200 fused Elemwise + 1 Gemm + 1 Cholesky"): config #4's
    model plus ``T`` independent likelihood terms of four families (normal, Student-t(3)-like, Laplace, logistic) over
    their own data vectors — the shape of a PyMC model with many observed variables.  Data shared, parameters explicit.
    Returns (params, outputs = [logp, d logp / d params])."""
    import pytensor
    import pytensor.tensor as pt

    from pytensor_amd import configs

    T = T or configs.WIDE_T
    params, outs = build_c4(vals)
    logp = outs[0]
    wmu, wls = pt.dvector("wmu"), pt.dvector("wls")
    terms = []
    for k in range(T):
        w = pytensor.shared(vals[f"w{k}"], name=f"w{k}")
        r = (w - wmu[k]) * pt.exp(-wls[k])
        fam = k % 4
        if fam == 0:
            terms.append((-0.5 * r**2 - wls[k]).sum())
        elif fam == 1:
            terms.append((-pt.log1p(r**2 / 3.0) * 2.0 - wls[k]).sum())
        elif fam == 2:
            terms.append((-pt.abs(r) - wls[k]).sum())
        else:
            terms.append((-r - 2.0 * pt.softplus(-r) - wls[k]).sum())
    total = logp
    for t in terms:  # (binary adds: the reference's Python Elemwise refuses more than 32 operands)
        total = total + t
    params = [*params, wmu, wls]
    return params, [total, *pytensor.grad(total, params)]


def build_wide200_gemm(vals, T=None):
    """north_star's target WITH a real ``Gemm`` (round 6): the multi-response form of the same model — R response
    columns ``Y (N, R)`` regressed on the shared design matrix, coefficients ``B (K, R)``, a fixed offset matrix
    ``O (N, R)`` (exposure / known effects): ``eta = O + X @ B + a[gidx][:, None]``.  The reference's
    ``GemmOptimizer`` (tensor/rewriting/blas.py:437) turns ``O + X @ B`` into ``Gemm(O, 1, X, B, 1)``
    (tensor/blas/gemm.py:76) and the gradient ``X.T @ (r / sigma) - (prior term)`` into a second one; the Cholesky
    prior acts on every column of ``B`` (a matrix triangular solve).  Plus the ``T`` likelihood terms of
    :func:`build_wide200`.  Data shared, parameters explicit.  Returns (params, outputs = [logp, d logp / d params])."""
    import pytensor
    import pytensor.tensor as pt
    from pytensor.tensor.linalg import cholesky, solve_triangular

    from pytensor_amd import configs

    T = configs.WIDE_T if T is None else T
    K, R = vals["B"].shape
    Y = pytensor.shared(vals["Y"], name="Y")
    O = pytensor.shared(vals["O"], name="O")
    X = pytensor.shared(vals["X"], name="X")
    gidx = pytensor.shared(vals["gidx"], name="gidx")
    Sigma = pytensor.shared(vals["Sigma"], name="Sigma")
    mu_g, log_tau, log_sigma = pt.dscalar("mu_g"), pt.dscalar("log_tau"), pt.dscalar("log_sigma")
    z, B = pt.dvector("z"), pt.dmatrix("B")
    tau, sigma = pt.exp(log_tau), pt.exp(log_sigma)
    a = mu_g + tau * z
    L = cholesky(Sigma)
    alpha = solve_triangular(L, B, lower=True)
    logp_B = -0.5 * pt.sum(alpha**2) - R * pt.sum(pt.log(pt.diag(L))) - 0.5 * K * R * np.log(2 * np.pi)
    eta = O + pt.dot(X, B) + a[gidx][:, None]
    r = (Y - eta) / sigma
    logp_y = pt.sum(-0.5 * r**2 - log_sigma - 0.5 * np.log(2 * np.pi))
    logp_z = pt.sum(-0.5 * z**2 - 0.5 * np.log(2 * np.pi))
    logp_hyp = -0.5 * (mu_g**2 + log_tau**2 + log_sigma**2)
    total = logp_y + logp_z + logp_B + logp_hyp
    wmu, wls = pt.dvector("wmu"), pt.dvector("wls")
    for k in range(T):
        w = pytensor.shared(vals[f"w{k}"], name=f"w{k}")
        rr = (w - wmu[k]) * pt.exp(-wls[k])
        fam = k % 4
        if fam == 0:
            total = total + (-0.5 * rr**2 - wls[k]).sum()
        elif fam == 1:
            total = total + (-pt.log1p(rr**2 / 3.0) * 2.0 - wls[k]).sum()
        elif fam == 2:
            total = total + (-pt.abs(rr) - wls[k]).sum()
        else:
            total = total + (-rr - 2.0 * pt.softplus(-rr) - wls[k]).sum()
    params = [mu_g, log_tau, z, B, log_sigma] + ([wmu, wls] if T else [])
    return params, [total, *pytensor.grad(total, params)]


def _build_c4_typed(vals, dtype):
    import pytensor
    import pytensor.tensor as pt
    from pytensor.tensor.linalg import cholesky, solve_triangular

    f = lambda a: np.asarray(a, dtype=dtype)
    K = vals["X"].shape[1]
    y = pytensor.shared(f(vals["y"]), name="y")
    X = pytensor.shared(f(vals["X"]), name="X")
    gidx = pytensor.shared(vals["gidx"], name="gidx")
    Sigma = pytensor.shared(f(vals["Sigma"]), name="Sigma")
    mu_g, log_tau, log_sigma = (pt.scalar(n, dtype=dtype) for n in ("mu_g", "log_tau", "log_sigma"))
    z, beta = pt.vector("z", dtype=dtype), pt.vector("beta", dtype=dtype)
    c = lambda v: np.asarray(v, dtype=dtype)
    tau, sigma = pt.exp(log_tau), pt.exp(log_sigma)
    a = mu_g + tau * z
    L = cholesky(Sigma)
    alpha = solve_triangular(L, beta, lower=True)
    logp_beta = c(-0.5) * pt.sum(alpha**2) - pt.sum(pt.log(pt.diag(L))) - c(0.5 * K * np.log(2 * np.pi))
    eta = a[gidx] + X @ beta
    r = (y - eta) / sigma
    logp_y = pt.sum(c(-0.5) * r**2 - log_sigma - c(0.5 * np.log(2 * np.pi)))
    logp_z = pt.sum(c(-0.5) * z**2 - c(0.5 * np.log(2 * np.pi)))
    logp_hyp = c(-0.5) * (mu_g**2 + log_tau**2 + log_sigma**2)
    logp = logp_y + logp_z + logp_beta + logp_hyp
    params = [mu_g, log_tau, z, beta, log_sigma]
    return params, [logp, *pytensor.grad(logp, params)]
