"""Reference-side builders of the BASELINE graphs — TEST INFRASTRUCTURE.

Used by ``oracle/make_golden.py`` (fixtures), by ``bench.py``'s ``cpu_baseline`` leg (the
reference's C linker timed on the bench host: ``mode="CVM"``) and by the end-to-end
GPU tests.  Needs an importable reference (``oracle/make_ref.py``); nothing here is product code.
"""

from __future__ import annotations

import numpy as np


def build_c4(vals):
    """BASELINE configs[3]: hierarchical-normal logp + grad; data as shared
    variables, parameters as explicit inputs.  Returns (params, outputs)."""
    import pytensor
    import pytensor.tensor as pt
    from pytensor.tensor.linalg import cholesky, solve_triangular

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
