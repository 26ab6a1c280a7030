"""Seeded random regression-style logp + gradient graphs (imported by ``make_golden.py``): the north_star path
(``Gemv -> Composite -> Gemv`` with a gather and a scatter-add riding along, fused into one pass over X by
pytensor_amd/fusion.py fuse_gemv_chain) for models nobody hand-picked.  TEST INFRASTRUCTURE.

Per case: N observations x K predictors (K from 1 to 257: below, at and beyond a pack / a wave / the 256-column
chunk), X as the caller's C-ordered array, a column block of a wider array (lda > K, odd lda) or the transpose of a
(K, N) array; with or without a group intercept ``a[gidx]`` (1, 7 or 300 groups: beyond the 256 scatter bins), with or
without observation weights; four likelihood families (normal with a scale parameter, Bernoulli-logit, Poisson-log,
Student-t); standard-normal priors.  Outputs: logp and its gradient with respect to every parameter.  Data are shared
variables (resident operands), parameters explicit inputs — the shape of BASELINE configs[3] (oracle/ref_graphs.build_c4).
Reference: Gemv (pytensor/tensor/blas/gemv.py:64-108), Elemwise / CAReduce (elemwise.py:375, 1233),
AdvancedSubtensor1 / AdvancedIncSubtensor1 (subtensor.py).
"""

from __future__ import annotations

import numpy as np
import pytensor
import pytensor.tensor as pt

from make_golden import case

_NK = [(1, 5), (7, 3), (513, 64), (513, 31), (4097, 8), (4097, 1), (300, 130), (150, 257), (1000, 2), (257, 127), (2048, 16), (640, 100)]


def _make(seed):
    def build():
        rng = np.random.default_rng(13000 + seed)
        N, K = _NK[seed % len(_NK)]
        fam = seed % 4
        G = int(rng.choice([1, 7, 300]))
        use_group = rng.random() < 0.7
        use_w = rng.random() < 0.4
        layout = str(rng.choice(["c", "wide", "wide_off", "f"]))
        Xv = rng.normal(size=(N, K)) / np.sqrt(K)
        if layout == "c":
            Xs = pytensor.shared(Xv, name="X")
            X = Xs
        elif layout in ("wide", "wide_off"):
            off = 0 if layout == "wide" else 1
            store = rng.normal(size=(N, K + 3))
            store[:, off:off + K] = Xv
            Xs = pytensor.shared(store, name="X")
            X = Xs[:, off:off + K]
        else:
            Xs = pytensor.shared(np.ascontiguousarray(Xv.T), name="X")
            X = Xs.T
        vals = {"X": Xs.get_value(borrow=True)}
        beta = pt.dvector("beta")
        params, pvals = [beta], {"beta": rng.normal(size=K)}
        eta = X @ beta
        prior = -0.5 * pt.sum(beta**2)
        shared = [Xs]
        if use_group:
            gv = rng.integers(0, G, size=N)
            gidx = pytensor.shared(gv, name="gidx")
            shared.append(gidx)
            vals["gidx"] = gv
            a = pt.dvector("a")
            params.append(a)
            pvals["a"] = rng.normal(size=G)
            eta = eta + a[gidx]
            prior = prior - 0.5 * pt.sum(a**2)
        if fam == 0:
            yv = rng.normal(size=N) * 2
            ls = pt.dscalar("log_sigma")
            params.append(ls)
            pvals["log_sigma"] = np.asarray(0.3)
            y = pytensor.shared(yv, name="y")
            r = (y - eta) * pt.exp(-ls)
            ll = -0.5 * r**2 - ls
            prior = prior - 0.5 * ls**2
        elif fam == 1:
            yv = rng.integers(0, 2, size=N).astype("float64")
            y = pytensor.shared(yv, name="y")
            ll = y * eta - pt.softplus(eta)
        elif fam == 2:
            yv = rng.poisson(2.0, size=N).astype("float64")
            y = pytensor.shared(yv, name="y")
            ll = y * eta - pt.exp(eta)
        else:
            yv = rng.standard_t(3, size=N)
            ls = pt.dscalar("log_sigma")
            params.append(ls)
            pvals["log_sigma"] = np.asarray(-0.2)
            y = pytensor.shared(yv, name="y")
            r = (y - eta) * pt.exp(-ls)
            ll = -2.0 * pt.log1p(r**2 / 3.0) - ls
            prior = prior - 0.5 * ls**2
        shared.append(y)
        vals["y"] = yv
        if use_w:
            wv = rng.uniform(0.5, 2.0, size=N)
            w = pytensor.shared(wv, name="w")
            shared.append(w)
            vals["w"] = wv
            ll = w * ll
        logp = pt.sum(ll) + prior
        outs = [logp, *pytensor.grad(logp, params)]
        # (make_golden stores the shared data by name next to the explicit parameter values)
        vals.update(pvals)
        return params, outs, vals

    return build


for _s in range(12):
    case(f"glm_fuzz_{_s}", rtol=1e-10)(_make(_s))


# ---- many observed variables: T independent likelihood terms over their own data vectors of DIFFERENT lengths ----
# (pytensor_amd/dispatch/wide.py: one launch for all terms' reductions, workgroups per term in proportion to its cost;
#  dispatch/tail.py: the scalar chains behind them, split when too long for one launch.  Golden wide_200 pins T = 48 equal
#  lengths; these pin ragged lengths — 1 element to 5000 — and term counts 1 .. 40 (the reference NumPy linker stops at 64 operands per node).)
def _make_wide(seed):
    def build():
        rng = np.random.default_rng(15000 + seed)
        T = int([1, 5, 17, 33, 40, 12][seed % 6])
        mu, ls = pt.dvector("mu"), pt.dvector("ls")
        vals = {"mu": rng.normal(size=T) * 0.5, "ls": rng.normal(size=T) * 0.3}
        terms = []
        for k in range(T):
            n = int(rng.choice([1, 2, 7, 64, 257, 1000, 5000]))
            wv = rng.normal(size=n) * 1.5 + 0.2
            w = pytensor.shared(wv, name=f"w{k}")
            vals[f"w{k}"] = wv
            r = (w - mu[k]) * pt.exp(-ls[k])
            fam = int(rng.integers(4))
            if fam == 0:
                terms.append((-0.5 * r**2 - ls[k]).sum())
            elif fam == 1:
                terms.append((-pt.log1p(r**2 / 3.0) * 2.0 - ls[k]).sum())
            elif fam == 2:
                terms.append((-pt.sqrt(1.0 + r**2) - ls[k]).sum())
            else:
                terms.append((-r - 2.0 * pt.softplus(-r) - ls[k]).sum())
        total = terms[0]
        for t in terms[1:]:
            total = total + t
        logp = total - 0.5 * pt.sum(mu**2) - 0.5 * pt.sum(ls**2)
        return [mu, ls], [logp, *pytensor.grad(logp, [mu, ls])], vals

    return build


for _s in range(6):
    case(f"wide_fuzz_{_s}", rtol=1e-10)(_make_wide(_s))
