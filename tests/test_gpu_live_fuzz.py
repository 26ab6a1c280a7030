"""GPU: seeded random graphs the fixtures do NOT hold — built with the reference's front end on the GPU box, compiled with
``mode="hip"`` and with the reference C linker (``Mode("cvm")``) in the same process, compared output by output.

The generators are the ones behind the committed ``layout_fuzz*`` / ``glm_fuzz*`` goldens (oracle/golden_cases_layout_fuzz.py,
golden_cases_glm_fuzz.py: random ranks, extents that straddle the tile sizes, operands as strided / reversed / permuted /
broadcast views, reductions over random axis subsets, softmax / log-sum-exp along random axes, regression models with
gathers and scatter-adds) with OTHER seeds: the fixtures pin 60 draws for ever, this module draws new ones — by default a
dozen per run (``PTHIP_FUZZ_CASES`` raises it; round 6 ran 2025, profiles/r8_live_fuzz.txt).  No fixture means no per-output
tolerance table: floats are held to ``|err| <= rtol*|want| + 64 eps * max|want|`` (rtol 1e-10 fp64 / 1e-4 fp32) — an
indexing, layout or reduction bug is an O(1) error — integers and booleans exactly, every call twice (eager, then captured)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_CASES = int(os.environ.get("PTHIP_FUZZ_CASES", "13"))  # one per family
SEED0 = int(os.environ.get("PTHIP_FUZZ_SEED0", "31000"))
ONLY = os.environ.get("PTHIP_FUZZ_FAMILY", "")  # e.g. "special,multi_response"


@pytest.fixture(scope="module")
def gens():
    import e2e_util

    pytensor = e2e_util.activate()
    if not e2e_util.have_gpu():
        pytest.fail("no HIP device visible: GPU tests must run on the MI355X box")
    sys.modules.setdefault("make_golden", __import__("make_golden"))
    import golden_cases_glm_fuzz as G
    import golden_cases_layout_fuzz as L

    return pytensor, e2e_util, L, G


def _multi_response(seed):
    """the multi-response regression of oracle/ref_graphs.build_wide200_gemm (two Gemm nodes, a matrix triangular solve)
    at random sizes on both sides of the skinny-GEMM kernels' thresholds (N >= 8192 rows, R <= 16 columns)"""
    def build():
        import ref_graphs
        from pytensor_amd import configs

        rng = np.random.default_rng(23000 + seed)
        N = int(rng.choice([300, 5000, 8192, 9001, 20011, 40000]))
        K = int(rng.choice([1, 5, 16, 37, 128, 130, 257]))
        R = int(rng.choice([1, 2, 5, 8, 9, 16, 17]))
        G = int(rng.choice([1, 7, 50, 300]))
        vals = configs.wide200_gemm_inputs(N=N, T=0, K=K, G=G, R=R, seed=seed)
        ins, outs = ref_graphs.build_wide200_gemm(vals, T=0)
        return ins, outs, vals

    return build


def _special(seed):
    """softmax / log_softmax / logsumexp (and a gradient through each) at extents on both sides of every switch of the
    round-6 row and column kernels (csrc/softmax.hip: thread-per-row up to 64 columns, wave-per-row, streamed rows, column
    tiles with a partial/finish split), along trailing, leading, middle and split axis sets, fp64 and fp32"""
    def build():
        import pytensor
        import pytensor.tensor as pt

        rng = np.random.default_rng(29000 + seed)
        dt = "float64" if rng.random() < 0.6 else "float32"
        rank = int(rng.choice([1, 2, 2, 2, 3, 3, 4]))
        small = [1, 2, 3, 7, 10, 31, 32, 33, 63, 64, 65]
        mid = [100, 127, 128, 129, 255, 256, 257, 511, 513, 1000, 1024, 2049]
        big = [4097, 8192, 16385, 40000, 100003]
        while True:
            shape = []
            for _ in range(rank):
                r = rng.random()
                shape.append(int(rng.choice(small if r < 0.55 else mid if r < 0.9 else big)))
            if np.prod(shape) <= 3_000_000:
                break
        kind = int(rng.integers(4))
        if rank == 1 or kind == 0:
            axes = (rank - 1,)
        elif kind == 1:
            axes = tuple(range(int(rng.integers(1, rank))))  # leading run
        elif kind == 2:
            a0 = int(rng.integers(rank))
            axes = tuple(range(a0, int(rng.integers(a0, rank)) + 1))  # any contiguous run
        else:
            axes = tuple(sorted(rng.choice(rank, size=int(rng.integers(1, rank + 1)), replace=False).tolist()))
        x = pt.tensor("x", dtype=dt, shape=(None,) * rank)
        w = pt.tensor("w", dtype=dt, shape=(None,) * rank)
        xv = (rng.standard_normal(shape) * float(rng.choice([0.1, 1.0, 30.0]))).astype(dt)
        if rng.random() < 0.3:
            xv.flat[rng.integers(xv.size, size=max(1, xv.size // 50))] = -np.inf  # masked entries
        wv = rng.standard_normal(shape).astype(dt)
        ax1 = axes[0] if len(axes) == 1 else None
        outs = [pt.special.logsumexp(x, axis=axes, keepdims=bool(rng.integers(2)))]
        # absolute floors: a gradient here is a difference of terms the size of the upstream gradient (dy*sm - sm*sum(dy*sm),
        # dy - sm*sum(dy)); both linkers round those terms, so the error scales with them, not with the (cancelled) result
        scales = {}
        if ax1 is not None or len(axes) == rank:
            sm = pt.special.softmax(x, axis=ax1)
            lsm = pt.special.log_softmax(x * 0.5, axis=ax1)
            scales[len(outs) + 2] = float(np.abs(wv).max())
            outs += [sm, lsm, pytensor.grad(pt.sum(sm * w), x)]
            if not np.isinf(xv).any():
                scales[len(outs)] = float(np.abs(wv).sum(axis=ax1).max())
                outs.append(pytensor.grad(pt.sum(lsm * w), x))
        if not np.isinf(xv).any():
            # the reference's gradient graph carries a term that cancels exactly in real arithmetic, (g - (g/s)*s) at each
            # row's maximum, with g = 2*lse: rounding noise eps*|g| lands on outputs of size |g|/n — the floor scales with g
            scales[len(outs)] = 2.0 * (float(np.abs(xv).max()) + float(np.log(max(2, xv.size))))
            outs.append(pytensor.grad(pt.sum(pt.special.logsumexp(x, axis=axes) ** 2), x))
        return [x, w], outs, {"x": xv, "w": wv, "_scales": scales}

    return build


def _linalg(seed):
    """Cholesky / triangular and general solves / det / inverse (+ a Gaussian-process style log-density and its gradient) at
    sizes on both sides of the one-CU, LDS-resident and blocked kernels' switches (csrc/linalg.hip), with and without batch
    dimensions, vector, few-column and many-column right-hand sides; matrices with condition number <= ~100"""
    def build():
        import pytensor
        import pytensor.tensor as pt
        from pytensor.tensor import slinalg

        rng = np.random.default_rng(37000 + seed)
        n = int(rng.choice([1, 2, 3, 16, 31, 32, 33, 64, 65, 100, 128, 129, 160, 161, 255, 256, 257, 300, 513]))
        batch = [(), (), (3,), (2, 2)][int(rng.integers(4))] if n <= 160 else ()
        nrhs = [None, 1, 2, 5, 16, 17, 40][int(rng.integers(7))]
        bshape = batch + ((n,) if nrhs is None else (n, nrhs))
        q, _ = np.linalg.qr(rng.standard_normal(batch + (n, n)))
        lam = np.exp(rng.uniform(0.0, np.log(100.0), batch + (n,)))
        spd = (q * lam[..., None, :]) @ np.swapaxes(q, -1, -2)
        spd = 0.5 * (spd + np.swapaxes(spd, -1, -2))
        gen = q @ (np.triu(rng.standard_normal(batch + (n, n)) * 0.3 / np.sqrt(n), 1) + lam[..., None] * np.eye(n))
        bv = rng.standard_normal(bshape)
        A = pt.tensor("A", dtype="float64", shape=(None,) * (len(batch) + 2))
        S = pt.tensor("S", dtype="float64", shape=(None,) * (len(batch) + 2))
        b = pt.tensor("b", dtype="float64", shape=(None,) * len(bshape))
        lower = bool(rng.integers(2))
        L = slinalg.cholesky(S, lower=lower)
        outs = [L, slinalg.solve_triangular(L, b, lower=lower, trans=int(rng.integers(2)), b_ndim=1 if nrhs is None else 2),
                slinalg.solve(A, b, b_ndim=1 if nrhs is None else 2),
                slinalg.solve(S, b, assume_a="pos", b_ndim=1 if nrhs is None else 2),
                pt.linalg.det(A), pt.linalg.inv(S)]
        if not batch:
            tri = pt.tril(A) if lower else pt.triu(A)
            outs.append(slinalg.solve_triangular(tri + n * pt.eye(n), b, lower=lower, unit_diagonal=bool(rng.integers(2))))
            y = b if nrhs is None else b[:, 0]
            alpha = slinalg.solve_triangular(L, y, lower=lower, trans=0 if lower else 1)
            logp = -0.5 * pt.sum(alpha * alpha) - pt.sum(pt.log(pt.diagonal(L)))
            outs += [logp, *pytensor.grad(logp, [S, b])]
        # backward error eps*cond on every entry, whatever its own size
        return [A, S, b], outs, {"A": gen, "S": spd, "b": bv, "_scales": {j: ("rel", 4000.0) for j in range(len(outs))}}

    return build


def _scan(seed):
    """Scan with random structure: 0-2 sequences (forwards or backwards), a sit-sot or a two-tap mit-sot state (vector or matrix),
    an optional second state fed by the first, an optional nit-sot output, non-sequences, an inner product against a constant
    matrix in half the draws — outputs: every trajectory, the last state, and the gradient of a scalar of them (Scan's own
    gradient is a second, backwards Scan)"""
    def build():
        import pytensor
        import pytensor.tensor as pt

        rng = np.random.default_rng(41000 + seed)
        T = int(rng.choice([1, 2, 3, 7, 20, 64]))
        d = int(rng.choice([1, 3, 16, 33, 64]))
        mat = bool(rng.integers(2))  # state (B, d) instead of (d,)
        B = int(rng.choice([2, 5, 64])) if mat else None
        st = (B, d) if mat else (d,)
        nseq, two_tap, second, nit, dot = int(rng.integers(3)), bool(rng.integers(2)), bool(rng.integers(2)), bool(rng.integers(2)), bool(rng.integers(2))
        back = bool(rng.integers(2))
        seqs = [pt.tensor(f"s{i}", dtype="float64", shape=(None,) * (1 + len(st))) for i in range(nseq)]
        h0 = pt.tensor("h0", dtype="float64", shape=(None,) * (1 + len(st)))  # (taps, *st)
        g0 = pt.tensor("g0", dtype="float64", shape=(None,) * len(st))
        W = pt.tensor("W", dtype="float64", shape=(None, None))
        a = pt.scalar("a", dtype="float64")
        vals = {f"s{i}": rng.standard_normal((T,) + st) * 0.5 for i in range(nseq)}
        vals.update(h0=rng.standard_normal((2 if two_tap else 1,) + st) * 0.5, g0=rng.standard_normal(st) * 0.5,
                    W=rng.standard_normal((d, d)) / np.sqrt(d), a=np.float64(rng.uniform(0.3, 0.9)))

        def step(*args):
            args = list(args)
            xs = [args.pop(0) for _ in range(nseq)]
            hs = [args.pop(0) for _ in range(2 if two_tap else 1)]  # oldest first
            g = args.pop(0) if second else None
            Wn, an = args
            pre = hs[-1] @ Wn if dot else hs[-1] * 0.7
            for x in xs:
                pre = pre + x
            if two_tap:
                pre = pre - 0.3 * hs[0]
            h = pt.tanh(pre) * an
            res = [h]
            if second:
                res.append(g * an + pt.sqr(h))
            if nit:
                res.append(pt.sum(h, axis=-1) if mat else pt.exp(-pt.sqr(h)))
            return res

        info = [dict(initial=h0, taps=[-2, -1]) if two_tap else h0[0]]
        if second:
            info.append(g0)
        if nit:
            info.append(None)
        res, _ = pytensor.scan(step, sequences=seqs, outputs_info=info, non_sequences=[W, a], n_steps=T, go_backwards=back)
        res = res if isinstance(res, list) else [res]
        cost = sum(pt.sum(pt.sqr(r)) for r in res) + pt.sum(res[0][-1])
        wrt = [h0, W, a] + seqs + ([g0] if second else [])
        outs = res + [res[0][-1], cost, *pytensor.grad(cost, wrt, disconnected_inputs="ignore")]
        ins = seqs + [h0, g0, W, a]
        # gradients add T*d terms of either sign: the floor follows the cost, not each entry
        vals["_scales"] = {j: ("rel", 200.0) for j in range(len(outs))}
        return ins, outs, vals

    return build


def _layout_big(seed):
    """the layout family at sizes where the tiled-transpose and split-reduction kernels engage (2e5 .. 4e6 elements; the
    committed layout fuzz stays under 5000): operands contiguous / stored permuted / broadcast / every-other-element views,
    an elementwise expression, reductions of it over random axis subsets, a permuted result"""
    def build():
        import pytensor.tensor as pt

        rng = np.random.default_rng(43000 + seed)
        dt = "float64" if rng.random() < 0.6 else "float32"
        rank = int(rng.choice([2, 2, 3, 3, 4]))
        pool = [1, 2, 3, 5, 8, 17, 32, 33, 64, 100, 129, 256, 300, 513, 1000, 1025, 2048, 4099]
        while True:
            shape = [int(rng.choice(pool)) for _ in range(rank)]
            if 200_000 <= np.prod(shape) <= 4_000_000:
                break
        perm = [int(i) for i in rng.permutation(rank)]
        inv = [perm.index(i) for i in range(rank)]
        vals, ins = {}, []

        def inp(name, shp):
            v = pt.tensor(name, dtype=dt, shape=(None,) * len(shp))
            vals[name] = rng.standard_normal(shp).astype(dt)
            ins.append(v)
            return v

        a = inp("a", shape)
        b = inp("b", [shape[i] for i in perm]).transpose(inv)  # stored permuted, viewed back
        bc = [bool(rng.integers(2)) for _ in range(rank)]
        c = inp("c", [1 if f else n for f, n in zip(bc, shape)])
        c = pt.specify_broadcastable(c, *[i for i, f in enumerate(bc) if f]) if any(bc) else c
        ax = int(rng.integers(rank))
        d = inp("d", [2 * n if i == ax else n for i, n in enumerate(shape)])[tuple(slice(None, None, 2) if i == ax else slice(None) for i in range(rank))]
        e = a * b + pt.tanh(c) * d - 0.25 * pt.sqr(b)
        axes = lambda: tuple(sorted(rng.choice(rank, size=int(rng.integers(1, rank + 1)), replace=False).tolist()))  # noqa: E731
        outs = [e, pt.sum(e, axis=axes()), pt.max(a + d, axis=axes()), pt.mean(pt.sqr(e), axis=axes(), keepdims=True),
                pt.sum(pt.exp(-pt.sqr(b)) * c, axis=int(rng.integers(rank))), e.transpose(perm) * 2.0, pt.prod(1.0 + 1e-3 * b, axis=axes()),
                pt.argmax(b + c, axis=int(rng.integers(rank))), pt.sum(e), pt.min(d, axis=axes())]
        vals["_scales"] = {1: ("rel", 50.0), 3: ("rel", 50.0), 4: ("rel", 50.0), 8: ("rel", 50.0)}
        return ins, outs, vals

    return build


def _families(L, G):
    return [("multi_response", _multi_response), ("special", _special), ("linalg", _linalg), ("scan", _scan), ("layout_big", _layout_big), ("layout_f64", lambda s: L._make(s, "float64")), ("layout_f32", lambda s: L._make(s, "float32")), ("layout_i64", lambda s: L._make(s, "int64")),
            ("layout2_f64", lambda s: L._make2(s)), ("layout2_f32", lambda s: L._make2(s, "float32")), ("layout4", lambda s: L._make4(s)),
            ("glm", lambda s: G._make(s)), ("wide", lambda s: G._make_wide(s))]


def _compare(got, want, what, scale=0.0):
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape and got.dtype == want.dtype, f"{what}: {got.shape} {got.dtype} vs {want.shape} {want.dtype}"
    if want.dtype.kind in "biu":
        np.testing.assert_array_equal(got, want, err_msg=what)
        return
    f32 = want.dtype == np.float32
    fin = np.isfinite(want)
    np.testing.assert_array_equal(np.isnan(got), np.isnan(want), err_msg=what + " (NaN pattern)")
    np.testing.assert_array_equal(got[np.isinf(want)], want[np.isinf(want)], err_msg=what + " (infinities)")
    if fin.any():
        w, g = want[fin].astype(np.float64), got[fin].astype(np.float64)
        wmax = float(np.max(np.abs(w)))
        floor = scale[1] * wmax if isinstance(scale, tuple) else max(wmax, scale)  # ("rel", f): f x the largest entry
        tol = (1e-4 if f32 else 1e-10) * np.abs(w) + 64 * float(np.finfo(want.dtype).eps) * floor
        worst = float(np.max(np.abs(g - w) / np.maximum(tol, 1e-300)))
        assert worst <= 1.0, f"{what}: |err| / tol = {worst:.3g}"


@pytest.mark.parametrize("k", range(N_CASES))
def test_random_graph_hip_equals_reference_cvm(gens, k):
    pytensor, E, L, G = gens
    fams = _families(L, G)
    if ONLY:
        fams = [f for f in fams if f[0] in ONLY.split(",")]
    name, make = fams[k % len(fams)]
    seed = SEED0 + k
    ins, outs, vals = make(seed)()
    f = pytensor.function(ins, outs, mode="hip", on_unused_input="ignore")
    fc = pytensor.function(ins, outs, mode=E.reference_mode(), on_unused_input="ignore")
    args = [np.asarray(vals[v.name], dtype=v.type.dtype) for v in ins]
    want = fc(*args)
    for call in range(3):  # eager, capture, replay
        got = f(*args)
        for j, (a, b) in enumerate(zip(got, want)):
            _compare(a, b, f"{name} seed {seed} out{j} call {call}", vals.get("_scales", {}).get(j, 0.0))
