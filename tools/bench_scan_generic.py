"""What a Scan step costs when the inner graph is NOT the GRU's product + epilogue shape: three of the reference's own
Scan benchmarks (tests/benchmarks/test_scan.py) under ``mode="hip"`` next to the reference C linker on this host.

  vector_taps   1000 steps, a mit-sot (taps -2, -1) and a sit-sot scalar state, two sequences   (test_scan.py:131-174)
  sit_sot_256   512 steps of x <- x + 1 on a 256-vector, whole trace kept                        (test_scan.py:184-250)
  seir_logp     1200 steps, three scalar states, two int32 sequences, 6 gammaln per step         (test_scan.py:27-103)

Needs the importable reference copy (oracle/_ref).  usage: python tools/bench_scan_generic.py
"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import make_ref
make_ref.activate()
import pytensor
import pytensor.tensor as pt
from pytensor import scan
from pytensor.compile.mode import Mode
import pytensor_amd
pytensor_amd.register()


def vector_taps():
    n = 1000
    seq1, seq2 = pt.vector("seq1", dtype="float64", shape=(n,)), pt.vector("seq2", dtype="float64", shape=(n,))
    mi, si = pt.vector("mitsot_init", dtype="float64", shape=(2,)), pt.scalar("sitsot_init", dtype="float64")

    def step(s1, s2, m1, m2, s):
        m3 = (m1 + s2 + m2 + s1) / np.sqrt(4)
        return m3, (s + m3) / np.sqrt(2)

    outs = scan(fn=step, sequences=[seq1, seq2], outputs_info=[dict(initial=mi, taps=[-2, -1]), dict(initial=si, taps=[-1])], return_updates=False)
    rng = np.random.default_rng(474)
    return [seq1, seq2, mi, si], outs, [rng.normal(size=n), rng.normal(size=n), rng.normal(size=2), np.asarray(rng.normal())], n


def sit_sot_256():
    n, size = 512, 256
    x0 = pt.vector("x0", shape=(size,), dtype="float64")
    xs = scan(fn=lambda x: x + 1, outputs_info=[x0], n_steps=n - 1, return_updates=False)
    return [x0], [xs], [np.zeros(size)], n - 1


def seir_logp():
    def binomln(n, k):
        return pt.gammaln(n + 1) - pt.gammaln(k + 1) - pt.gammaln(n - k + 1)

    def blp(n, p, value):
        return binomln(n, value) + value * pt.log(p) + (n - value) * pt.log(1 - p)

    C_t, D_t = pt.vector("C_t", dtype="int32", shape=(1200,)), pt.vector("D_t", dtype="int32", shape=(1200,))
    st0, et0, it0 = pt.scalar("s_t0"), pt.scalar("e_t0"), pt.scalar("i_t0")
    beta, gamma, delta = pt.scalar("beta"), pt.scalar("gamma"), pt.scalar("delta")

    def one(ct0, dt0, st0, et0, it0, beta, gamma, delta):
        bt0 = (st0 * beta).astype(st0.dtype)
        return st0 - bt0, et0 + bt0 - ct0, it0 + ct0 - dt0, blp(et0, gamma, ct0), blp(it0, delta, dt0)

    st, et, it, lc, ld = scan(fn=one, sequences=[C_t, D_t], outputs_info=[st0, et0, it0, None, None], non_sequences=[beta, gamma, delta], return_updates=False)
    vals = [np.array([3, 5, 8, 13, 21, 26, 10, 3] * 150, dtype=np.int32), np.array([1, 2, 3, 7, 9, 11, 5, 1] * 150, dtype=np.int32),
            np.array(100.0), np.array(50.0), np.array(25.0), np.array(0.277792), np.array(0.135330), np.array(0.108753)]
    return [C_t, D_t, st0, et0, it0, beta, gamma, delta], [lc.sum() + ld.sum(), st, et, it], vals, 1200


def timeit(f, vals, reps):
    f(*vals)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        f(*vals)
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


def main():
    for name, build in (("vector_taps", vector_taps), ("sit_sot_256", sit_sot_256), ("seir_logp", seir_logp)):
        ins, outs, vals, steps = build()
        fh = pytensor.function(ins, outs, mode="hip")
        fc = pytensor.function(ins, outs, mode=Mode(linker="cvm" if pytensor.config.cxx else "py", optimizer="fast_run"))
        fh.trust_input = fc.trust_input = True
        a, b = fh(*vals), fc(*vals)
        for x, y in zip(a, b):
            np.testing.assert_allclose(x, y, rtol=1e-10, atol=1e-12, equal_nan=True)
        for _ in range(3):
            fh(*vals)  # (the second call captures the plan)
        th, tc = timeit(fh, vals, 7), timeit(fc, vals, 7)
        exe = fh.vm.jit_fn
        scans = [n for n in exe.graph.nodes if n.op in ("Scan", "ScanLoop")]
        inner = [m.op for n in scans if n.op == "Scan" for m in n.params["inner"].nodes]
        print(json.dumps({"scan": name, "steps": steps, "hip_ms": round(th * 1e3, 3), "hip_us_per_step": round(th / steps * 1e6, 2), "cvm_ms": round(tc * 1e3, 3),
                          "cvm_us_per_step": round(tc / steps * 1e6, 2), "hip_over_cvm": round(th / tc, 2), "outer_ops": [n.op for n in scans], "inner_ops": inner,
                          "replays": exe.stats.get("replays")}), flush=True)


if __name__ == "__main__":
    main()
