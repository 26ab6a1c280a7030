"""How the hoisted sequence products of config #5 and the Scan's step kernels share the chip in one plan replay.

Reads a rocprofv3 rocpd database of `tools/bench_configs.py c5 --no-check` and, for the fastest replay (the
tightest window of 2 x T step kernels), prints: the step kernels' median duration and start-to-start period while a GEMM is in flight
and while none is, the GEMM durations inside and outside the loop, and the queues involved.

usage: python tools/c5_overlap_timeline.py <results.db> [T=1000]
"""
import sqlite3
import statistics as st
import sys


def main(path, T=1000):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    namec = "name" if "name" in cols else "kernel_name"
    qc = next((q for q in ("queue_id", "stream_id", "queue") if q in cols), None)
    rows = c.execute(f"select {namec}, start, end" + (f", {qc}" if qc else ", 0") + " from kernels order by start").fetchall()
    allsteps = [r for r in rows if r[0].startswith("dotew_")]
    # the fastest window of 2T consecutive step kernels = one plan replay (the eager passes run at the tracer's pace)
    n = 2 * T
    i0 = min(range(0, len(allsteps) - n + 1), key=lambda i: allsteps[i + n - 1][1] - allsteps[i][1])
    steps = allsteps[i0:i0 + n]
    t0, t1 = steps[0][1], steps[-1][2]
    gemms = [r for r in rows if "gemm" in r[0] and r[2] > t0 - 2_000_000 and r[1] < t1]
    inside = [g for g in gemms if g[2] > t0]
    print(f"replay window {(t1 - t0) / 1e3:.1f} us = {(t1 - t0) / 1e3 / T:.2f} us/step; {len(gemms)} GEMM launches near it, {len(inside)} overlapping the loop")
    print(f"queues: steps {sorted({s[3] for s in steps})}, gemms {sorted({g[3] for g in gemms})}")

    def busy(a, b):
        return any(g[1] < b and g[2] > a for g in inside)

    for label, sel in (("GEMM in flight", True), ("no GEMM", False)):
        d, p = [], []
        for i in range(1, len(steps)):
            s = steps[i]
            if busy(steps[i - 1][1], s[2]) == sel:
                d.append((s[2] - s[1]) / 1e3)
                p.append((s[1] - steps[i - 1][1]) / 1e3)
        if d:
            print(f"steps, {label}: n={len(d)}  duration median {st.median(d):.2f} us  start-to-start median {st.median(p):.2f} us  (2 kernels per step -> {2 * st.median(p):.2f} us/step)")
    for label, gs in (("overlapping the loop", inside), ("before the loop", [g for g in gemms if g[2] <= t0])):
        if gs:
            ds = [(g[2] - g[1]) / 1e3 for g in gs]
            print(f"GEMMs {label}: n={len(gs)} median {st.median(ds):.1f} us  min {min(ds):.1f}  max {max(ds):.1f}  total {sum(ds):.0f} us")
    # the first chunk boundary: what the step stream does while it waits
    gaps = sorted(((steps[i][1] - steps[i - 1][2]) / 1e3, i) for i in range(1, len(steps)))[-6:]
    print("largest gaps between consecutive step kernels (us, index):", [(round(g, 1), i) for g, i in gaps])


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1000)
