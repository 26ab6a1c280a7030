"""Print the kernel timeline (start offset, duration, queue) of N consecutive dispatches in a
rocprofv3 rocpd database — used to see how a multi-stream plan replay actually overlaps.

usage: python tools/rocpd_timeline.py <results.db> [N=45] [skip_from_end=0]
"""
import sqlite3
import sys


def main(path, n=45, skip=0):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    namec = "name" if "name" in cols else "kernel_name"
    qc = next((q for q in ("queue_id", "stream_id", "queue") if q in cols), None)
    sel = f"select {namec}, start, end" + (f", {qc}" if qc else ", 0") + " from kernels order by start desc limit ? offset ?"
    rows = list(reversed(c.execute(sel, (n, skip)).fetchall()))
    t0 = rows[0][1]
    print("| start us | end us | dur us | queue | kernel |")
    print("|---:|---:|---:|---:|---|")
    for name, s, e, q in rows:
        print(f"| {(s - t0) / 1e3:.1f} | {(e - t0) / 1e3:.1f} | {(e - s) / 1e3:.1f} | {q} | `{name[:70]}` |")


if __name__ == "__main__":
    a = sys.argv
    main(a[1], int(a[2]) if len(a) > 2 else 45, int(a[3]) if len(a) > 3 else 0)
