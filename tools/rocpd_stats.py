"""Summarise a rocprofv3 rocpd SQLite database into a per-kernel stats table.

usage: python tools/rocpd_stats.py <results.db> [> profiles/<name>.md]
"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    namec = "name" if "name" in cols else "kernel_name"
    rows = c.execute(
        f"select {namec}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by {namec} order by 3 desc"
    ).fetchall()
    tot = sum(r[2] for r in rows) or 1
    print(f"| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for n, cnt, s, a, mn, mx in rows:
        n = n if len(n) < 110 else n[:107] + "..."
        print(f"| `{n}` | {cnt} | {s/1e6:.3f} | {a/1e3:.2f} | {mn/1e3:.2f} | {mx/1e3:.2f} | {100*s/tot:.1f} |")


if __name__ == "__main__":
    main(sys.argv[1])
