"""Summarise a rocprofv3 rocpd SQLite database into a per-kernel stats table (the median is the
figure to quote: the first launch of a kernel carries code-object load / cold caches and can be
two orders of magnitude above the rest).

usage: python tools/rocpd_stats.py <results.db> [> profiles/<name>.md]
"""
import sqlite3
import statistics
import sys


def main(path):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    namec = "name" if "name" in cols else "kernel_name"
    per = {}
    for n, s, e in c.execute(f"select {namec}, start, end from kernels"):
        per.setdefault(n, []).append(e - s)
    rows = sorted(per.items(), key=lambda kv: -sum(kv[1]))
    tot = sum(sum(v) for _, v in rows) or 1
    print("| kernel | calls | total ms | avg us | median us | min us | max us | % |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|")
    for n, v in rows:
        n = n if len(n) < 110 else n[:107] + "..."
        print(f"| `{n}` | {len(v)} | {sum(v)/1e6:.3f} | {sum(v)/len(v)/1e3:.2f} | {statistics.median(v)/1e3:.2f} | {min(v)/1e3:.2f} | {max(v)/1e3:.2f} | {100*sum(v)/tot:.1f} |")


if __name__ == "__main__":
    main(sys.argv[1])
