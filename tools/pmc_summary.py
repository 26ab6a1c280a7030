"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, as the
MI355X guide prescribes: TCC has 4 slots, FETCH_SIZE costs 3, WRITE_SIZE 2).

Corrections (MI355X_MICROARCH.md §HBM): counter unit = KiB; on gfx950 FETCH_SIZE reports
exactly 1/2 of the bytes of a wide coalesced streaming read -> doubled.  WRITE_SIZE is
uncalibrated (reported as is).

usage: python tools/pmc_summary.py <fetch.db> <write.db> <out.json>
"""
import json
import sqlite3
import sys


def per_kernel(path, counter):
    c = sqlite3.connect(path)
    q = "select name, count(*), avg(counter_value), avg(duration) from pmc_events where counter_name=? group by name"
    return {r[0]: {"calls": r[1], "avg": r[2], "avg_ns": r[3]} for r in c.execute(q, (counter,))}


def main(fetch_db, write_db, out):
    f = per_kernel(fetch_db, "FETCH_SIZE")
    w = per_kernel(write_db, "WRITE_SIZE")
    res = {}
    for k in sorted(set(f) | set(w), key=lambda k: -(f.get(k, {}).get("avg", 0))):
        fb = f.get(k, {}).get("avg", 0.0) * 1024 * 2
        wb = w.get(k, {}).get("avg", 0.0) * 1024
        res[k] = {
            "calls": f.get(k, w.get(k))["calls"],
            "fetch_bytes_corrected": fb,
            "write_bytes": wb,
            "hbm_bytes": fb + wb,
            "avg_us_profiled": f.get(k, w.get(k))["avg_ns"] / 1e3,
        }
    json.dump(res, open(out, "w"), indent=1)
    for k, v in list(res.items())[:12]:
        print(f"{v['hbm_bytes']/1e6:12.2f} MB  {v['avg_us_profiled']:9.1f} us  {k[:90]}")


if __name__ == "__main__":
    main(*sys.argv[1:4])
