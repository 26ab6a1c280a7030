"""MFMA utilisation of the GEMM kernels from rocprofv3 PMC passes.

usage: rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d DIR -o x -- python tools/bench_gemm.py 0 4
       python tools/pmc_mfma.py <results.db>

SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD with the matrix pipe busy (summed over the
counter instances rocprofv3 reports); utilisation = busy / (kernel duration in shader cycles x
1024 SIMDs) — the gfx94x `MfmaUtil` formula written out (ROCm 7.2 ships no gfx950 derived metrics).
"""
import json
import sqlite3
import sys

CLOCK_GHZ = 2.4
SIMDS = 256 * 4
FILTER = ["gemm"]


def main(path):
    c = sqlite3.connect(path)
    rows = c.execute(
        "select name, counter_name, sum(counter_value), count(distinct dispatch_id), avg(duration) from pmc_events group by name, counter_name"
    ).fetchall()
    out = {}
    for name, counter, total, ndisp, dur_ns in rows:
        if not any(t in name for t in FILTER):
            continue
        k = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:48]
        d = out.setdefault(k, {"dispatches": ndisp, "avg_us": round(dur_ns / 1e3, 1)})
        d[counter] = total / max(ndisp, 1)
    for k, d in out.items():
        cyc = d["avg_us"] * 1e-6 * CLOCK_GHZ * 1e9
        if "SQ_VALU_MFMA_BUSY_CYCLES" in d:
            d["mfma_util"] = round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * SIMDS), 3)  # of the time at the 2.4 GHz peak clock
            if d.get("GRBM_GUI_ACTIVE"):
                # GRBM_GUI_ACTIVE: shader-clock cycles the GPU was active during the dispatch = the clock it actually ran at
                d["clock_ghz"] = round(d["GRBM_GUI_ACTIVE"] / (d["avg_us"] * 1e3), 3)
                d["mfma_util_at_actual_clock"] = round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / (d["GRBM_GUI_ACTIVE"] * SIMDS), 3)
        print(json.dumps({k: d}))


if __name__ == "__main__":
    if len(sys.argv) > 2:
        FILTER[:] = sys.argv[2].split(",")
    main(sys.argv[1])
