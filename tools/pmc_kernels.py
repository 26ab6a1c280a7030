"""Per-kernel hardware counters: run a command under several ``rocprofv3 --pmc`` passes (one small
counter group per pass — the guide's rule; never combined with trace domains) and print a
markdown table with one row per kernel and one column per counter (mean per dispatch; counters
that are per-XCD / per-SE instances are summed over instances).

usage: [PMC_GROUPS=2,8,9] python tools/pmc_kernels.py <out.md> <kernel-name-substring[,substring...]> -- <command ...>
(also writes <out>.json)
"""
import glob
import os
import shutil
import sqlite3
import subprocess
import sys
import tempfile

GROUPS = [
    ["SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY"],
    ["SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM", "SQ_WAIT_ANY"],
    ["SQ_INSTS_VALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS", "SQ_INSTS_MFMA"],
    ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INST_LEVEL_VMEM", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_WR"],
    ["TCC_HIT_sum", "TCC_MISS_sum"],
    ["TCC_REQ_sum", "TCC_EA0_RDREQ_sum"],
    ["TCP_TCC_READ_REQ_sum", "TCP_TCC_READ_REQ_LATENCY_sum"],
    ["TCP_PENDING_STALL_CYCLES_sum", "TA_BUSY_avr"],
    ["FETCH_SIZE"],
    ["WRITE_SIZE"],
    ["GRBM_GUI_ACTIVE", "MeanOccupancyPerCU"],
]


def one_pass(counters, cmd):
    d = tempfile.mkdtemp(prefix="pthip_pmc_", dir="/tmp")
    try:
        r = subprocess.run(["rocprofv3", "--pmc", *counters, "-d", d, "-o", "p", "--", *cmd], cwd="/tmp",
                           env={**os.environ, "TMPDIR": "/tmp"}, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=240)
        dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
        if not dbs:
            return None, f"no database (rc={r.returncode}): {r.stderr.decode(errors='replace')[-300:]}"
        con = sqlite3.connect(dbs[0])
        # one row per (dispatch, counter instance): sum the instances of a dispatch, then average dispatches
        q = ("select name, counter_name, avg(v), count(*), avg(dur) from (select name, counter_name, dispatch_id, sum(counter_value) v, avg(duration) dur "
             "from pmc_events group by name, counter_name, dispatch_id) group by name, counter_name")
        try:
            rows = con.execute(q).fetchall()
        except sqlite3.OperationalError:
            q = "select name, counter_name, avg(counter_value), count(*), avg(duration) from pmc_events group by name, counter_name"
            rows = con.execute(q).fetchall()
        return rows, None
    except Exception as e:  # noqa: BLE001
        return None, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(d, ignore_errors=True)


def main():
    out, subs = sys.argv[1], sys.argv[2].split(",")
    cmd = [os.path.abspath(a) if os.path.exists(a) and not os.path.isabs(a) else a for a in sys.argv[sys.argv.index("--") + 1:]]
    out = os.path.abspath(out)
    table, calls, notes = {}, {}, []
    sel = os.environ.get("PMC_GROUPS")  # e.g. "2,8,9": only these counter groups (one rocprofv3 pass each)
    groups = [GROUPS[int(i)] for i in sel.split(",")] if sel else GROUPS
    for g in groups:
        rows, err = one_pass(g, cmd)
        if rows is None:
            notes.append(f"pass {' '.join(g)}: {err}")
            continue
        for name, ctr, v, n, dur in rows:
            if any(s in name for s in subs):
                table.setdefault(name, {})[ctr] = v
                calls[name] = n
                table[name].setdefault("us_profiled", (dur or 0) / 1e3)
    cols = ["us_profiled"] + [c for g in GROUPS for c in g]
    with open(out, "w") as fh:
        fh.write("command: `" + " ".join(cmd) + "`\n\n")
        for name, vals in table.items():
            fh.write(f"### `{name}`  ({calls[name]} dispatches per pass)\n\n| counter | mean per dispatch |\n|---|---:|\n")
            for c in cols:
                if c in vals:
                    fh.write(f"| {c} | {vals[c]:.4g} |\n")
            fh.write("\n")
        for n in notes:
            fh.write(f"- {n}\n")
    import json

    with open(os.path.splitext(out)[0] + ".json", "w") as fh:  # the same numbers for tools that read them (bench_configs._alu_roofline)
        json.dump({name: {**vals, "dispatches": calls[name]} for name, vals in table.items()}, fh, indent=1)
    print(open(out).read())


if __name__ == "__main__":
    main()
