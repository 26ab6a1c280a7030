#!/bin/bash
# per-kernel cold timing of the round-6 log-sum-exp / column-softmax kernels: rocprofv3 --kernel-trace over tools/bench_lse_kernels.py,
# one row per (kernel, grid) -> gpurun_out/r7f/lse_kernels.md (kept as profiles/r7_lse_kernels.md)
export TMPDIR=/tmp; cd /tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r7f; mkdir -p $O
rocprofv3 --kernel-trace -d /tmp/pl -o k -- python $R/tools/bench_lse_kernels.py 12 > $O/run.log 2>&1
python - <<'PY' > $O/lse_kernels.md
import sqlite3, glob, statistics
db = glob.glob('/tmp/pl/**/*.db', recursive=True)[0]
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
namec = "name" if "name" in cols else "kernel_name"
gx = next((q for q in ("grid_size_x","grid_x","workgroup_count_x") if q in cols), None)
print(cols)
rows = c.execute(f"select {namec}, start, end, {gx if gx else 0}, {'grid_size_y' if 'grid_size_y' in cols else 0} from kernels order by start").fetchall()
per = {}
for n, s, e, g, gy in rows:
    per.setdefault((n[:90], g, gy), []).append((e - s) / 1e3)
print("| kernel | grid x | grid y | calls | median us | min us |")
print("|---|---:|---:|---:|---:|---:|")
for (n, g, gy), v in per.items():
    print(f"| `{n}` | {g} | {gy} | {len(v)} | {statistics.median(v):.2f} | {min(v):.2f} |")
PY
cat $O/lse_kernels.md | cut -c1-220
