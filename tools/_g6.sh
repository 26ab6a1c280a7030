export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2i
mkdir -p $O
cd $R
timeout 600 python tools/parity_margins.py --device > $O/margins.json 2> $O/margins.err; tail -2 $O/margins.err
python - <<P
import json
m=json.load(open("$O/margins.json"))
for name,rec in m.items():
    if "error" in rec: print(name,"ERROR",rec["error"]); continue
    bad=[(k,round(x["over"],1),"%.2e"%x["max_abs"],"%.2e"%x["scale"],x["n_over"],x["n"]) for k,x in enumerate(rec["hip_vs_cvm"]) if x and x["over"]>1]
    if bad: print(name,bad[:8])
P
