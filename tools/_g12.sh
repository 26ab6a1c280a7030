export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2o
mkdir -p $O
cd $R
timeout 600 python tools/parity_margins.py --device lu_reuse solve_sym_eigh_generalised indexing_nd > $O/margins.json 2> $O/margins.err; tail -3 $O/margins.err
python - <<P
import json
m=json.load(open("$O/margins.json"))
for name,rec in m.items():
    if "error" in rec: print(name,"ERROR",rec["error"]); continue
    print(name,[(k,round(x["over"],2),"%.2e"%x["max_abs"],"%.2e"%x["scale"],x["n_over"],x["n"]) if x else (k,"int") for k,x in enumerate(rec["hip_vs_cvm"])])
P
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_plan.py -q -x -k "lu_reuse or solve or eigh or general or blockwise or indexing" 2>&1 | tail -6
