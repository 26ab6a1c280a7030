cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3t; mkdir -p $O
for i in 1 2; do timeout 120 python tools/chol_trace.py 4096 float64 2>&1 | tail -24 | tee $O/chol_trace_4096_d$i.txt; done
for i in 1 2; do timeout 120 python tools/bench_chol.py 4096 2>&1 | tail -2; done
