cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3t; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_chol_blocked.py -x -q --timeout 120 -p no:cacheprovider 2>&1 | grep -v "Warning\|warnings.warn" | tail -25 > $O/chol_tests.log
tail -5 $O/chol_tests.log
timeout 120 python tools/chol_trace.py 4096 float64 2>&1 | tail -16 | tee $O/chol_trace_4096_b.txt
for n in 512 1024 2048 4096; do timeout 120 python tools/bench_chol.py $n 2>&1 | tail -2; done | tee $O/chol_bench_dag_b.txt
