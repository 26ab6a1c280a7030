cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3y; mkdir -p $O
timeout 300 python tools/bench_gp.py 512 2048 4096 2>&1 | tail -3 | tee $O/gp_bench.txt
echo "launch-per-step Cholesky + one-thread solves (the round's starting point):" | tee -a $O/gp_bench.txt
PTHIP_CHOL=steps PTHIP_TRSM=generic timeout 300 python tools/bench_gp.py 2048 2>&1 | tail -1 | tee -a $O/gp_bench.txt
