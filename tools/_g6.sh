# GPU call 6 (round 3): Cholesky n=2048 timeline, truncated-gradient test, refscan module, e2e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3f; mkdir -p $O
export TMPDIR=/tmp
cd /tmp; rocprofv3 --kernel-trace -d /tmp/pk_chol -o k -- python $GRAFT_REPO_ROOT/tools/bench_chol.py 2048 > $GRAFT_REPO_ROOT/$O/chol2048_under_rocprof.txt 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_timeline.py $(find /tmp/pk_chol -name "*.db" | head -1) 140 > $GRAFT_REPO_ROOT/$O/chol2048_timeline.md 2>&1
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_refsuite_scan.py -q --timeout 120 --tb=short -p no:cacheprovider 2>&1 | tail -15 > $O/refscan.log
timeout 300 python -m pytest tests/test_gpu_e2e.py -q -x -k "truncated or scan" 2>&1 | tail -5 > $O/e2e.log
tail -100 $O/chol2048_timeline.md; tail -6 $O/refscan.log; tail -3 $O/e2e.log
