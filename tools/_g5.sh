# GPU call 5 (round 3): Cholesky n=2048 fp64 profile, full refscan module, dotew tests
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3e; mkdir -p $O
export TMPDIR=/tmp
cd /tmp; rocprofv3 --kernel-trace --stats -d /tmp/pk_chol -o k -- python $GRAFT_REPO_ROOT/tools/bench_chol.py 2048 > $GRAFT_REPO_ROOT/$O/chol2048_under_rocprof.txt 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/pk_chol -name "*.db" | head -1) > $GRAFT_REPO_ROOT/$O/chol2048_kernel_stats.md 2>&1
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_refsuite_scan.py -q --timeout 120 -rA --tb=line -p no:cacheprovider 2>&1 | grep -E "^(FAILED|ERROR|SKIPPED|XFAIL|XPASS)|passed|failed|^/|Error" | cut -c1-400 > $O/refscan.log
timeout 300 python -m pytest tests/test_gpu_dotew.py -q -x 2>&1 | tail -5 > $O/dotew_tests.log
head -16 $O/chol2048_kernel_stats.md; tail -12 $O/refscan.log; tail -3 $O/dotew_tests.log
