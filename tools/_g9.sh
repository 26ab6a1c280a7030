# GPU call 9 (round 3)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3i; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_chol_blocked.py -q -x 2>&1 | tail -8 > $O/chol_tests.log
(for n in 512 1024 2048 4096; do timeout 120 python tools/bench_chol.py $n; done; echo "scalar panel solve:"; PTHIP_CHOL_TRSM=scalar timeout 120 python tools/bench_chol.py 4096) > $O/chol_bench.txt 2>&1
cd /tmp; rocprofv3 --kernel-trace --stats -d /tmp/pk_chol -o k -- python $GRAFT_REPO_ROOT/tools/bench_chol.py 4096 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/pk_chol -name "*.db" | head -1) > $GRAFT_REPO_ROOT/$O/chol4096_kernel_stats.md 2>&1
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_fullsize.py -q -x -k c5 -s 2>&1 | grep -E "c5 out|worst|passed|failed|Error|assert" | head -20 > $O/c5_full.log
timeout 300 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_dotew.py -q -x 2>&1 | tail -5 > $O/e2e.log
timeout 900 python -m pytest tests/test_gpu_refsuite_linalg.py -q --timeout 120 -rf --tb=line -p no:cacheprovider 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed|^/|Error" | cut -c1-300 | tail -40 > $O/reflinalg.log
tail -4 $O/chol_tests.log; cat $O/chol_bench.txt; head -12 $O/chol4096_kernel_stats.md; cat $O/c5_full.log; tail -3 $O/e2e.log; tail -25 $O/reflinalg.log
