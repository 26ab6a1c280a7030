# GPU call 7 (round 3): Cholesky after the staging fix; reference linalg/special modules under hip
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3g; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_chol_blocked.py -q -x 2>&1 | tail -4 > $O/chol_tests.log
(for n in 256 512 1024 2048 4096; do timeout 120 python tools/bench_chol.py $n; done) > $O/chol_bench.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_refsuite_linalg.py -q --timeout 120 -rA --tb=line -p no:cacheprovider 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed|^/|Error" | cut -c1-300 > $O/reflinalg.log
tail -3 $O/chol_tests.log; cat $O/chol_bench.txt; grep -c "^FAILED" $O/reflinalg.log; tail -3 $O/reflinalg.log
