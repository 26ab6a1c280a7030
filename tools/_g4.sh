# GPU call 4 (round 3)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3d; mkdir -p $O
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_chol_blocked.py -q -x 2>&1 | tail -25 > $O/chol_tests.log
timeout 300 python tools/bench_chol.py 256 512 1024 2048 4096 > $O/chol_bench.txt 2>&1
timeout 200 python tools/overlap_probe.py > $O/overlap_probe.txt 2>&1
timeout 300 python -m pytest tests/test_gpu_refsuite_scan.py -q --timeout 120 --tb=short -p no:cacheprovider -k "test_pushforward and not _2 and not mitmot" 2>&1 | grep -v Warning | tail -80 > $O/refscan_tb.log
timeout 300 python -m pytest tests/test_gpu_e2e.py tests/test_coherence.py -q -x 2>&1 | tail -15 > $O/e2e.log
cat $O/chol_bench.txt; tail -3 $O/chol_tests.log; cat $O/overlap_probe.txt; tail -40 $O/refscan_tb.log; tail -4 $O/e2e.log
