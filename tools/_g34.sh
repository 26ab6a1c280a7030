cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3u; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_trsm_blocked.py -x -q --timeout 120 -p no:cacheprovider 2>&1 | grep -v "Warning\|warnings.warn" | tail -25 > $O/trsm_tests.log
tail -12 $O/trsm_tests.log
timeout 300 python tools/bench_trsm.py 2048 4096 2>&1 | tail -20 | tee $O/trsm_bench.txt
echo generic; PTHIP_TRSM=generic timeout 300 python tools/bench_trsm.py 2048 2>&1 | tail -8 | tee $O/trsm_bench_generic.txt
