cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3u; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_trsm_blocked.py -x -q --timeout 120 -p no:cacheprovider 2>&1 | grep -v "Warning\|warnings.warn" | tail -25 > $O/trsm_tests.log
tail -6 $O/trsm_tests.log
timeout 300 python tools/bench_trsm.py 512 2048 4096 8192 2>&1 | grep '"nrhs": 1,' | tee $O/trsv_bench.txt
