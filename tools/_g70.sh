cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3y; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_trsm_blocked.py -x -q --timeout 120 -p no:cacheprovider 2>&1 | grep -v "Warning\|warnings.warn" | tail -4
timeout 100 python tools/bench_trsm.py 2048 4096 2>&1 | grep '"nrhs": 1,' | grep float64 | tee $O/trsv_bench2.txt
