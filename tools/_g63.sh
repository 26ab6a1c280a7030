cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3y; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_trsm_blocked.py tests/test_gpu_lu_blocked.py tests/test_gpu_decomp.py -x -q --timeout 120 -p no:cacheprovider 2>&1 | grep -v "Warning\|warnings.warn" | tail -4
timeout 200 python tools/bench_trsm.py 2048 4096 2>&1 | grep -v '"nrhs": 1,' | grep float64 | tee $O/trsm_bench4.txt
