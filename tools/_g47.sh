cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3u; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_trsm_blocked.py tests/test_gpu_lu_blocked.py -x -q --timeout 120 -p no:cacheprovider 2>&1 | grep -v "Warning\|warnings.warn" | tail -8
timeout 300 python tools/bench_trsm.py 2048 4096 2>&1 | grep -v '"nrhs": 1,' | tee $O/trsm_bench2.txt
