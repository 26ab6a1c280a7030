cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3y; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_chol_blocked.py tests/test_gpu_trsm_blocked.py tests/test_gpu_lu_blocked.py -x -q --timeout 300 -p no:cacheprovider 2>&1 | grep -v "Warning\|warnings.warn" | tail -6
timeout 900 python -m pytest tests/test_gpu_refsuite_linalg.py -x -q --timeout 300 -p no:cacheprovider -k "olve or inv or Inv or lu or LU" 2>&1 | grep -v "Warning\|warnings.warn" | tail -4
timeout 100 python tools/bench_lu.py 128 2>&1 | tail -1 | tee $O/lu128.txt
PTHIP_TRSM=lds timeout 100 python tools/bench_lu.py 128 2>&1 | tail -1 | tee -a $O/lu128.txt
