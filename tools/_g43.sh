cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3v; mkdir -p $O
timeout 300 python tools/bench_getrf.py 2>&1 | tail -10 | tee $O/getrf_bench.txt
timeout 600 python -m pytest tests/test_gpu_trsm_blocked.py tests/test_gpu_chol_blocked.py tests/test_gpu_decomp.py -x -q --timeout 120 -p no:cacheprovider 2>&1 | grep -v "Warning\|warnings.warn" | tail -5
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r3v/prof_lu -o lu -- python $GRAFT_REPO_ROOT/tools/bench_getrf.py 2048 > /dev/null 2>&1; ls $GRAFT_REPO_ROOT/gpurun_out/r3v/prof_lu | head
