cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3v; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_lu_blocked.py -x -q --timeout 300 -p no:cacheprovider 2>&1 | grep -v "Warning\|warnings.warn" | tail -6
timeout 300 python tools/bench_getrf.py 256 512 1024 2048 4096 2>&1 | grep float64 | tee $O/getrf_bench_r4.txt
