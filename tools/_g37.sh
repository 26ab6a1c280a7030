cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3v; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_lu_blocked.py -x -q --timeout 300 -p no:cacheprovider 2>&1 | grep -v "Warning\|warnings.warn" | tail -30 > $O/lu_tests.log
tail -30 $O/lu_tests.log
for n in 256 512 1024 2048; do timeout 200 python tools/bench_lu.py $n 2>&1 | tail -1; done | tee $O/lu_sizes.txt
