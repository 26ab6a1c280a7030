cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3y; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_chol_blocked.py -x -q --timeout 300 -p no:cacheprovider 2>&1 | grep -v "Warning\|warnings.warn" | tail -6
timeout 600 python -m pytest tests/test_gpu_refsuite_linalg.py -x -q --timeout 300 -p no:cacheprovider -k "holesky or cho_solve or Cho" 2>&1 | grep -v "Warning\|warnings.warn" | tail -3
for n in 512 2048 4096; do timeout 100 python tools/bench_chol.py $n 2>&1 | tail -2; done | tee $O/chol_bench_direct.txt
