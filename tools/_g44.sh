cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3w; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_properties.py tests/test_gpu_chol_blocked.py tests/test_gpu_errors.py -x -q --timeout 300 -p no:cacheprovider -k "chol or Chol or potrf or linalg or solve or Solve or nan or NaN" 2>&1 | grep -v "Warning\|warnings.warn" | tail -8
timeout 900 python -m pytest tests/test_gpu_refsuite_linalg.py -x -q --timeout 300 -p no:cacheprovider -k "holesky or cho_solve or Cho" 2>&1 | grep -v "Warning\|warnings.warn" | tail -5
for n in 32 64 100 128 141; do timeout 100 python tools/bench_chol.py $n 2>&1 | grep float64; done | tee $O/potrf_lds_bench.txt
timeout 100 python tools/bench_linalg.py 128 200 2>&1 | tail -3 | tee $O/linalg128.txt
