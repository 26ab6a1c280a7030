cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3r; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_refsuite_index.py -q --timeout 120 -rf --tb=short -p no:cacheprovider 2>&1 | grep -v "Warning\|warnings.warn" | tail -60 > $O/refindex.log
tail -50 $O/refindex.log
