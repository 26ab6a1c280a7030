cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_refsuite_linalg.py tests/test_gpu_e2e.py -x -q --timeout 300 -p no:cacheprovider 2>&1 | grep -v "Warning\|warnings.warn" | tail -4
