cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3w; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_properties.py tests/test_gpu_chol_blocked.py tests/test_gpu_errors.py tests/test_gpu_parity.py -x -q --timeout 300 -p no:cacheprovider 2>&1 | grep -v "Warning\|warnings.warn" | tail -6
timeout 900 python -m pytest tests/test_gpu_refsuite_linalg.py tests/test_gpu_e2e.py -x -q --timeout 300 -p no:cacheprovider 2>&1 | grep -v "Warning\|warnings.warn" | tail -4
