cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_e2e.py -q -k "library_functions" 2>&1 | tail -30
