cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_misc.py -q 2>&1 | tail -20
