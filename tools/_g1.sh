cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_decomp.py tests/test_random.py -q -m gpu -k "blockwise or multivariate" 2>&1 | tail -20
