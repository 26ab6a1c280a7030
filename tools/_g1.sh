cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_decomp.py tests/test_gpu_parity.py -q -k "qr or svd or pinv or decomp" 2>&1 | tail -3
timeout 600 python tools/_svdtime.py 2>&1 | tail -12
