cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2y
timeout 600 python tools/parity_margins.py --device qr_modes svd_pinv_lstsq tridiagonal_solve linalg_misc > gpurun_out/r2y/margins.json 2> gpurun_out/r2y/margins.err; tail -5 gpurun_out/r2y/margins.err
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "qr_modes or svd_pinv or tridiagonal or linalg_misc" 2>&1 | tail -30
