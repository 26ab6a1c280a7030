cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2y
timeout 600 python tools/parity_margins.py --device special_inverse_polygamma > gpurun_out/r2y/margins5.json 2> gpurun_out/r2y/margins5.err; tail -5 gpurun_out/r2y/margins5.err
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "special_inverse" 2>&1 | tail -30
