cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2y
timeout 600 python tools/parity_margins.py --device expm_and_grad choose_permute_conv2d > gpurun_out/r2y/margins3.json 2> gpurun_out/r2y/margins3.err; tail -5 gpurun_out/r2y/margins3.err
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "expm_and_grad or choose_permute" 2>&1 | tail -30
