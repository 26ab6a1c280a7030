cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2y
timeout 600 python tools/parity_margins.py --device sylvester_lyapunov > gpurun_out/r2y/margins6.json 2> gpurun_out/r2y/margins6.err; tail -5 gpurun_out/r2y/margins6.err
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "sylvester" 2>&1 | tail -30
