cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2y
timeout 600 python tools/parity_margins.py --device ifelse_lazy index_layout_misc > gpurun_out/r2y/margins2.json 2> gpurun_out/r2y/margins2.err; tail -5 gpurun_out/r2y/margins2.err
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_plan.py -q -k "ifelse_lazy or index_layout" 2>&1 | tail -30
