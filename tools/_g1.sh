cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2z
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -12 > gpurun_out/r2z/pytest.log; cat gpurun_out/r2z/pytest.log
timeout 600 python bench.py --steps 300 --warmup 30 > gpurun_out/r2z/bench.json 2> gpurun_out/r2z/bench.err; tail -c 1500 gpurun_out/r2z/bench.json
