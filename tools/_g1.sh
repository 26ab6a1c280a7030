cd $GRAFT_REPO_ROOT
timeout 300 python tools/_c1calls.py 2>&1 | grep -v "^$" | head -3
timeout 600 python tools/bench_configs.py c1 --reps 10 --no-check 2>&1 | grep -o '"ms_device": [0-9.]*, "ms_call": [0-9.]*'
timeout 900 python -m pytest tests/test_gpu_plan.py tests/test_gpu_e2e.py -x -q 2>&1 | tail -5
