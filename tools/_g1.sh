cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2z
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r2z/pytest2.log; cat gpurun_out/r2z/pytest2.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/profile_round.sh r2z > gpurun_out/r2z/profile_round.log 2>&1; tail -5 gpurun_out/r2z/profile_round.log
