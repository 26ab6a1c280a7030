export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2m
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_collective.py -q -rs > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -8 $O/pytest.log
