export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2j
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -E "FAILED|ERROR|passed|failed|rc=" $O/pytest.log | head -40
