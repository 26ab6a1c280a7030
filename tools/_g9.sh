export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2l
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_collective.py tests/test_gpu_parity.py -q -rs -k "collective or gemv or Gemv or c3 or blas or rccl or identity or ranks" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -8 $O/pytest.log
cd /tmp
timeout 200 python $R/tools/bench_gemv.py 2>&1 | grep -v Warn | tee $O/gemv.txt
