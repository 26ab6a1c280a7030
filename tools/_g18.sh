export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2u
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_refsuite.py -x -q -k "gemm or dot or bdot or c3 or c5 or Gemm or blas or batched" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -E "FAILED|ERROR|passed|failed|rc=" $O/pytest.log | head
cd /tmp
timeout 300 python $R/tools/bench_gemm.py 0 1 2 3 6 2>&1 | tee $O/sgemm.txt
