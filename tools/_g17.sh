export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2t
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_refsuite.py tests/test_gpu_e2e.py -x -q -k "gemm or dot or bdot or c3 or c5 or Gemm or blas or batched or gru or scan" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -E "FAILED|ERROR|passed|failed|rc=" $O/pytest.log | head
cd /tmp
for p in 0 1; do PTHIP_SGEMM_PERSIST=$p timeout 300 python $R/tools/bench_gemm.py 0 1 2 3 6 2>&1 | sed "s/^/persist=$p /"; done | tee $O/sgemm_persist.txt
