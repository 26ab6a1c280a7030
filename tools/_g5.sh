export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2h
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_refsuite.py -x -q -k "gemm or dot or bdot or c3 or c5 or Gemm or blas or indexing or subtensor or Subtensor" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
cd /tmp
for bk in 16 32; do PTHIP_SGEMM_BK=$bk timeout 300 python $R/tools/bench_gemm.py 0 1 2 3 6 2>&1 | sed "s/^/bk=$bk /"; done | tee $O/sgemm_bk.txt
timeout 600 python $R/tools/bench_configs.py c1 c2 c3 c5 > $O/configs.json 2> $O/configs.err; cut -c1-700 $O/configs.json; tail -3 $O/configs.err
