cd $GRAFT_REPO_ROOT
O=gpurun_out/r3l; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_sgemm256.py -q -x 2>&1 | tail -3 > $O/tests.log
(echo "prefetch on:"; timeout 200 python tools/bench_gemm.py 0 2 3 6; echo "prefetch off:"; PTHIP_SGEMM_256_PF=0 timeout 200 python tools/bench_gemm.py 0 2 3 6) > $O/bench.txt 2>&1
tail -3 $O/tests.log; cat $O/bench.txt
