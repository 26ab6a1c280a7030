# GPU call 11 (round 3): the 256x256 fp32 GEMM
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3k; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_sgemm256.py -q -x 2>&1 | tail -15 > $O/sgemm256_tests.log
(echo "256-tile kernel on:"; timeout 200 python tools/bench_gemm.py 0 1 2 3 6; echo "off:"; PTHIP_SGEMM_256=0 timeout 200 python tools/bench_gemm.py 0 1 2 3 6) > $O/sgemm256_bench.txt 2>&1
timeout 300 python -m pytest tests/test_gpu_fullsize.py -q -x -k "c3 or c5" 2>&1 | tail -4 > $O/fullsize.log
timeout 200 python tools/bench_configs.py c3 c5 --reps 5 --no-check 2>/dev/null > $O/configs.txt
tail -8 $O/sgemm256_tests.log; cat $O/sgemm256_bench.txt; tail -3 $O/fullsize.log; cut -c1-600 $O/configs.txt
