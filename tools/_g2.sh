export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2e
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_dotew.py -x -q > $O/pytest_dotew.log 2>&1; echo "rc=$?" >> $O/pytest_dotew.log; tail -15 $O/pytest_dotew.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_plan.py tests/test_gpu_fullsize.py tests/test_gpu_e2e.py -q -k "scan or gru or c5 or Scan" > $O/pytest_scan.log 2>&1; echo "rc=$?" >> $O/pytest_scan.log; tail -15 $O/pytest_scan.log
cd /tmp
timeout 300 python $R/tools/bench_configs.py c5 > $O/c5.json 2> $O/c5.err; cat $O/c5.json; tail -3 $O/c5.err
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pk5 -o k -- python $R/tools/profile_c5_small.py > $O/c5_small.log 2>&1
DB=$(find /tmp/pk5 -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $DB > $O/c5_kernel_stats.md
python $R/tools/rocpd_timeline.py $DB 16 > $O/c5_timeline.md
tail -2 $O/c5_small.log; head -12 $O/c5_kernel_stats.md; cat $O/c5_timeline.md
