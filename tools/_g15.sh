export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2r
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_dotew.py tests/test_gpu_plan.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_e2e.py -x -q -k "dotew or dot_epilogue or scan or gru or c5 or Scan or plan or c4" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -E "FAILED|ERROR|passed|failed|rc=" $O/pytest.log | head -20
cd /tmp
timeout 300 python $R/tools/bench_configs.py c5 --no-check 2>/dev/null | cut -c1-400 | tee $O/c5.txt
for pf in 0 1; do PTHIP_PARAM_FETCH=$pf timeout 200 python $R/tools/profile_c4_replay.py 400 2>&1 | tail -1 | sed "s/^/param_fetch=$pf C4 /"; done | tee $O/param_fetch.txt
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pk5 -o k -- python $R/tools/profile_c5_small.py > $O/c5_small.log 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/pk5 -name "*.db" | head -1) | head -6
