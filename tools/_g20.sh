export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2v
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_plan.py -x -q -k "wide_terms or c4 or hier or plan" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -E "FAILED|ERROR|passed|failed|rc=|Error" $O/pytest.log | head -12
cd /tmp
for n in 100000 1000000; do timeout 300 python $R/tools/bench_wide.py $n 2>&1 | tail -2; done | tee $O/wide.txt
for n in 1000000; do PTHIP_WIDE=0 timeout 300 python $R/tools/bench_wide.py $n 2>&1 | tail -1 | sed 's/^/WIDE=0 /'; done | tee -a $O/wide.txt
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pw -o k -- python $R/tools/bench_wide.py 1000000 20 > /dev/null 2>&1; python $R/tools/rocpd_stats.py $(find /tmp/pw -name "*.db" | head -1) | head -8 | tee $O/wide_kernel_stats.md
