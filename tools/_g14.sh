export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2q
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_plan.py tests/test_gpu_e2e.py tests/test_gpu_parity.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -E "FAILED|ERROR|passed|failed|rc=" $O/pytest.log | head -20
cd /tmp
for lm in 0 4 8; do
  PTHIP_LIST_MAX=$lm timeout 200 python $R/tools/profile_c4_replay.py 300 2>&1 | tail -1 | sed "s/^/list_max=$lm C4 /"
  PTHIP_LIST_MAX=$lm timeout 300 python $R/tools/bench_configs.py c1 c2 c3 --no-check 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('list_max=$lm', d['key'], 'ms_device', round(d['ms_device'],5), 'ms_call', round(d['ms_call'],5))"
done | tee $O/list_sweep.txt
