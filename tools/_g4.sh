export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2g
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -6 $O/pytest.log
cd /tmp
timeout 600 python $R/bench.py > $O/bench.json 2> $O/bench.err
python - <<P
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("C4", d["value"], d["ms_per_step"], d["roofline"]["detail"]["kernel_ms"], d["roofline"]["frac"], d["cpu_baseline"]["value"])
for k,v in d["configs"].items(): print(k, round(v["ms_device"],5), round(v.get("ms_call",0),5), round(v["frac"],4))
P
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pk4 -o k -- python $R/tools/profile_c4_replay.py 40 > $O/c4_replay.log 2>&1
DB=$(find /tmp/pk4 -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $DB > $O/c4_kernel_stats.md
python $R/tools/rocpd_timeline.py $DB 24 > $O/c4_timeline.md
tail -1 $O/c4_replay.log; head -12 $O/c4_kernel_stats.md; cat $O/c4_timeline.md
PTHIP_PLAN_TRACE=1 timeout 200 python $R/tools/profile_c4_replay.py 300 2>&1 | tail -4 | cut -c1-300 | tee $O/plan_trace.txt
PTHIP_TAIL_PRELOAD=0 timeout 200 python $R/tools/profile_c4_replay.py 300 2>&1 | tail -1 | sed 's/^/tail_preload=0 /' | tee -a $O/plan_trace.txt
