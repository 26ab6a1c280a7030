export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2f
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_dotew.py -q > $O/pytest_dotew.log 2>&1; echo "rc=$?" >> $O/pytest_dotew.log; tail -4 $O/pytest_dotew.log
cd /tmp
for ch in 8 16; do PTHIP_DOTEW_CHUNK=$ch timeout 300 python $R/tools/bench_configs.py c5 2>/dev/null | cut -c1-140 | sed "s/^/chunk=$ch /"; done | tee $O/c5_chunk.txt
for nt in 0 1; do
  PTHIP_NT_LOADS=$nt timeout 300 python $R/bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-configs > $O/bench_nt$nt.json 2>$O/bench_nt$nt.err
  python - <<P
import json
d=json.loads(open("$O/bench_nt$nt.json").read().strip().splitlines()[-1])
print("nt=$nt C4", d["value"], d["ms_per_step"], d["roofline"]["detail"]["kernel_ms"], d["roofline"]["frac"])
P
  PTHIP_NT_LOADS=$nt timeout 300 python $R/tools/bench_configs.py c2 2>/dev/null | cut -c1-200 | sed "s/^/nt=$nt /"
done | tee $O/nt.txt
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pk4 -o k -- python $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-configs > $O/bench_under_rocprof.json 2> $O/kt.err
DB=$(find /tmp/pk4 -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $DB > $O/c4_kernel_stats.md
python $R/tools/rocpd_timeline.py $DB 36 > $O/c4_timeline.md
head -14 $O/c4_kernel_stats.md; cat $O/c4_timeline.md
PTHIP_PLAN_TRACE=1 timeout 200 python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-configs 2>&1 | tail -12 | cut -c1-300 > $O/plan_trace.txt; cat $O/plan_trace.txt
