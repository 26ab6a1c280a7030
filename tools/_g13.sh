export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2p
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -E "FAILED|ERROR|passed|failed|rc=" $O/pytest.log | head -30
cd /tmp
timeout 600 python $R/bench.py > $O/bench.json 2> $O/bench.err
python - <<P
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("C4", d["value"], d["ms_per_step"], d["roofline"]["kernel"][:40], d["roofline"]["detail"]["kernel_ms"], d["roofline"]["frac"], d["roofline"]["traffic"], d["cpu_baseline"]["value"])
for k,v in d["configs"].items(): print(k, round(v["ms_device"],5), round(v["frac"],4), v.get("kernel_ms"), v.get("kernel_frac"))
P
