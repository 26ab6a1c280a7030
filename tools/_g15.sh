cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3n; mkdir -p $O
pp='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d["configs"]; print({k:(round(c[k]["ms_device"],5), round(c[k].get("kernel_ms",0),5)) for k in ("c2_cheap","c3_bdot","c1")}, round(d["value"]), d.get("value_via_function"))'
echo "A: no cpu baseline, no via-function"; timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-via-function --no-live-pmc 2>/dev/null | python -c "$pp"
echo "B: no cpu baseline"; timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-live-pmc 2>/dev/null | python -c "$pp"
echo "C: full"; timeout 300 python bench.py --steps 20 --warmup 5 2>/dev/null | python -c "$pp"
echo "D: standalone configs"; timeout 200 python tools/bench_configs.py c1 c2 --reps 10 --no-check 2>/dev/null | cut -c1-200
