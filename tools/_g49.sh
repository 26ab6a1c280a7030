cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3x; mkdir -p $O
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 3000 $O/bench.json
timeout 300 python bench.py --steps 20 --warmup 5 --no-via-function > $O/bench_driver_style.json 2>> $O/bench.err; python - <<'P'
import json
d=json.loads(open("gpurun_out/r3x/bench_driver_style.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","roofline")})
P
