export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for g in 1024 2048 4096; do for u in 1 2 4; do
PTHIP_EW_MAXGRID=$g PTHIP_EW_UNROLL=$u timeout 120 python $R/tools/bench_configs.py c2 --no-check --reps 20 2>/dev/null | grep c2_cheap | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('grid=$g unroll=$u', 'ms_device', round(d['ms_device'],5), 'kernel_ms', round(d.get('kernel_ms',0),5))"
done; done | tee $R/gpurun_out/ew_sweep.txt
