cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3z; mkdir -p $O
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err
timeout 1200 bash tools/profile_round.sh r3z > $O/profile_round.log 2>&1
tail -c 1200 $O/bench_driver_style.json; echo; tail -40 $O/profile_round.log
