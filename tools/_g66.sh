cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3y; mkdir -p $O
cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_c4 -o c4 -- python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-via-function > $O/c4_bench_under_profiler.json 2>/dev/null
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $(find $O/prof_c4 -name "*.db" | head -1) 2>/dev/null | head -12 | tee $O/c4_kernel_stats.md
