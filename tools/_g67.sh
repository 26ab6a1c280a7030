cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3y; mkdir -p $O
cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/pk_y -o k -- python $GRAFT_REPO_ROOT/tools/profile_c4_replay.py 500 > $O/c4_replay.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $(find /tmp/pk_y -name "*.db" | head -1) 2>/dev/null | head -11 | tee $O/c4_kernel_stats.md
