cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3y; mkdir -p $O
cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_linalg -o la -- python $GRAFT_REPO_ROOT/tools/linalg_workload.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $(find $O/prof_linalg -name "*.db" | head -1) 2>/dev/null | head -24 > $O/linalg_kernel_stats.md; head -14 $O/linalg_kernel_stats.md
timeout 400 python tools/pmc_kernels.py $O/linalg_pmc.md chol_dag_kernel,trsv_dag_kernel,tri_inv256_kernel -- python $GRAFT_REPO_ROOT/tools/linalg_workload.py 2>&1 | tail -3
head -40 $O/linalg_pmc.md
