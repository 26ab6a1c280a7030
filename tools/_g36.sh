cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3u; mkdir -p $O
for n in 128 512 1024; do timeout 120 python tools/bench_lu.py $n 2>&1 | tail -1; done | tee $O/lu_sizes.txt
timeout 200 python tools/bench_decomp.py 2>&1 | tail -12 | tee $O/decomp_sizes.txt
