cd $GRAFT_REPO_ROOT
O=gpurun_out/r3m; mkdir -p $O
export TMPDIR=/tmp
(timeout 300 python tools/dotew_variants.py 256 acc4 acc4,ntb) > $O/dotew_nt.txt 2>&1
timeout 200 python tools/bench_configs.py chol c3 --reps 5 --no-check 2>/dev/null | cut -c1-420 > $O/configs.txt
cat $O/dotew_nt.txt; cat $O/configs.txt
