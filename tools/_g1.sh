export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2d
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
cd /tmp
timeout 600 python $R/bench.py > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pk5 -o k -- python $R/tools/profile_c5_small.py > $O/c5_small.log 2>&1
DB=$(find /tmp/pk5 -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $DB > $O/c5_kernel_stats.md
python $R/tools/rocpd_timeline.py $DB 24 > $O/c5_timeline.md
cat $O/c5_small.log | tail -3; head -20 $O/c5_kernel_stats.md; cat $O/c5_timeline.md
