#!/bin/bash
# Final tree of round 6: rocprofv3 kernel stats + steady-state timeline of the bench command, PMC traffic of the dominant kernel.
# usage: bash tools/profile_r9_c4.sh <tag>  -> gpurun_out/<tag>/*
TAG=${1:-r9c4}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/pk_$TAG -o k -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-configs --no-live-pmc > $O/bench_under_rocprof.json 2> $O/kt.err
python $R/tools/rocpd_stats.py $(find /tmp/pk_$TAG -name "*.db" | head -1) > $O/c4_kernel_stats.md
rocprofv3 --kernel-trace -d /tmp/pr_$TAG -o k -- python $R/tools/profile_c4_replay.py 40 > $O/c4_replay.log 2>&1
python $R/tools/rocpd_timeline.py $(find /tmp/pr_$TAG -name "*.db" | head -1) 24 > $O/c4_timeline.md
rocprofv3 --pmc FETCH_SIZE -d /tmp/pf_$TAG -o f -- python $R/tools/profile_c4_replay.py 30 > /dev/null 2> $O/pmcf.err
rocprofv3 --pmc WRITE_SIZE -d /tmp/pw_$TAG -o w -- python $R/tools/profile_c4_replay.py 30 > /dev/null 2> $O/pmcw.err
python $R/tools/pmc_summary.py $(find /tmp/pf_$TAG -name "*.db" | head -1) $(find /tmp/pw_$TAG -name "*.db" | head -1) $O/pmc_c4.json > $O/pmc_c4.txt
head -8 $O/c4_kernel_stats.md | cut -c1-170; cat $O/pmc_c4.txt | head -12; cut -c1-400 $O/bench_under_rocprof.json
