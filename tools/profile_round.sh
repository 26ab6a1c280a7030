#!/bin/bash
# Round profile set (run on the GPU box through gpurun): the bench line, rocprofv3 kernel stats +
# replay timeline of the same command, HBM traffic from two separate PMC passes (the pool refuses
# --pmc together with the trace domains), the other BASELINE configs and the LU-op timings.
# usage: bash tools/profile_round.sh <tag>     -> gpurun_out/<tag>/*  (copy what is kept to profiles/)
TAG=${1:-round}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
python $R/bench.py --steps 300 --warmup 30 > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats -d /tmp/pk_$TAG -o k -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/kt.err
DB=$(find /tmp/pk_$TAG -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $DB > $O/c4_kernel_stats.md
python $R/tools/rocpd_timeline.py $DB 40 > $O/c4_timeline.md
rocprofv3 --pmc FETCH_SIZE -d /tmp/pf_$TAG -o f -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline > /dev/null 2> $O/pmcf.err
rocprofv3 --pmc WRITE_SIZE -d /tmp/pw_$TAG -o w -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline > /dev/null 2> $O/pmcw.err
python $R/tools/pmc_summary.py $(find /tmp/pf_$TAG -name "*.db" | head -1) $(find /tmp/pw_$TAG -name "*.db" | head -1) $O/pmc_c4.json
python $R/tools/bench_configs.py > $O/configs.json 2> $O/configs.err
python $R/tools/bench_lu.py > $O/lu.json 2>&1
tail -c 700 $O/bench.json; echo; head -8 $O/c4_kernel_stats.md; head -c 500 $O/pmc_c4.json; echo; cut -c1-170 $O/configs.json; cat $O/lu.json
