#!/bin/bash
# Round-6 final profile set (files r8_*): the bench line, rocprofv3 kernel stats + timeline of the bench command, PMC traffic of
# the dominant kernel, the cold / warm hot-path sweep with the kernel stats of the cold run, configs #2 / #3 cold.
# usage: bash tools/profile_r8_final.sh <tag>   -> gpurun_out/<tag>/*
TAG=${1:-r8}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
python $R/bench.py --steps 300 --warmup 30 > $O/bench.json 2> $O/bench.err
cp $R/gpurun_out/bench_detail.json $O/bench_detail.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d /tmp/pk_$TAG -o k -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-configs --no-live-pmc > $O/bench_under_rocprof.json 2> $O/kt.err
python $R/tools/rocpd_stats.py $(find /tmp/pk_$TAG -name "*.db" | head -1) > $O/c4_kernel_stats.md
rocprofv3 --kernel-trace -d /tmp/pr_$TAG -o k -- python $R/tools/profile_c4_replay.py 40 > $O/c4_replay.log 2>&1
python $R/tools/rocpd_timeline.py $(find /tmp/pr_$TAG -name "*.db" | head -1) 24 > $O/c4_timeline.md
rocprofv3 --pmc FETCH_SIZE -d /tmp/pf_$TAG -o f -- python $R/tools/profile_c4_replay.py 30 > /dev/null 2> $O/pmcf.err
rocprofv3 --pmc WRITE_SIZE -d /tmp/pw_$TAG -o w -- python $R/tools/profile_c4_replay.py 30 > /dev/null 2> $O/pmcw.err
python $R/tools/pmc_summary.py $(find /tmp/pf_$TAG -name "*.db" | head -1) $(find /tmp/pw_$TAG -name "*.db" | head -1) $O/pmc_c4.json > $O/pmc_c4.txt
python $R/tools/bench_hotpath.py --reps 20 --out $O/hotpath_cold.md > $O/hotpath_cold.jsonl 2> $O/hotpath.err
rocprofv3 --kernel-trace --stats -d /tmp/ph_$TAG -o k -- python $R/tools/bench_hotpath.py ew careduce --reps 8 > $O/hotpath_under_rocprof.jsonl 2> $O/hk.err
python $R/tools/rocpd_stats.py $(find /tmp/ph_$TAG -name "*.db" | head -1) > $O/hotpath_cold_kernel_stats.md
python $R/tools/bench_configs.py c1 c2 c3 c5 wide200 wide200gemm --reps 10 > $O/configs.jsonl 2> $O/configs.err
tail -c 2200 $O/bench.json; echo; head -8 $O/c4_kernel_stats.md | cut -c1-200; cat $O/pmc_c4.txt | head -4; grep -c frac $O/hotpath_cold.jsonl; cut -c1-400 $O/configs.jsonl
