#!/bin/bash
# Round profile set (run on the GPU box through gpurun): the bench line, rocprofv3 kernel stats of the
# same command, the steady-state replay timeline of config #4, HBM traffic from two separate PMC
# passes (the pool refuses --pmc together with the trace domains), rocprofv3 kernel stats of the
# other BASELINE configs (#1/#2, #3, #5), the kernel microbenchmarks.
# usage: bash tools/profile_round.sh <tag>     -> gpurun_out/<tag>/*  (copy what is kept to profiles/)
TAG=${1:-round}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
python $R/bench.py --steps 300 --warmup 30 > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats -d /tmp/pk_$TAG -o k -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-configs --no-live-pmc > $O/bench_under_rocprof.json 2> $O/kt.err
python $R/tools/rocpd_stats.py $(find /tmp/pk_$TAG -name "*.db" | head -1) > $O/c4_kernel_stats.md
rocprofv3 --kernel-trace -d /tmp/pr_$TAG -o k -- python $R/tools/profile_c4_replay.py 40 > $O/c4_replay.log 2>&1
python $R/tools/rocpd_timeline.py $(find /tmp/pr_$TAG -name "*.db" | head -1) 24 > $O/c4_timeline.md
rocprofv3 --pmc FETCH_SIZE -d /tmp/pf_$TAG -o f -- python $R/tools/profile_c4_replay.py 30 > /dev/null 2> $O/pmcf.err
rocprofv3 --pmc WRITE_SIZE -d /tmp/pw_$TAG -o w -- python $R/tools/profile_c4_replay.py 30 > /dev/null 2> $O/pmcw.err
python $R/tools/pmc_summary.py $(find /tmp/pf_$TAG -name "*.db" | head -1) $(find /tmp/pw_$TAG -name "*.db" | head -1) $O/pmc_c4.json > $O/pmc_c4.txt
for cfg in c2 c3; do
  rocprofv3 --kernel-trace --stats -d /tmp/pc_${cfg}_$TAG -o k -- python $R/tools/bench_configs.py $cfg --reps 10 --no-check > $O/${cfg}_under_rocprof.json 2> $O/${cfg}.err
  python $R/tools/rocpd_stats.py $(find /tmp/pc_${cfg}_$TAG -name "*.db" | head -1) > $O/${cfg}_kernel_stats.md
done
rocprofv3 --kernel-trace --stats -d /tmp/pc_c5_$TAG -o k -- python $R/tools/profile_c5_small.py > $O/c5_small.log 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/pc_c5_$TAG -name "*.db" | head -1) > $O/c5_kernel_stats.md
python $R/tools/rocpd_timeline.py $(find /tmp/pc_c5_$TAG -name "*.db" | head -1) 16 > $O/c5_timeline.md
python $R/tools/bench_gemm.py > $O/gemm.json 2>&1
python $R/tools/bench_gemv.py > $O/gemv.json 2>&1
python $R/tools/bench_lu.py > $O/lu.json 2>&1
tail -c 600 $O/bench.json; echo; head -8 $O/c4_kernel_stats.md; cat $O/pmc_c4.txt | head -4; head -7 $O/c2_kernel_stats.md; head -8 $O/c3_kernel_stats.md; head -7 $O/c5_kernel_stats.md; cat $O/gemm.json $O/gemv.json
# round 4: linalg tier next to the host (LU, Eigh, decompositions, the GP graph), fixed-cost ubenches
python $R/tools/bench_getrf.py 256 512 1024 2048 4096 > $O/getrf.txt 2>&1
python $R/tools/bench_eigh.py 64 128 256 512 1024 2048 > $O/eigh.txt 2>&1
python $R/tools/bench_decomp.py > $O/decomp.txt 2>&1
PTHIP_GP_NODES=1 python $R/tools/bench_gp.py 512 2048 4096 > $O/gp.txt 2>&1
