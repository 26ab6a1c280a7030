#!/bin/bash
# rocprofv3 evidence for the three kernel families of tools/bench_hotpath.py (run on the GPU box through gpurun):
#   kernel-trace summaries (median launch durations per kernel name) of the ew / careduce / softmax groups, and HBM
#   traffic (FETCH_SIZE x2 per the guide's gfx950 correction, WRITE_SIZE; separate --pmc passes) of one representative
#   case per family.  usage: bash tools/profile_hotpath.sh <tag>  -> gpurun_out/<tag>/  (copy what is kept to profiles/)
TAG=${1:-r5hot}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for grp in ew careduce softmax; do
  rm -rf /tmp/ph_$grp
  rocprofv3 --kernel-trace --stats -d /tmp/ph_$grp -o k -- python $R/tools/bench_hotpath.py $grp --reps 10 > $O/${grp}_under_rocprof.jsonl 2> $O/${grp}.err
  python $R/tools/rocpd_stats.py $(find /tmp/ph_$grp -name "*.db" | head -1) > $O/${grp}_kernel_stats.md
done
PMC_GROUPS=8,9 python $R/tools/pmc_kernels.py $O/ew_traffic.md ewt_ -- python $R/tools/bench_hotpath.py ew --reps 3 --only ew_rowcol_4096 > /dev/null 2>&1
PMC_GROUPS=8,9 python $R/tools/pmc_kernels.py $O/ewT_traffic.md ewt_ -- python $R/tools/bench_hotpath.py ew --reps 3 --only ew_transposed_4096 > /dev/null 2>&1
PMC_GROUPS=8,9 python $R/tools/pmc_kernels.py $O/careduce_traffic.md rnd_ -- python $R/tools/bench_hotpath.py careduce --reps 3 --only transposed_axis1 > /dev/null 2>&1
PMC_GROUPS=8,9 python $R/tools/pmc_kernels.py $O/softmax_traffic.md softmax -- python $R/tools/bench_hotpath.py softmax --reps 3 --only softmax_axis1_8192 > /dev/null 2>&1
PMC_GROUPS=8,9 python $R/tools/pmc_kernels.py $O/lse_traffic.md rnd_ -- python $R/tools/bench_hotpath.py softmax --reps 3 --only logsumexp_axis1_8192 > /dev/null 2>&1
for f in ew ewT careduce softmax lse; do echo "== $f"; grep -A6 "^###" $O/${f}_traffic.md | head -16; done
head -14 $O/ew_kernel_stats.md | cut -c1-150; head -8 $O/careduce_kernel_stats.md | cut -c1-150; head -10 $O/softmax_kernel_stats.md | cut -c1-150
