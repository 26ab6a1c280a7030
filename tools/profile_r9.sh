#!/bin/bash
# Round-6 closing profile of north_star's wide_200 after the generated kernels' own log1p / log ("r9" files):
# kernel stats + steady-state timeline, then counters in their own passes.   usage: bash tools/profile_r9.sh [tag]
TAG=${1:-r9}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/pw_$TAG -o k -- python $R/tools/profile_wide200.py 40 > $O/wide200_replay.log 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/pw_$TAG -name "*.db" | head -1) > $O/wide200_kernel_stats.md
python $R/tools/rocpd_timeline.py $(find /tmp/pw_$TAG -name "*.db" | head -1) 30 > $O/wide200_timeline.md
PMC_GROUPS=0,1,2,8,9 python $R/tools/pmc_kernels.py $O/wide200_pmc.md multi_,tail_,gchain_ -- python $R/tools/profile_wide200.py 12 > $O/pmc.log 2>&1
tail -2 $O/wide200_replay.log; head -14 $O/wide200_kernel_stats.md; tail -16 $O/wide200_timeline.md; sed -n 20,42p $O/wide200_pmc.md
