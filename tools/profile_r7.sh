#!/bin/bash
# Round-6 ("r7" files) profile set.  usage: bash tools/profile_r7.sh <tag> [parts...]   parts: hot wide cfg
TAG=${1:-r7}
shift
PARTS=${@:-hot wide cfg}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for part in $PARTS; do
case $part in
hot)
  python $R/tools/bench_hotpath.py --reps 20 --out $O/hotpath_cold.md > $O/hotpath_cold.jsonl 2> $O/hotpath.err
  ;;
wide)
  rocprofv3 --kernel-trace --stats -d /tmp/pw_$TAG -o k -- python $R/tools/profile_wide200.py 40 > $O/wide200_replay.log 2>&1
  python $R/tools/rocpd_stats.py $(find /tmp/pw_$TAG -name "*.db" | head -1) > $O/wide200_kernel_stats.md
  python $R/tools/rocpd_timeline.py $(find /tmp/pw_$TAG -name "*.db" | head -1) 30 > $O/wide200_timeline.md
  ;;
cfg)
  python $R/tools/bench_configs.py c2 c3 --reps 10 --no-check > $O/configs_cold.jsonl 2> $O/configs.err
  ;;
esac
done
tail -3 $O/*.err; head -60 $O/hotpath_cold.md; cat $O/wide200_kernel_stats.md | head -30; cat $O/wide200_timeline.md | tail -32; cat $O/configs_cold.jsonl | cut -c1-700
